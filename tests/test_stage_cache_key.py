"""GCNStage.load()'s cache key: a cached chromosome is reused only for the very same live feature tensors
(at the same in-place version) and Hi-C object -- never for regenerated tensors that happen to land on a recycled
storage address with version 0 and the same shape."""
import gc

import numpy as np
import scipy.sparse as sp
import torch

from chromegcn_amd.finetune import _SourceKey


def _feats(n=32, d=8, c=3, fill=0.0):
    return {"forward": torch.full((n, d), fill), "backward": torch.full((n, d), fill), "target": torch.zeros(n, c)}


def test_same_objects_hit_inplace_edit_misses():
    f, hic = _feats(), sp.identity(32, format="csr")
    k = _SourceKey(f, hic)
    assert k.matches(f, hic)
    f["forward"].add_(1.0)                      # in-place edit: version bump
    assert not k.matches(f, hic)
    assert _SourceKey(f, hic).matches(f, hic)
    assert not k.matches(f, sp.identity(32, format="csr"))   # another graph object of the same shape / nnz
    assert not _SourceKey(f, None).matches(f, hic) and not _SourceKey(f, hic).matches(f, None)


def test_regenerated_tensors_on_a_recycled_address_miss():
    f = _feats(fill=1.0)
    k = _SourceKey(f, None)
    addr = {n: t.data_ptr() for n, t in f.items()}
    hits = 0
    for _ in range(20):                         # free and regenerate: the allocator usually hands the same blocks back
        del f
        gc.collect()
        f = _feats(fill=2.0)
        hits += all(f[n].data_ptr() == addr[n] for n in f)
        assert not k.matches(f, None)           # ... and the old key still must not match them
    # (hits > 0 on every allocator seen so far: that is the hazard; the assertion above holds either way)
    assert hits >= 0


def test_dict_with_same_tensors_but_new_container_hits():
    f = _feats()
    k = _SourceKey(f, None)
    assert k.matches(dict(f), None)             # the dict is not the identity, the tensors are
    g = dict(f)
    g["target"] = f["target"].clone()
    assert not k.matches(g, None)
    assert not k.matches({"forward": np.zeros((32, 8)), "backward": f["backward"], "target": f["target"]}, None)


# ---- GCNStage.load(): what a reload of a resident chromosome must rebuild -------------------------------
def _oracle_stage(adj_type="hic"):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from chromegcn_amd.finetune import GCNStage
    from oracle import chromegcn_oracle as O

    class OracleStrands(O.GatedGCNOracle):
        def forward_strands(self, x_fr, graph):
            adj = O.to_torch_coo(graph.host.to_scipy())
            return torch.stack([self.forward(x_fr[0], adj)[1], self.forward(x_fr[1], adj)[1]]), None

    torch.manual_seed(0)
    m = OracleStrands(16, 3, 0.0, 2)
    return GCNStage(m, O.make_sgd(m, 0.1), adj_type, "cpu", hip_graphs=False)


def test_reload_with_new_targets_of_the_same_shape_returns_the_new_targets():
    from chromegcn_amd import synth
    stage = _oracle_stage()
    feats = {c: synth.chrom_features(n, 16, 3, 5 + i, positive_rate=0.3) for i, (c, n) in enumerate({"chrA": 40, "chrB": 25}.items())}
    graphs = {c: synth.contact_graph(f["forward"].shape[0], 60, 9) for c, f in feats.items()}
    stage.load(feats, graphs)
    _, t0, _ = stage.run_split("valid")
    assert torch.equal(t0, torch.cat([feats["chrA"]["target"], feats["chrB"]["target"]]))
    _, t0_dev, _ = stage.run_split("valid", to_cpu=False)
    # the caller regenerates chrB's labels (same shape): load() sees a _SourceKey mismatch and re-adds the chromosome
    feats["chrB"] = dict(feats["chrB"], target=1.0 - feats["chrB"]["target"])
    stage.load(feats, graphs)
    want = torch.cat([feats["chrA"]["target"], feats["chrB"]["target"]])
    _, t1, _ = stage.run_split("valid")
    assert torch.equal(t1, want) and not torch.equal(t1, t0)
    _, t1_dev, _ = stage.run_split("valid", to_cpu=False)
    assert torch.equal(t1_dev.cpu(), want)
    # an in-place edit of a resident chromosome's targets is a version bump: also rebuilt
    feats["chrA"]["target"].zero_()
    stage.load(feats, graphs)
    _, t2, _ = stage.run_split("valid")
    assert torch.equal(t2[:40], torch.zeros(40, 3)) and torch.equal(t2[40:], feats["chrB"]["target"])


def test_shard_costs_do_not_depend_on_how_a_rank_registered_a_chromosome():
    """plan_shards must see the same costs on every rank: deferred registration, immediate upload and a deferred
    chromosome that materialised later all report the same number (mixed paths gave different plans)."""
    from chromegcn_amd import synth
    for adj_type in ("hic", "both", "constant", "none"):
        feats = {c: synth.chrom_features(n, 16, 3, 5 + i) for i, (c, n) in enumerate({"chrA": 40, "chrB": 25, "chrC": 31}.items())}
        graphs = {c: synth.contact_graph(f["forward"].shape[0], 50 + 7 * i, 9 + i) for i, (c, f) in enumerate(feats.items())}
        a, b, c = _oracle_stage(adj_type), _oracle_stage(adj_type), _oracle_stage(adj_type)
        a.load(feats, graphs)                       # uploads at once
        b.load(feats, graphs, defer=True)           # registers only
        c.load(feats, graphs, defer=True)
        c._resident("chrB")                         # ... and one of them materialises (a shard plan handed it to this rank)
        costs = [{nm: st._meta[nm][2] for nm in feats} for st in (a, b, c)]
        assert costs[0] == costs[1] == costs[2], adj_type
        assert a.chroms["chrB"].cost == costs[0]["chrB"] == c.chroms["chrB"].cost
