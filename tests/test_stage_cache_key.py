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
