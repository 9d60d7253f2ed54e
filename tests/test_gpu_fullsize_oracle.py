"""Full-size parity against the ORACLE (not against the build's own kernels): whole train steps of the HIP engine in
the reference's semantics -- all four aggregations, d loss / d features, SGD -- versus oracle.finetune_epoch on the host,
at BASELINE.json's shapes:

  config1   n =  5 000, 125 k pairs            (configs[0])
  chr21     n =  5 776, 250 k pairs            (configs[1]; feature table 5.9 MB: shallow-batch gather kernels)
  chr1      n = 29 910, 250 k pairs, hic-like  (feature table 30.6 MB: the DEEP = true gather kernels, only reachable
                                                at this size)
  both13k   n = 13 000, 'both' adjacency       (explicit-value CSR, table 13.3 MB > 12 MB: HAS_VAL + DEEP kernels)
  k562      n =  5 776, d = 256, L = 4         (configs[3]; beyond the reference -- its constructor builds 1 or 2 layers --
                                                so the oracle is the restatement rule "repeat models/ChromeModels.py:42-46")

  chr21_hub / chr1_hub / chr21_hub_d256: heavy-tailed degrees with hubs of 2 000 ... 10 000 neighbours (top-K-style graphs,
                data/7create_graph_new.py:93-104): the hub routes of the gather kernels at full size
  chr21_C164 / chr21_C256 / d256_C256: the same chr21-size step with 164 and 256 labels (the reference takes C from the
                data, main.py:35): the training head walks labels in passes of 128 (cgcn_head_train), so these are the
                only cases in which a second pass accumulates dym and the 256-row partial layout is used.

Checked per step: loss, sigmoid(pred), every parameter gradient, d loss / d features of both strands, the parameters
after the SGD step, BatchNorm running statistics.  Tolerance: fp32 atol = rtol = 1e-4 (north star) on every tensor
against the fp32 oracle, AND -- because gradients of a mean-reduced loss are ~1e-6 and pass any absolute 1e-4 -- the
SCALE-RELATIVE error max|hip - truth| / max|truth| of every gradient, where truth is the same oracle step run in
float64.  Bound: 1e-4 for EVERY tensor, the gate-bias gradients `W*.bias` included.  Those are single scalars -- the
sum of ~10^4-10^5 signed per-row terms that cancel to 0.17 % of their absolute sum -- and they amplify any error that is
COHERENT over the rows by 600x: round 2's fp32 BatchNorm-backward column means (per-column constants in every row's
dL/dXn) put dW2.bias at 2.2e-4; the float64 second stage of those sums (cgcn_common.hpp, head_stats_finalize) is what
keeps it below the fp32 oracle's own error now (tests/probes/bias_sum_probe.py separates the stages).  All measured figures
(HIP and fp32 oracle, side by side) are printed."""
import numpy as np
import pytest
import torch

import chromegcn_amd as C
from chromegcn_amd import synth
from chromegcn_amd.finetune import GCNStage
from oracle import chromegcn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
D, NC = 128, 103
CASES = [   # name, n, pairs, hic_like, adj_type, seed, d, layers, labels
    ("config1", 5000, 125000, False, "hic", 0, 128, 2, NC),
    ("chr21", synth.chrom_nodes("chr21"), 250000, False, "hic", 21, 128, 2, NC),
    ("chr1", synth.chrom_nodes("chr1"), 250000, True, "hic", 1, 128, 2, NC),
    ("both13k", 13000, 60000, True, "both", 7, 128, 2, NC),
    ("k562_d256_L4", synth.chrom_nodes("chr21"), 250000, False, "hic", 33, 256, 4, NC),
    # config 1 again with the forward's two-launch route forced (feature-sliced aggregation + row-local kernel, column
    # statistics merged over 2 tiles per workgroup): the route chr1 / both13k take by size, at a table that fits the L2s
    ("config1_forced_split", 5000, 125000, False, "hic", 0, 128, 2, NC),
    # more than 128 labels: two label passes of the fused head, 256-row partial records
    ("chr21_C164", synth.chrom_nodes("chr21"), 250000, False, "hic", 22, 128, 2, 164),
    ("chr21_C256", synth.chrom_nodes("chr21"), 250000, False, "hic", 23, 128, 2, 256),
    ("d256_C256", synth.chrom_nodes("chr21"), 250000, False, "hic", 24, 256, 2, 256),
    ("chr1_C129", synth.chrom_nodes("chr1"), 250000, False, "hic", 25, 128, 2, 129),
    # top-K-style graphs (synth "hub" generator: degrees 1 ... 10^4): the hub paths of the sliced gather kernels at full
    # size (cooperative wave walk by cost model, rows > 768 neighbours walked by the whole workgroup); the graphs' longest
    # row (cgcn_graph_aux::max_row_len > 2 048) sends chr21-size tables through the sliced forward too.  The fused
    # forward's own hub path (LONG_ROW) meets the oracle in tests/test_gpu_parity.py (hubs of 600 ... 1 700 neighbours)
    ("chr21_hub", synth.chrom_nodes("chr21"), 250000, "hub", "hic", 26, 128, 2, NC),
    ("chr1_hub", synth.chrom_nodes("chr1"), 250000, "hub", "hic", 27, 128, 2, NC),
    ("chr21_hub_d256", synth.chrom_nodes("chr21"), 250000, "hub", "hic", 28, 256, 2, NC),
    # the same hub graphs with a CONDITIONED model (GC weights x 8 instead of x 40: tanh and the gates leave their linear
    # range without saturating the hub windows): these hold the plain 1e-4 bound against float64, no relaxation
    ("chr21_hub_w8", synth.chrom_nodes("chr21"), 250000, "hub", "hic", 26, 128, 2, NC),
    ("chr1_hub_w8", synth.chrom_nodes("chr1"), 250000, "hub", "hic", 27, 128, 2, NC),
    # adj_type 'constant' (+-7 band + I, utils/util_methods.py:137-150): the band route -- k_band_aggregate in both layers'
    # forward, k_bwd_band with the second-stage sums, the head's slabs and the fused SGD step riding -- at chr21 size, at the
    # genome's mean chromosome size, and at d = 256 / L = 4
    ("chr21_constant", synth.chrom_nodes("chr21"), 250000, False, "constant", 29, 128, 2, NC),
    ("chr10_constant", synth.chrom_nodes("chr10"), 250000, False, "constant", 30, 128, 2, NC),
    ("k562_constant_d256_L4", synth.chrom_nodes("chr21"), 250000, False, "constant", 31, 256, 4, NC),
    # every case above runs the dense products in the library's default form (split: six bf16 MFMA partial products of an
    # exact 3-way operand split, cgcn_common.hpp); these run the fp32 MFMA chain of rounds 1-5 -- the one-launch forward,
    # the row-local forward and the ring backward each have both forms -- against the same oracle at the same bounds
    ("chr21_fp32chain", synth.chrom_nodes("chr21"), 250000, False, "hic", 21, 128, 2, NC),
    ("chr1_fp32chain", synth.chrom_nodes("chr1"), 250000, True, "hic", 1, 128, 2, NC),
    ("config1_forced_split_fp32chain", 5000, 125000, False, "hic", 0, 128, 2, NC),
]


def _scaled_oracle(seed, d=D, layers=2, labels=NC, gain=40.0):
    torch.manual_seed(seed)
    orc = O.GatedGCNOracle(d, labels, 0.0, layers)
    with torch.no_grad():  # the reference init (gain 0.02) leaves tanh / gates in their linear range: scale up
        for k in range(1, layers + 1):
            getattr(orc, "GC%d" % k).weight.mul_(gain * (128.0 / d) ** 0.5)
            getattr(orc, "W%d" % k).weight.mul_(3)
    return orc


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_train_steps_match_oracle_at_full_size(case):
    name, n, pairs, hic_like, adj_type, seed, d, layers, labels = case
    from chromegcn_amd import _lib
    _lib.load().cgcn_debug_set_fwd_split_bytes(0 if "forced_split" in name else -1)
    _lib.load().cgcn_debug_set_products(0 if name.endswith("_fp32chain") else -1)
    # The host oracle's fp32 sums depend on torch's thread count (another test module pins it to 1 at import, i.e. for a
    # whole `pytest tests` session but not for this file alone: sequential sums over a hub's 10^4 neighbours are 100x less
    # accurate than the chunked ones).  Fix it here, so that the oracle is the same oracle in every session.
    threads = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        _train_steps_case(name, n, pairs, hic_like, adj_type, seed, d, layers, labels)
    finally:
        torch.set_num_threads(threads)
        _lib.load().cgcn_debug_set_fwd_split_bytes(-1)
        _lib.load().cgcn_debug_set_products(-1)


def _train_steps_case(name, n, pairs, hic_like, adj_type, seed, d, layers, labels):
    feats = synth.chrom_features(n, d, labels, 1000 + seed)
    hic = synth.contact_graph(n, pairs, seed, hic_like)
    orc = _scaled_oracle(seed, d, layers, labels, gain=8.0 if name.endswith("_w8") else 40.0)
    model = C.ChromeGCN(d, d, labels, 0.0, True, layers)
    model.load_state_dict(orc.state_dict())
    model.to(DEV)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    stage = GCNStage(model, opt, adj_type, DEV, hip_graphs=True, input_grad=True, cache_input_aggregation=False)
    stage.add_chromosome(name, feats, hic)
    oopt = O.make_sgd(orc, 0.25)
    import copy
    orc64 = copy.deepcopy(orc).double()          # float64 truth for the conditioning analysis (same op order)
    oopt64 = O.make_sgd(orc64, 0.25)
    feats64 = {k: v.double() for k, v in feats.items()}
    cache, cache64 = {}, {}
    worst, worst32, worst_hip32 = {}, {}, {}
    for step in range(2):
        # the float64 truth starts EVERY step from the parameters (and BatchNorm statistics) the two fp32 paths hold: what
        # is compared is the arithmetic of one step on identical inputs, not two optimisation trajectories drifting apart
        orc64.load_state_dict({k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in orc.state_dict().items()})
        loss, probs, dx = stage.train_step(name)
        torch.cuda.synchronize()
        grads_hip = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()}
        ig = {}
        preds, _, tot = O.finetune_epoch(orc, {name: feats}, {name: hic}, oopt, "train", adj_type, adj_cache=cache, input_grads=ig)
        if not cache64:
            cache64[name] = cache[name].double()
        ig64 = {}
        O.finetune_epoch(orc64, {name: feats64}, {name: hic}, oopt64, "train", adj_type, adj_cache=cache64, input_grads=ig64)
        assert abs(loss.item() - tot) <= 1e-4 + 1e-4 * abs(tot), (step, loss.item(), tot)
        np.testing.assert_allclose(probs.cpu().numpy(), preds.numpy(), atol=1e-4, rtol=1e-4, err_msg="probs step %d" % step)
        # d loss / d features, both strands
        dxo = np.stack([ig[name][0].numpy(), ig[name][1].numpy()])
        np.testing.assert_allclose(dx.cpu().numpy(), dxo, atol=1e-4, rtol=1e-4)
        dx64 = np.stack([ig64[name][0].numpy(), ig64[name][1].numpy()])
        worst["dx"] = max(worst.get("dx", 0.0), _rel(dx.cpu().numpy(), dx64))
        worst32["dx"] = max(worst32.get("dx", 0.0), _rel(dxo, dx64))
        worst_hip32["dx"] = max(worst_hip32.get("dx", 0.0), _rel(dx.cpu().numpy(), dxo))
        # every parameter gradient (the oracle's .grad survives its optimizer.step())
        g64 = {k: p.grad.numpy() for k, p in orc64.named_parameters()}
        for k, p in orc.named_parameters():
            np.testing.assert_allclose(grads_hip[k], p.grad.numpy(), atol=1e-4, rtol=1e-4, err_msg=k)
            worst["d" + k] = max(worst.get("d" + k, 0.0), _rel(grads_hip[k], g64[k]))
            worst32["d" + k] = max(worst32.get("d" + k, 0.0), _rel(p.grad.numpy(), g64[k]))
            worst_hip32["d" + k] = max(worst_hip32.get("d" + k, 0.0), _rel(grads_hip[k], p.grad.numpy()))
        # parameters after the SGD step + BatchNorm running statistics
        osd = orc.state_dict()
        for k, v in model.state_dict().items():
            tol = 1e-5 if "running" in k else 1e-4
            np.testing.assert_allclose(v.cpu().numpy(), osd[k].numpy(), atol=tol, rtol=tol, err_msg="%s after step %d" % (k, step))
    print("\n[%s] scale-relative max error vs the float64 oracle: HIP / fp32 oracle   |   HIP vs the fp32 oracle" % name)
    for k in sorted(worst):
        print("   %-22s %.2e / %.2e   |   %.2e" % (k, worst[k], worst32[k], worst_hip32[k]))
    # Bound: 1e-4 against float64.  On the top-K-style (hub) graphs with the x40 test weights fp32 arithmetic itself is
    # ill-conditioned: hubs of 10^4 neighbours saturate tanh (1 - z^2 of an fp32-rounded z: the GC*.weight gradients of BOTH
    # fp32 paths are 1e-3 off the float64 truth and 5e-5 apart), and the bias-type sums cancel to 6e-4 of their absolute sum,
    # so two fp32 evaluation orders differ from each other by 3e-4 ... 1.4e-3 there (third column; profiles/
    # r04_hub_parity_three_way.txt) -- an agreement of the two fp32 paths is therefore NOT asserted.  For those tensors the
    # bound is 10x the fp32 oracle's own error, on the x40 hub cases only and only where the oracle is more than 2e-5 off the
    # truth.  What carries the strict claim for hub graphs are the conditioned cases (*_w8: the same graphs, GC weights x8):
    # 1e-4 against float64 for every tensor, no relaxation, like every other case.
    hub = "hub" in name and not name.endswith("_w8")
    def bound(k):
        return 10.0 * worst32[k] if (hub and worst32[k] > 2e-5) else 1e-4
    bad = {k: (worst[k], worst32[k], worst_hip32[k]) for k in worst if worst[k] > bound(k)}
    assert not bad, "scale-relative gradient error above its bound (HIP vs float64, fp32 oracle vs float64, HIP vs fp32 oracle): %s" % bad


def test_eval_forward_matches_oracle_at_chr1_size():
    """inference path (eval-mode BatchNorm, no dropout) through the DEEP gather kernels"""
    name, n, pairs, hic_like, adj_type, seed, _d, _layers, _labels = CASES[2]
    feats = synth.chrom_features(n, D, NC, 1000 + seed)
    hic = synth.contact_graph(n, pairs, seed, hic_like)
    orc = _scaled_oracle(seed)
    with torch.no_grad():
        orc.batch_norm.running_mean.uniform_(-0.2, 0.2)
        orc.batch_norm.running_var.uniform_(0.5, 1.5)
    model = C.ChromeGCN(D, D, NC, 0.0, True, 2)
    model.load_state_dict(orc.state_dict())
    model.to(DEV)
    stage = GCNStage(model, None, adj_type, DEV, hip_graphs=True)
    stage.add_chromosome(name, feats, hic)
    loss, probs = stage.eval_step(name)
    preds, _, tot = O.finetune_epoch(orc, {name: feats}, {name: hic}, None, "valid", adj_type)
    assert abs(loss.item() - tot) <= 1e-4 + 1e-4 * abs(tot)
    np.testing.assert_allclose(probs.cpu().numpy(), preds.numpy(), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("d,layers,input_grad", [(256, 1, False), (256, 2, False), (128, 1, False)])
def test_head_second_stage_riding_in_the_row_local_launch(d, layers, input_grad):
    """When nobody differentiates the features (the engine's default) the LAST launch of the step has no gather, so the head's
    deferred second stage (dW_out / db_out slabs, BatchNorm column sums) rides in the row-local launch instead -- at d = 256 in
    k_bwd_rowlocal256s's trailing workgroups (staged through its slot memory, 1 024 threads), at d = 128 in the ring kernel's.
    One-layer models take that placement for the ONLY layer.  Two train steps against the oracle."""
    n, pairs, labels, seed = 3000, 40000, NC, 41
    feats = synth.chrom_features(n, d, labels, 1000 + seed)
    hic = synth.contact_graph(n, pairs, seed)
    torch.manual_seed(seed)
    orc = O.GatedGCNOracle(d, labels, 0.0, layers)
    with torch.no_grad():
        for k in range(1, layers + 1):
            getattr(orc, "GC%d" % k).weight.mul_(20.0 * (128.0 / d) ** 0.5)
            getattr(orc, "W%d" % k).weight.mul_(3)
    model = C.ChromeGCN(d, d, labels, 0.0, True, layers)
    model.load_state_dict(orc.state_dict())
    model.to(DEV)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    stage = GCNStage(model, opt, "hic", DEV, hip_graphs=True, input_grad=input_grad, cache_input_aggregation=False)
    stage.add_chromosome("c", feats, hic)
    oopt = O.make_sgd(orc, 0.25)
    cache = {}
    threads = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        for step in range(2):
            loss, probs, _ = stage.train_step("c")
            torch.cuda.synchronize()
            preds, _, tot = O.finetune_epoch(orc, {"c": feats}, {"c": hic}, oopt, "train", "hic", adj_cache=cache)
            assert abs(loss.item() - tot) <= 1e-4 + 1e-4 * abs(tot)
            np.testing.assert_allclose(probs.cpu().numpy(), preds.numpy(), atol=1e-4, rtol=1e-4)
            osd = orc.state_dict()
            for k, v in model.state_dict().items():
                tol = 1e-5 if "running" in k else 1e-4
                np.testing.assert_allclose(v.cpu().numpy(), osd[k].numpy(), atol=tol, rtol=tol, err_msg="%s after step %d" % (k, step))
    finally:
        torch.set_num_threads(threads)
