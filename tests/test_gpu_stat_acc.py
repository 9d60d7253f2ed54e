"""Accumulate mode of the head's BatchNorm batch statistics (include/chromegcn.h: cgcn_layer_fwd_colstats_plan,
CGCN_COLSTATS_ACCUMULATE; d = 128 and, since ABI v24, d = 256; any table size -- it takes the two-launch forward): the
per-workgroup sums travel as 64-bit fixed-point integer atomics and the head's main kernel derives mean / invstd from the
totals -- no finalize / finish launch.  Checked here: same training as records mode (the two differ only in how the same
per-workgroup statistics are combined), bit-reproducible, the running statistics and the call count updated exactly once per
step, and that the mode is decided PER CHROMOSOME from its own features (round 5 flipped a process-wide
library switch): one chromosome outside the range leaves the others, other stages and later stages untouched."""
import warnings

import numpy as np
import pytest
import torch

import chromegcn_amd as C
from chromegcn_amd import _lib, synth
from chromegcn_amd.finetune import GCNStage

pytestmark = pytest.mark.gpu
DEV = "cuda"
N, PAIRS, LABELS = 9000, 60000, 21      # 9 000 x 2 x 128 x 4 B = 9.2 MB: a split-size table


def _train(acc, steps=3, dropout=0.2, hip_graphs=True, scale=1.0, seed=0, labels=LABELS, d=128, n=N, layers=2):
    torch.manual_seed(seed)
    model = C.ChromeGCN(d, d, labels, dropout, True, layers).to(DEV)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    stage = GCNStage(model, opt, "hic", DEV, hip_graphs=hip_graphs, input_grad=True, cache_input_aggregation=False, stat_acc=acc)
    feats = synth.chrom_features(n, d, labels, 3)
    feats = {k: (v * scale if k != "target" else v) for k, v in feats.items()}
    stage.add_chromosome("c", feats, synth.contact_graph(n, PAIRS, 3))
    losses = []
    for _ in range(steps):
        loss, probs, _ = stage.train_step("c")
        losses.append(float(loss))
    torch.cuda.synchronize()
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return losses, state, probs.detach().clone(), stage


def test_the_plan_for_both_modes():
    import ctypes
    lib = _lib.load()
    rows = ctypes.c_int(0)
    for d in (128, 256):
        for n in (N, 5000, 40):            # split-size, fused-size, tiny: accumulate mode is the caller's choice at every size
            tiles = lib.cgcn_layer_fwd_colstats_plan(n, 2, d, _lib.COLSTATS_ACCUMULATE, ctypes.byref(rows))
            assert rows.value == -1 and tiles * 2 * d * 2 * 4 >= (8 * 2 * d * 2 + 1) * 8
            tiles = lib.cgcn_layer_fwd_colstats_plan(n, 2, d, _lib.COLSTATS_RECORDS, ctypes.byref(rows))
            assert rows.value >= 8 and rows.value % 8 == 0 and tiles == (n + rows.value - 1) // rows.value


@pytest.mark.parametrize("d,n,layers", [(128, 5000, 2), (256, 5776, 2), (256, 9000, 3), (128, 40, 1), (256, 3000, 2), (128, 5000, 1), (128, 40, 3)])
def test_accumulate_mode_at_other_widths_and_sizes(d, n, layers):
    """VERDICT r5 #1c / #5: d = 256 (k_layer_dense256 -> k_head_fused<256> -> k_bwd_rowlocal256s) and tables below the split
    size (round 5: the fused forward, records only; now k_layer_fwd adds the totals itself, zeroed by the layer before -- a
    one-layer model zeroes in its own aggregation launch) against records mode; bit-reproducible"""
    la, sa, pa, _ = _train(True, d=d, n=n, layers=layers)
    lr, sr, pr, _ = _train(False, d=d, n=n, layers=layers)
    l2, s2, p2, _ = _train(True, d=d, n=n, layers=layers)
    assert la == l2 and torch.equal(pa, p2) and all(torch.equal(sa[k], s2[k]) for k in sa)
    np.testing.assert_allclose(la, lr, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(pa.cpu().numpy(), pr.cpu().numpy(), rtol=1e-5, atol=1e-6)
    for k in sr:
        np.testing.assert_allclose(sa[k].float().cpu().numpy(), sr[k].float().cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    assert int(sa["batch_norm.num_batches_tracked"]) == int(sr["batch_norm.num_batches_tracked"]) == 3 * 2


@pytest.mark.parametrize("dropout", [0.0, 0.2])
def test_accumulate_mode_trains_like_records_mode(dropout):
    la, sa, pa, _ = _train(True, dropout=dropout)
    lr, sr, pr, _ = _train(False, dropout=dropout)
    np.testing.assert_allclose(la, lr, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(pa.cpu().numpy(), pr.cpu().numpy(), rtol=1e-5, atol=1e-6)
    for k in sr:
        np.testing.assert_allclose(sa[k].float().cpu().numpy(), sr[k].float().cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    assert int(sa["batch_norm.num_batches_tracked"]) == int(sr["batch_norm.num_batches_tracked"]) == 3 * 2   # one call per strand and step
    assert not torch.equal(sa["batch_norm.running_mean"], torch.zeros_like(sa["batch_norm.running_mean"]))


def test_accumulate_mode_is_bit_reproducible_and_graph_replay_equals_eager():
    l1, s1, p1, _ = _train(True, steps=4)
    l2, s2, p2, _ = _train(True, steps=4)
    l3, s3, p3, _ = _train(True, steps=4, hip_graphs=False)
    assert l1 == l2 == l3
    assert torch.equal(p1, p2) and torch.equal(p1, p3)
    for k in s1:
        assert torch.equal(s1[k], s2[k]) and torch.equal(s1[k], s3[k]), k


def test_features_outside_the_fixed_point_range_fall_back_to_records():
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        losses, state, probs, stage = _train(None, steps=2, scale=600.0)      # 9 000 x 3 000^2 >> 2^29; None = the engine decides
    assert any("fixed-point" in str(x.message) for x in w)
    assert stage.chroms["c"].stat_acc is False
    assert all(np.isfinite(losses)) and bool(torch.isfinite(probs).all())
    assert bool(torch.isfinite(state["batch_norm.running_var"]).all())
    _, _, _, stage2 = _train(None, steps=1)
    assert stage2.chroms["c"].stat_acc is True                               # in range: accumulate mode by default


def test_the_fallback_is_per_chromosome_not_per_process():
    """two stages in one process, one chromosome out of range in the first.  That chromosome trains on records;
    the in-range chromosome of the same stage -- and a stage built afterwards -- stays in accumulate mode and is bit-identical
    to a stage that never saw the out-of-range chromosome."""
    def stage_with(bad_first):
        torch.manual_seed(11)
        model = C.ChromeGCN(128, 128, LABELS, 0.2, True, 2).to(DEV)
        opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
        st = GCNStage(model, opt, "hic", DEV, hip_graphs=True, input_grad=True, cache_input_aggregation=False)
        if bad_first:
            fb = synth.chrom_features(N, 128, LABELS, 5)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                st.add_chromosome("bad", {k: (v * 600.0 if k != "target" else v) for k, v in fb.items()}, synth.contact_graph(N, PAIRS, 5))
        st.add_chromosome("ok", synth.chrom_features(N, 128, LABELS, 3), synth.contact_graph(N, PAIRS, 3))
        return st, model
    sa, ma = stage_with(True)
    assert sa.chroms["bad"].stat_acc is False and sa.chroms["ok"].stat_acc is True
    sb, mb = stage_with(False)
    assert sb.chroms["ok"].stat_acc is True
    for _ in range(3):
        la, pa, _ = sa.train_step("ok")
        lb, pb, _ = sb.train_step("ok")
        assert float(la) == float(lb) and torch.equal(pa, pb)
    for (k, u), (_, v) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(u, v), k
    lbad, pbad, _ = sa.train_step("bad")                                      # records mode: finite whatever the magnitude
    assert np.isfinite(float(lbad)) and bool(torch.isfinite(pbad).all())
    # ... and forcing records for "ok" is the OTHER arithmetic (same statistics, combined in another order): close, not equal bits
    torch.manual_seed(11)
    mr = C.ChromeGCN(128, 128, LABELS, 0.2, True, 2).to(DEV)
    sr = GCNStage(mr, torch.optim.SGD(mr.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6), "hic", DEV, hip_graphs=True,
                  input_grad=True, cache_input_aggregation=False, stat_acc=False)
    sr.add_chromosome("ok", synth.chrom_features(N, 128, LABELS, 3), synth.contact_graph(N, PAIRS, 3))
    assert sr.chroms["ok"].stat_acc is False


def test_direct_calls_outside_the_range_read_nan_not_garbage():
    """accumulate mode FORCED (stat_acc=True: the caller's statement, here a false one): a workgroup partial that does not fit
    raises the overflow word and the statistics come out NaN (loud), never a wrapped integer"""
    losses, state, probs, _ = _train(True, steps=1, scale=5000.0)     # a workgroup's 24 nodes: sum x^2 ~ 1e8 >= 2^22
    assert not np.isfinite(losses[0])


def test_module_level_entry_points_default_to_records():
    """ChromeGCN.forward_loss / the custom operators take any magnitude unless the caller opts in"""
    torch.manual_seed(2)
    m = C.ChromeGCN(128, 128, 9, 0.0, True, 2).to(DEV)
    m.train()
    n = 9000
    g = C.process_graph("hic", {"c": synth.contact_graph(n, 30000, 3)}, n, "c", device=DEV)
    x = 5000.0 * torch.randn(2, n, 128, device=DEV)
    tgt = (torch.rand(n, 9, device=DEV) < 0.3).float()
    loss, probs, _ = m.forward_loss(x, g, tgt)
    assert np.isfinite(loss.item()) and bool(torch.isfinite(probs).all())
    loss2, _, _ = m.forward_loss(x, g, tgt, stat_acc=True)
    assert not np.isfinite(loss2.item())


@pytest.mark.parametrize("S,n", [(1, 40), (2, 40), (1, 333), (2, 2)])
def test_small_and_single_strand_tables_in_accumulate_mode(S, n):
    """accumulate mode on tables far smaller than it is meant for (it takes the two-launch route at every size): fewer aggregation
    workgroups than the eight that share the zeroing, one strand, the smallest batch BatchNorm accepts -- against records mode"""
    res = {}
    for acc in (1, 0):
        torch.manual_seed(7)
        m = C.ChromeGCN(128, 128, 9, 0.0, True, 2).to(DEV)
        m.train()
        g = C.process_graph("hic", {"c": synth.contact_graph(n, max(1, n // 2), 3)}, n, "c", device=DEV)
        x = torch.randn(S, n, 128, device=DEV, requires_grad=True)
        tgt = (torch.rand(n, 9, device=DEV) < 0.3).float()
        loss, probs, _ = m.forward_loss(x, g, tgt, stat_acc=bool(acc))
        loss.backward()
        res[acc] = (loss.item(), probs.detach().clone(), x.grad.clone(), m.batch_norm.running_var.clone(),
                    m.out.weight.grad.clone(), int(m.batch_norm.num_batches_tracked))
    a, r = res[1], res[0]
    assert np.isfinite(a[0]) and abs(a[0] - r[0]) < 1e-6
    assert a[5] == r[5] == S
    for u, v, what in ((a[1], r[1], "probs"), (a[2], r[2], "dx"), (a[3], r[3], "running_var"), (a[4], r[4], "dW_out")):
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=2e-5, atol=1e-7 + 2e-5 * float(v.abs().max()), err_msg=what)


def test_more_than_128_labels_two_label_passes_in_accumulate_mode():
    """C = 150: the head kernel runs twice (label passes of 128): the loss shares add up over the passes, the backward sums and
    their binary points are taken in the last pass from ALL of W_out, the bookkeeping happens in the first"""
    la, sa, pa, _ = _train(True, steps=2, labels=150)
    lr, sr, pr, _ = _train(False, steps=2, labels=150)
    np.testing.assert_allclose(la, lr, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(pa.cpu().numpy(), pr.cpu().numpy(), rtol=1e-5, atol=1e-6)
    for k in sr:
        np.testing.assert_allclose(sa[k].float().cpu().numpy(), sr[k].float().cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
