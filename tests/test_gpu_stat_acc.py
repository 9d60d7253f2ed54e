"""Accumulate mode of the head's BatchNorm batch statistics (ABI v23, include/chromegcn.h: cgcn_layer_fwd_colstats_tiles
rows = -1; d = 128 on tables that take the two-launch forward): the per-workgroup sums travel as 64-bit fixed-point integer
atomics and the head's main kernel derives mean / invstd from the totals -- no finalize launch.  Checked here: same training
as records mode (the two differ only in how the same per-workgroup statistics are combined), bit-reproducible, the running
statistics and the call count updated exactly once per step, and the engine's fallback for features outside the range."""
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


@pytest.fixture(autouse=True)
def _restore_mode():
    yield
    _lib.load().cgcn_debug_set_stat_acc(-1)
    GCNStage._stat_acc_off = False


def _train(acc, steps=3, dropout=0.2, hip_graphs=True, scale=1.0, seed=0, labels=LABELS):
    _lib.load().cgcn_debug_set_stat_acc(1 if acc else 0)
    torch.manual_seed(seed)
    model = C.ChromeGCN(128, 128, labels, dropout, True, 2).to(DEV)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    stage = GCNStage(model, opt, "hic", DEV, hip_graphs=hip_graphs, input_grad=True, cache_input_aggregation=False)
    feats = synth.chrom_features(N, 128, labels, 3)
    feats = {k: (v * scale if k != "target" else v) for k, v in feats.items()}
    stage.add_chromosome("c", feats, synth.contact_graph(N, PAIRS, 3))
    losses = []
    for _ in range(steps):
        loss, probs, _ = stage.train_step("c")
        losses.append(float(loss))
    torch.cuda.synchronize()
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return losses, state, probs.detach().clone()


def test_the_library_reports_accumulate_mode_for_this_shape():
    import ctypes
    lib = _lib.load()
    rows = ctypes.c_int(0)
    lib.cgcn_debug_set_stat_acc(1)
    tiles = lib.cgcn_layer_fwd_colstats_tiles(N, 2, 128, ctypes.byref(rows))
    assert rows.value == -1 and tiles * 2 * 128 * 2 * 4 >= (8 * 2 * 128 * 2 + 1) * 8
    lib.cgcn_debug_set_stat_acc(0)
    tiles = lib.cgcn_layer_fwd_colstats_tiles(N, 2, 128, ctypes.byref(rows))
    assert rows.value > 8 and tiles == (N + rows.value - 1) // rows.value
    lib.cgcn_debug_set_stat_acc(1)
    assert lib.cgcn_layer_fwd_colstats_tiles(5000, 2, 128, ctypes.byref(rows)) > 0 and rows.value == 8   # small table: fused route, records
    assert lib.cgcn_layer_fwd_colstats_tiles(N, 2, 256, ctypes.byref(rows)) > 0 and rows.value > 0       # d = 256: records


@pytest.mark.parametrize("dropout", [0.0, 0.2])
def test_accumulate_mode_trains_like_records_mode(dropout):
    la, sa, pa = _train(True, dropout=dropout)
    lr, sr, pr = _train(False, dropout=dropout)
    np.testing.assert_allclose(la, lr, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(pa.cpu().numpy(), pr.cpu().numpy(), rtol=1e-5, atol=1e-6)
    for k in sr:
        np.testing.assert_allclose(sa[k].float().cpu().numpy(), sr[k].float().cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    assert int(sa["batch_norm.num_batches_tracked"]) == int(sr["batch_norm.num_batches_tracked"]) == 3 * 2   # one call per strand and step
    assert not torch.equal(sa["batch_norm.running_mean"], torch.zeros_like(sa["batch_norm.running_mean"]))


def test_accumulate_mode_is_bit_reproducible_and_graph_replay_equals_eager():
    l1, s1, p1 = _train(True, steps=4)
    l2, s2, p2 = _train(True, steps=4)
    l3, s3, p3 = _train(True, steps=4, hip_graphs=False)
    assert l1 == l2 == l3
    assert torch.equal(p1, p2) and torch.equal(p1, p3)
    for k in s1:
        assert torch.equal(s1[k], s2[k]) and torch.equal(s1[k], s3[k]), k


def test_features_outside_the_fixed_point_range_fall_back_to_records():
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        losses, state, probs = _train(True, steps=2, scale=600.0)      # 9 000 x 3 000^2 >> 2^30
    assert any("fixed-point" in str(x.message) for x in w)
    assert all(np.isfinite(losses)) and bool(torch.isfinite(probs).all())
    assert bool(torch.isfinite(state["batch_norm.running_var"]).all())


def test_direct_calls_outside_the_range_read_nan_not_garbage():
    """the C ABI without the engine's guard: a workgroup partial that does not fit raises the overflow word and the statistics
    come out NaN (loud), never a wrapped integer"""
    GCNStage._stat_acc_off = True        # keep the engine from switching modes: this is the library's own behaviour
    losses, state, probs = _train(True, steps=1, scale=5000.0)     # a workgroup's 24 nodes: sum x^2 ~ 1e8 >= 2^22
    assert not np.isfinite(losses[0])


@pytest.mark.parametrize("S,n", [(1, 40), (2, 40), (1, 333), (2, 2)])
def test_small_and_single_strand_tables_in_accumulate_mode(S, n):
    """accumulate mode on tables far smaller than it is meant for (forced onto the two-launch route): fewer aggregation
    workgroups than the eight that share the zeroing, one strand, the smallest batch BatchNorm accepts -- against records mode"""
    lib = _lib.load()
    res = {}
    try:
        lib.cgcn_debug_set_fwd_split_bytes(0)
        for acc in (1, 0):
            lib.cgcn_debug_set_stat_acc(acc)
            torch.manual_seed(7)
            m = C.ChromeGCN(128, 128, 9, 0.0, True, 2).to(DEV)
            m.train()
            g = C.process_graph("hic", {"c": synth.contact_graph(n, max(1, n // 2), 3)}, n, "c", device=DEV)
            x = torch.randn(S, n, 128, device=DEV, requires_grad=True)
            tgt = (torch.rand(n, 9, device=DEV) < 0.3).float()
            loss, probs, _ = m.forward_loss(x, g, tgt)
            loss.backward()
            res[acc] = (loss.item(), probs.detach().clone(), x.grad.clone(), m.batch_norm.running_var.clone(),
                        m.out.weight.grad.clone(), int(m.batch_norm.num_batches_tracked))
    finally:
        lib.cgcn_debug_set_fwd_split_bytes(-1)
    a, r = res[1], res[0]
    assert np.isfinite(a[0]) and abs(a[0] - r[0]) < 1e-6
    assert a[5] == r[5] == S
    for u, v, what in ((a[1], r[1], "probs"), (a[2], r[2], "dx"), (a[3], r[3], "running_var"), (a[4], r[4], "dW_out")):
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=2e-5, atol=1e-7 + 2e-5 * float(v.abs().max()), err_msg=what)


def test_more_than_128_labels_two_label_passes_in_accumulate_mode():
    """C = 150: the head kernel runs twice (label passes of 128): the loss shares add up over the passes, the backward sums and
    their binary points are taken in the last pass from ALL of W_out, the bookkeeping happens in the first"""
    la, sa, pa = _train(True, steps=2, labels=150)
    lr, sr, pr = _train(False, steps=2, labels=150)
    np.testing.assert_allclose(la, lr, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(pa.cpu().numpy(), pr.cpu().numpy(), rtol=1e-5, atol=1e-6)
    for k in sr:
        np.testing.assert_allclose(sa[k].float().cpu().numpy(), sr[k].float().cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
