"""Fused head (relu -> BatchNorm1d -> dropout -> Linear -> strand mean -> BCE) against a float64 torch
restatement of models/ChromeModels.py:48-51 + finetune.py:43,45,52.  (The reference-recorded vectors
reach this code through tests/test_gpu_loop.py and the model tests, which run the whole step.)"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from chromegcn_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ref_head(x64, bn, out, target64, training):
    """float64 CPU: per-strand BatchNorm (sequential running-stat updates), mean of logits, BCE"""
    logits = []
    for s in range(x64.shape[0]):
        y = bn(F.relu(x64[s]))
        logits.append(out(y))
    pred = sum(logits) / len(logits)
    loss = F.binary_cross_entropy_with_logits(pred, target64)
    return loss, torch.sigmoid(pred)


def make(S, n, d, C, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(S, n, d, generator=g) * 1.3 + 0.2
    tgt = (torch.rand(n, C, generator=g) < 0.2).float()
    bn = nn.BatchNorm1d(d); out = nn.Linear(d, C)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(d, generator=g)); bn.bias.copy_(0.1 * torch.randn(d, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(d, generator=g)); bn.running_var.copy_(1 + 0.3 * torch.rand(d, generator=g))
        out.weight.copy_(torch.randn(C, d, generator=g) / np.sqrt(d) * 2)
    return x, tgt, bn, out


@pytest.mark.parametrize("S,n,d,C", [(2, 333, 128, 103), (1, 37, 128, 19), (2, 2, 128, 7), (2, 150, 256, 200),
                                     (1, 70, 256, 130), (2, 4100, 128, 103)])
def test_head_train_matches_float64(S, n, d, C):
    x, tgt, bn, out = make(S, n, d, C, 7)
    import copy
    bn64, out64 = copy.deepcopy(bn).double(), copy.deepcopy(out).double()
    bn64.train()
    x64 = x.double().requires_grad_(True)
    loss64, probs64 = ref_head(x64, bn64, out64, tgt.double(), True)
    (loss64 * 1.7).backward()

    bn, out = bn.to(DEV), out.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    rng = torch.tensor([1, 0], dtype=torch.int64, device=DEV)
    loss, probs = ops.head_loss(xg, bn, out, tgt.to(DEV), True, 0.0, rng)
    (loss * 1.7).backward()
    assert abs(loss.item() - loss64.item()) < 1e-5
    np.testing.assert_allclose(probs.cpu().numpy(), probs64.detach().numpy(), atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), x64.grad.numpy(), atol=1e-4 * x64.grad.abs().max().item(), rtol=1e-4)
    for a, b in [(bn.weight, bn64.weight), (bn.bias, bn64.bias), (out.weight, out64.weight), (out.bias, out64.bias)]:
        ref = b.grad.numpy()
        np.testing.assert_allclose(a.grad.cpu().numpy(), ref, atol=1e-5 * max(1.0, np.abs(ref).max()), rtol=1e-4)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), bn64.running_mean.numpy(), atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), bn64.running_var.numpy(), atol=1e-6, rtol=1e-5)
    assert int(bn.num_batches_tracked.item()) == S == int(bn64.num_batches_tracked.item())


@pytest.mark.parametrize("S,n,d,C", [(2, 333, 128, 103), (1, 1, 128, 5), (2, 90, 256, 257 - 1)])
def test_head_eval_matches_float64(S, n, d, C):
    x, tgt, bn, out = make(S, n, d, C, 8)
    import copy
    bn64, out64 = copy.deepcopy(bn).double().eval(), copy.deepcopy(out).double()
    with torch.no_grad():
        loss64, probs64 = ref_head(x.double(), bn64, out64, tgt.double(), False)
    bn, out = bn.to(DEV).eval(), out.to(DEV)
    with torch.no_grad():
        loss, probs = ops.head_loss(x.to(DEV), bn, out, tgt.to(DEV), False, 0.2, None)
    assert abs(loss.item() - loss64.item()) < 1e-5
    np.testing.assert_allclose(probs.cpu().numpy(), probs64.numpy(), atol=1e-5, rtol=1e-4)
    assert int(bn.num_batches_tracked.item()) == 0


def _probe_mask(n_rows, d, p, seed, counter):
    """Recover the kernel's keep-mask for element indices [0, n_rows*d): with bn_w = 0, bn_b = 1,
    W_out = I, b_out = 0 and S = 1 the logits ARE mask / (1-p).  The mask is a pure function of
    (seed, counter, element index), so rows [n, 2n) of this probe are strand 1 of an S = 2 call."""
    bn = nn.BatchNorm1d(d).to(DEV); out = nn.Linear(d, d).to(DEV)
    with torch.no_grad():
        bn.weight.zero_(); bn.bias.fill_(1.0); out.weight.copy_(torch.eye(d)); out.bias.zero_()
    rng = torch.tensor([seed, counter], dtype=torch.int64, device=DEV)
    x = torch.randn(1, n_rows, d, device=DEV)
    _, probs = ops.head_loss(x, bn, out, torch.zeros(n_rows, d, device=DEV), True, p, rng)
    return (probs > 0.6).cpu()  # sigmoid(1/(1-p)) > 0.73 when kept, sigmoid(0) = 0.5 when dropped


def test_head_dropout_forward_and_backward_use_the_same_mask():
    """float64 restatement with the kernel's own mask made explicit: loss, probs and every gradient must
    match, which they only do if forward and backward regenerate identical masks."""
    import copy
    S, n, d, C, p = 2, 61, 128, 11, 0.3
    x, tgt, bn, out = make(S, n, d, C, 9)
    mask = _probe_mask(S * n, d, p, 1234, 5).view(S, n, d).double()
    assert 0.6 < mask.mean().item() < 0.8
    bn64, out64 = copy.deepcopy(bn).double().train(), copy.deepcopy(out).double()
    x64 = x.double().requires_grad_(True)
    logits = [out64(bn64(F.relu(x64[s])) * mask[s] / (1 - p)) for s in range(S)]
    pred = sum(logits) / S
    loss64 = F.binary_cross_entropy_with_logits(pred, tgt.double())
    loss64.backward()

    bn, out = bn.to(DEV), out.to(DEV)
    rng = torch.tensor([1234, 5], dtype=torch.int64, device=DEV)
    xg = x.to(DEV).requires_grad_(True)
    loss, probs = ops.head_loss(xg, bn, out, tgt.to(DEV), True, p, rng)
    loss.backward()
    assert int(rng[1].item()) == 5  # the head only reads the counter (cgcn_sgd_step advances it)
    assert abs(loss.item() - loss64.item()) < 1e-5
    np.testing.assert_allclose(probs.cpu().numpy(), torch.sigmoid(pred).detach().numpy(), atol=1e-5, rtol=1e-4)
    ref = x64.grad.numpy()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), ref, atol=1e-4 * np.abs(ref).max(), rtol=1e-4)
    for a_, b_ in [(bn.weight, bn64.weight), (bn.bias, bn64.bias), (out.weight, out64.weight), (out.bias, out64.bias)]:
        r = b_.grad.numpy()
        np.testing.assert_allclose(a_.grad.cpu().numpy(), r, atol=1e-4 * max(1e-6, np.abs(r).max()), rtol=1e-4)
    # a different step counter draws a different mask
    rng[1] = 6
    l6, _ = ops.head_loss(xg.detach(), bn, out, tgt.to(DEV), True, p, rng)
    assert l6.item() != loss.item()


def test_head_dropout_keep_rate():
    # bn_w = 0, bn_b = 1 -> y = 1 before dropout; W_out = ones/d, so pred = kept fraction / (1-p) per node
    S, n, d, C, p = 2, 500, 128, 3, 0.25
    bn = nn.BatchNorm1d(d).to(DEV); out = nn.Linear(d, C).to(DEV)
    with torch.no_grad():
        bn.weight.zero_(); bn.bias.fill_(1.0); out.weight.fill_(1.0 / d); out.bias.zero_()
    x = torch.randn(S, n, d, device=DEV)
    rng = torch.tensor([99, 0], dtype=torch.int64, device=DEV)
    _, probs = ops.head_loss(x, bn, out, torch.zeros(n, C, device=DEV), True, p, rng)
    pred = torch.logit(probs[:, 0].double())
    keep = (pred * (1 - p)).mean().item()
    assert abs(keep - (1 - p)) < 0.01, keep


@pytest.mark.parametrize("S,n,d", [(2, 333, 128), (1, 37, 128), (2, 5, 128), (2, 1500, 128), (2, 70, 256), (1, 16, 256)])
def test_layer_fwd_colstats_are_the_tile_statistics_of_relu_output(S, n, d):
    """cgcn_layer_fwd's optional colstats output (first stage of the head's BatchNorm statistics): per node tile the
    exact mean / sum of squared deviations of relu(Xn), ragged last tile included."""
    import ctypes
    from chromegcn_amd import _lib, graph as G
    from chromegcn_amd import synth
    lib = _lib.load()
    g = G.upload(G.normalize_graph("hic", synth.contact_graph(n, max(1, 3 * n), n + d), n), DEV)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(S, n, d, generator=gen).to(DEV)
    W = (torch.randn(d, d, generator=gen) / d ** 0.5).to(DEV); b = (0.1 * torch.randn(d, generator=gen)).to(DEV)
    wg = (torch.randn(d, generator=gen) / d ** 0.5).to(DEV); cg = torch.zeros(1, device=DEV)
    xn = torch.empty_like(x); gate = torch.empty(S, n, device=DEV)
    rows = ctypes.c_int(0)
    tiles = lib.cgcn_layer_fwd_colstats_plan(n, S, d, _lib.COLSTATS_RECORDS, ctypes.byref(rows))
    R = rows.value
    assert tiles == (n + R - 1) // R and R >= 1
    cs = torch.full((tiles, S, d, 2), float("nan"), device=DEV)
    P = _lib.ptr
    _lib.check(lib.cgcn_layer_fwd(_lib.stream_ptr(), n, S, d, P(g.rowptr), P(g.col), P(g.val), P(g.row_scale), P(x), P(W), P(b),
                                  P(wg), P(cg), P(xn), None, None, P(gate), 0.0, None, 0, None, P(cs), R, None), "cgcn_layer_fwd")
    y = torch.relu(xn).double().cpu().numpy()
    cs = cs.cpu().numpy()
    for t in range(tiles):
        blk = y[:, t * R:min(n, (t + 1) * R), :]
        np.testing.assert_allclose(cs[t, :, :, 0], blk.mean(axis=1), atol=1e-6, rtol=1e-5)
        np.testing.assert_allclose(cs[t, :, :, 1], ((blk - blk.mean(axis=1, keepdims=True)) ** 2).sum(axis=1), atol=1e-5, rtol=1e-4)
