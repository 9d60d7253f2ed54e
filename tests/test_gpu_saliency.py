"""SDDMM and the adjacency saliency of scripts/visualize.py:29-55, against the reference's own method
(dense adjacency with requires_grad on the CPU, here driven through the oracle model)."""
import numpy as np
import pytest
import torch

import chromegcn_amd as C
from chromegcn_amd import graph as G
from chromegcn_amd import ops
from chromegcn_amd.saliency import adjacency_saliency
from oracle import chromegcn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("S,d", [(1, 128), (2, 128), (1, 256), (2, 256)])
def test_sddmm_matches_numpy(S, d):
    for n, pairs, seed in [(257, 1500, 1), (1, 0, 2), (300, 9000, 3)]:
        h = G.normalize_graph("hic", O.random_symmetric_graph(n, pairs, seed), n)
        g = G.upload(h, DEV)
        rng = np.random.RandomState(seed)
        a = rng.randn(S, n, d).astype(np.float32); b = rng.randn(S, n, d).astype(np.float32)
        out = ops.sddmm(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), g).cpu().numpy()
        rows = np.repeat(np.arange(n), np.diff(h.rowptr))
        want = np.einsum("skd,skd->k", a[:, rows, :].astype(np.float64), b[:, h.col, :].astype(np.float64))
        np.testing.assert_allclose(out, want, atol=1e-4, rtol=1e-4)


def test_adjacency_saliency_matches_dense_autograd():
    n, d, c = 120, 128, 7
    a = O.random_symmetric_graph(n, 500, 4)
    torch.manual_seed(2)
    orc = O.GatedGCNOracle(d, c, 0.0, 2).eval()
    with torch.no_grad():
        for k, p in orc.named_parameters():
            if "GC" in k and k.endswith("weight"):
                p.copy_(torch.randn_like(p) / np.sqrt(d) * 1.5)
    x_f, x_r = torch.randn(n, d), torch.randn(n, d)
    targets = (torch.rand(n, c) < 0.3).float()
    # the reference's way: dense adjacency with requires_grad (scripts/visualize.py:29-55)
    adj = O.process_graph("hic", {"c": a}, n, "c").to_dense().requires_grad_(True)

    def dense_forward(x):
        h = x
        for k in (1, 2):
            gc, wk = getattr(orc, "GC%d" % k), getattr(orc, "W%d" % k)
            z = torch.tanh(adj @ (h @ gc.weight) + gc.bias)
            g = torch.sigmoid(wk(z))
            h = (1 - g) * h + g * z
        return orc.out(orc.batch_norm(torch.relu(h)))
    pred = (dense_forward(x_f) + dense_forward(x_r)) / 2
    torch.sigmoid(pred).backward(gradient=targets)
    adj_grad = torch.abs(adj * adj.grad).detach()
    s = adj_grad.sum(1); s[s == 0] = 1
    adj_grad = adj_grad / s.view(-1, 1)
    m, _ = torch.max(adj_grad, 1); m[m == 0] = 1
    adj_grad = adj_grad / m.view(-1, 1)

    model = C.ChromeGCN(d, d, c, 0.0, True, 2)
    model.load_state_dict(orc.state_dict())
    model.to(DEV).eval()
    g, sal = adjacency_saliency(model, x_f.to(DEV), x_r.to(DEV), C.process_graph("hic", {"c": a}, n, "c", device=DEV), targets.to(DEV))
    rows = np.repeat(np.arange(n), np.diff(g.rowptr.cpu().numpy()))
    want = adj_grad.numpy()[rows, g.col.cpu().numpy()]
    np.testing.assert_allclose(sal.cpu().numpy(), want, atol=1e-4 * np.abs(want).max(), rtol=1e-4)
    # nothing outside the pattern
    mask = np.zeros((n, n), bool); mask[rows, g.col.cpu().numpy()] = True
    assert np.all(adj_grad.numpy()[~mask] == 0)


def _dense_reference_saliency(orc, a, x_f, x_r, targets, layers=2):
    """scripts/visualize.py:29-55 as written: dense adjacency with requires_grad on the host, |adj * adj.grad|, row-sum and
    row-max normalisation"""
    n = x_f.shape[0]
    adj = O.process_graph("hic", {"c": a}, n, "c").to_dense().requires_grad_(True)

    def dense_forward(x):
        h = x
        for k in range(1, layers + 1):
            gc, wk = getattr(orc, "GC%d" % k), getattr(orc, "W%d" % k)
            z = torch.tanh(adj @ (h @ gc.weight) + gc.bias)
            g = torch.sigmoid(wk(z))
            h = (1 - g) * h + g * z
        return orc.out(orc.batch_norm(torch.relu(h)))
    pred = (dense_forward(x_f) + dense_forward(x_r)) / 2
    torch.sigmoid(pred).backward(gradient=targets)
    adj_grad = torch.abs(adj * adj.grad).detach()
    s = adj_grad.sum(1); s[s == 0] = 1
    adj_grad = adj_grad / s.view(-1, 1)
    m, _ = torch.max(adj_grad, 1); m[m == 0] = 1
    return adj_grad / m.view(-1, 1)


@pytest.mark.timeout(900)
def test_adjacency_saliency_at_chr21_size_matches_the_reference_method():
    """the reference's own dense method still fits the host at chr21 size (5 776^2 fp32 = 133 MB per matrix);
    250 000 contact pairs, both strands, two layers"""
    from chromegcn_amd import synth
    n, d, c = synth.chrom_nodes("chr21"), 128, 19
    a = synth.contact_graph(n, 250000, 21)
    torch.manual_seed(4)
    orc = O.GatedGCNOracle(d, c, 0.0, 2).eval()
    with torch.no_grad():
        for k in (1, 2):
            getattr(orc, "GC%d" % k).weight.mul_(8.0)
    feats = synth.chrom_features(n, d, c, 77, positive_rate=0.2)
    x_f, x_r, targets = feats["forward"], feats["backward"], feats["target"].float()
    threads = torch.get_num_threads()
    torch.set_num_threads(8)
    try:
        want_dense = _dense_reference_saliency(orc, a, x_f, x_r, targets)
    finally:
        torch.set_num_threads(threads)
    model = C.ChromeGCN(d, d, c, 0.0, True, 2)
    model.load_state_dict(orc.state_dict())
    model.to(DEV).eval()
    g, sal = adjacency_saliency(model, x_f.to(DEV), x_r.to(DEV), C.process_graph("hic", {"c": a}, n, "c", device=DEV), targets.to(DEV))
    rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    want = want_dense.numpy()[rows, col]
    got = sal.cpu().numpy()
    assert got.shape == want.shape == (rowptr[-1],)
    np.testing.assert_allclose(got, want, atol=1e-4, rtol=1e-4)
    mask = np.zeros((n, n), bool); mask[rows, col] = True
    assert np.all(want_dense.numpy()[~mask] == 0)                    # nothing outside the pattern


@pytest.mark.timeout(900)
def test_saliency_properties_at_chr1_size():
    """n = 29 910: the dense method would need 3.6 GB per matrix.  The SDDMM against float64 numpy on sampled rows; the
    normalisation kernel against numpy on every row; rows normalised to a maximum of exactly 1; invariance under a rescaling of
    the targets (everything before the normalisation is linear in them)"""
    from chromegcn_amd import synth
    n, d, c, S = synth.chrom_nodes("chr1"), 128, 11, 2
    a = synth.contact_graph(n, 250000, 1, True)
    graph = C.process_graph("hic", {"c": a}, n, "c", device=DEV)
    rowptr, col = graph.rowptr.cpu().numpy(), graph.col.cpu().numpy()
    rng = np.random.RandomState(5)
    A, B = rng.randn(S, n, d).astype(np.float32), rng.randn(S, n, d).astype(np.float32)
    At, Bt = torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)
    out = ops.sddmm(At, Bt, graph)
    out2 = ops.sddmm(At, Bt, graph, out=out.clone())                 # accumulate form: twice the product
    out, out2 = out.cpu().numpy(), out2.cpu().numpy()
    np.testing.assert_array_equal(out2, out + out)
    for i in rng.choice(n, 400, replace=False):
        k0, k1 = rowptr[i], rowptr[i + 1]
        want = np.einsum("sd,skd->k", A[:, i, :].astype(np.float64), B[:, col[k0:k1], :].astype(np.float64))
        np.testing.assert_allclose(out[k0:k1], want, atol=2e-4, rtol=1e-4)
    # the normalisation alone, against numpy in the reference's order of operations
    raw = torch.from_numpy(out).to(DEV)
    norm = ops.saliency_normalize(raw, graph).cpu().numpy()
    v = np.abs(out)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    s = np.zeros(n, np.float32); np.add.at(s, rows, v); s[s == 0] = 1
    q = v / s[rows]
    m = np.zeros(n, np.float32); np.maximum.at(m, rows, q); m[m == 0] = 1
    np.testing.assert_allclose(norm, q / m[rows], rtol=2e-6, atol=1e-7)   # (the row sums differ by summation order only)
    rowmax = np.zeros(n, np.float32); np.maximum.at(rowmax, rows, norm)
    assert np.all((rowmax == 1.0) | (rowmax == 0.0)) and norm.min() >= 0.0 and norm.max() <= 1.0
    # the whole path: scale invariance in the targets
    torch.manual_seed(6)
    model = C.ChromeGCN(d, d, c, 0.0, True, 2).to(DEV).eval()
    with torch.no_grad():
        model.GC1.weight.mul_(8.0); model.GC2.weight.mul_(8.0)
    feats = synth.chrom_features(n, d, c, 78, positive_rate=0.2)
    xf, xr, t = feats["forward"].to(DEV), feats["backward"].to(DEV), feats["target"].float().to(DEV)
    _, s1 = adjacency_saliency(model, xf, xr, graph, t)
    _, s2 = adjacency_saliency(model, xf, xr, graph, 2.0 * t)
    _, r1 = adjacency_saliency(model, xf, xr, graph, t, normalize=False)
    _, r2 = adjacency_saliency(model, xf, xr, graph, 2.0 * t, normalize=False)
    np.testing.assert_allclose(s2.cpu().numpy(), s1.cpu().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r2.cpu().numpy(), 2.0 * r1.cpu().numpy(), rtol=1e-5, atol=1e-12)
    assert bool(torch.isfinite(s1).all()) and float(s1.max()) == 1.0
