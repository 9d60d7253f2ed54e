"""The feature-sliced kernels against the whole-row kernels of the SAME library on seeded random graphs: shapes around
the kernels' boundaries (n around multiples of 64, n < 8, rows around the 192-neighbour hub threshold of the sliced
walk and the 512-neighbour one of the fused gather, empty-but-for-the-self-loop rows), both strand counts, both widths,
binary and valued adjacency.  cgcn_spmm: the sliced route sums a row in list order like the whole-row kernel, so
ordinary rows agree bitwise and hub rows to re-association; the gated layer (forward, every gradient): split route
(k_aggregate_sliced + k_layer_dense, k_bwd_sliced) vs the fused forward, tolerance of fp32 re-association."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from chromegcn_amd import _lib
from chromegcn_amd import graph as G
from chromegcn_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def random_graph(rng, n, kind):
    dens = rng.choice([0.002, 0.02, 0.2]) if n > 50 else 0.3
    m = sp.random(n, n, dens, format="lil", random_state=rng.randint(1 << 30), dtype=np.float32)
    m = ((m + m.T) > 0).astype(np.float32).tolil()
    for deg in (191, 192, 193, 600):   # rows straddling the hub thresholds
        if n > deg + 5 and rng.rand() < 0.6:
            i = int(rng.randint(n))
            sel = rng.choice(n, deg, replace=False)
            m[i, sel] = 1
            m[sel, i] = 1
    m.setdiag(0)
    return G.normalize_graph(kind, m.tocsr(), n)


def run_layer(g, S, d, seed, split):
    lib = _lib.load()
    lib.cgcn_debug_set_fwd_split_bytes(0 if split else 1 << 40)
    try:
        rng = np.random.RandomState(seed)
        t = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in dict(
            x=rng.randn(S, g.n, d).astype(np.float32), W=(rng.randn(d, d) / np.sqrt(d)).astype(np.float32),
            b=(rng.randn(d) * 0.2).astype(np.float32), wg=(rng.randn(1, d) / np.sqrt(d)).astype(np.float32),
            cg=np.array([0.1], dtype=np.float32)).items()}
        xn, gate = ops.gated_layer(t["x"], t["W"], t["b"], t["wg"], t["cg"], g)
        gup = torch.from_numpy(rng.randn(S, g.n, d).astype(np.float32)).to(DEV)
        (xn * gup).sum().add(gate.sum()).backward()
        y = ops.spmm(t["x"].detach(), g)
        return [xn.detach(), gate.detach(), y] + [t[k].grad for k in ("x", "W", "b", "wg", "cg")]
    finally:
        lib.cgcn_debug_set_fwd_split_bytes(-1)


@pytest.mark.parametrize("S,d", [(2, 128), (1, 128), (2, 256), (1, 256)])
def test_sliced_routes_agree_with_whole_row_routes(S, d):
    rng = np.random.RandomState(100 * S + d)
    sizes = [1, 2, 7, 8, 9, 63, 64, 65, 127, 128, 129, 200, 333, 640, 1000]
    for n in sizes:
        kind = ["hic", "both"][int(rng.randint(2))]
        g = G.upload(random_graph(rng, n, kind), DEV)
        a = run_layer(g, S, d, n, split=False)
        b = run_layer(g, S, d, n, split=True)
        names = ["Xn", "gate", "spmm", "dX", "dW", "db", "dwg", "dcg"]
        for nm, u, v in zip(names, a, b):
            scale = float(u.abs().max()) + 1e-30
            err = float((u - v).abs().max()) / scale
            assert err < 2e-5, "%s: n=%d %s S=%d d=%d: scale-relative difference %.2e" % (nm, n, kind, S, d, err)


@pytest.mark.parametrize("S,d", [(2, 128), (1, 256)])
def test_sixteen_bit_column_indices_give_bit_identical_results(S, d):
    """graphs with at most 65 536 columns carry a uint16 copy of their column indices (graph.aux_ptr -> cgcn_graph_aux.col16), which the
    feature-sliced kernels walk instead of the int32 list: same neighbours in the same order -> identical bits, for the
    aggregation (cgcn_spmm's sliced route) and for the backward gather (cgcn_layer_bwd); hub rows included."""
    lib = _lib.load()
    P = _lib.ptr
    rng = np.random.RandomState(7 + d)
    for n in (65, 1000, 40000):
        i = rng.randint(0, n, 15 * n); j = rng.randint(0, n, 15 * n)       # COO pairs (scipy's lil path takes minutes at 40 k)
        for deg in (193, 600):
            if n > deg + 5:
                i = np.concatenate([i, np.full(deg, int(rng.randint(n)))]); j = np.concatenate([j, rng.choice(n, deg, replace=False)])
        keep = i != j
        m = sp.coo_matrix((np.ones(int(keep.sum()), dtype=np.float32), (i[keep], j[keep])), shape=(n, n)).tocsr()
        m = m + m.T
        m.data[:] = 1.0
        g = G.upload(G.normalize_graph("hic", m, n), DEV)
        assert G.col16_ptr(g.col) is not None and g.val is None
        aux16 = G.GraphAux(G.col16_ptr(g.col), None, 0)   # cgcn_graph_aux carrying the 16-bit copy and nothing else
        c16 = ctypes.addressof(aux16)
        x = torch.randn(S, n, d, device=DEV)
        lib.cgcn_debug_set_fwd_split_bytes(0)       # the sliced route at every size
        try:
            y32, y16 = torch.empty_like(x), torch.empty_like(x)
            for y, c in ((y32, None), (y16, c16)):
                _lib.check(lib.cgcn_spmm(_lib.stream_ptr(), n, n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(y), c), "spmm")
            assert torch.equal(y32, y16)
            z, h, dxn = torch.tanh(torch.randn_like(x)), torch.randn_like(x), torch.randn_like(x)
            gate = torch.rand(S, n, device=DEV)
            W, wg = torch.randn(d, d, device=DEV) / d ** 0.5, torch.randn(d, device=DEV)
            ws_b = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
            ws = torch.empty(ws_b, dtype=torch.uint8, device=DEV)
            outs = []
            for c in (None, c16):
                dx, dhs = torch.empty_like(x), torch.empty_like(x)
                dW, db, dwg, dcg = torch.empty(d, d, device=DEV), torch.empty(d, device=DEV), torch.empty(d, device=DEV), torch.empty(1, device=DEV)
                _lib.check(lib.cgcn_layer_bwd(_lib.stream_ptr(), n, S, d, P(g.rowptr_t), P(g.col_t), None, P(g.row_scale), P(x), P(z), P(h),
                                              P(gate), P(W), P(wg), P(dxn), None, P(dx), P(dhs), P(dW), P(db), P(dwg), P(dcg), 0, 0.0, None, 0,
                                              None, P(ws), ws_b, None, None, c), "bwd")
                outs.append((dx, dW, db, dwg, dcg))
            for a, b in zip(*outs):
                assert torch.equal(a, b)
        finally:
            lib.cgcn_debug_set_fwd_split_bytes(-1)
    # more than 65 536 columns: no 16-bit copy is registered
    big = G.upload(G.normalize_graph("none", None, 70000), DEV)
    assert G.col16_ptr(big.col) is None


@pytest.mark.parametrize("S,d,kind", [(2, 128, "hic"), (1, 256, "both")])
def test_super_hub_rows_and_skewed_waves(S, d, kind):
    """rows of thousands of neighbours (walked by all 8 waves of the workgroup, SLICED_SUPER), two of them in one 64-row
    tile, next to rows of 1 ... 200 neighbours (the cost-model choice between the per-row and the cooperative walk of a
    wave): the sliced aggregation against float64 scipy, and the whole layer forward + backward against the fused /
    whole-row route of the same library."""
    rng = np.random.RandomState(3 + d)
    n = 6000
    i = rng.randint(0, n, 4 * n); j = rng.randint(0, n, 4 * n)
    hubs = [(70, 800), (71, 2500), (3000, 5000), (n - 1, 1200)]          # (row, degree); rows 70 and 71 share a tile
    hubs += [(int(r), int(dg)) for r, dg in zip(rng.choice(np.arange(200, 2900), 40, replace=False), rng.randint(100, 260, 40))]
    for r, deg in hubs:
        i = np.concatenate([i, np.full(deg, r)]); j = np.concatenate([j, rng.choice(n, deg, replace=False)])
    keep = i != j
    m = sp.coo_matrix((np.ones(int(keep.sum()), dtype=np.float32), (i[keep], j[keep])), shape=(n, n)).tocsr()
    m = m + m.T
    m.data[:] = 1.0
    host = G.normalize_graph(kind, m, n)
    g = G.upload(host, DEV)
    deg = np.diff(g.rowptr.cpu().numpy())
    assert deg.max() > 4000 and (deg > 768).sum() >= 4
    lib = _lib.load()
    P = _lib.ptr
    x = torch.randn(S, n, d, device=DEV)
    y = torch.empty_like(x)
    lib.cgcn_debug_set_fwd_split_bytes(0)
    try:
        _lib.check(lib.cgcn_spmm(_lib.stream_ptr(), n, n, S, d, P(g.rowptr), P(g.col), None if g.val is None else P(g.val),
                                 P(g.row_scale), P(x), P(y), G.aux_ptr(g.col)), "spmm")
    finally:
        lib.cgcn_debug_set_fwd_split_bytes(-1)
    # cgcn_graph_aux::max_row_len routes this graph through the sliced kernels without the debug hook, table size
    # regardless (6 ... 12 MB here: S = 2, d = 128 can be below the split threshold): same bits as the forced route
    assert G.max_row_len(g.col) == int(deg.max())
    y2 = torch.empty_like(x)
    _lib.check(lib.cgcn_spmm(_lib.stream_ptr(), n, n, S, d, P(g.rowptr), P(g.col), None if g.val is None else P(g.val),
                             P(g.row_scale), P(x), P(y2), G.aux_ptr(g.col)), "spmm")
    assert torch.equal(y, y2)
    A = sp.csr_matrix((np.ones(g.col.numel()) if g.val is None else g.val.cpu().numpy().astype(np.float64),
                       g.col.cpu().numpy(), g.rowptr.cpu().numpy()), shape=(n, n))
    rs = g.row_scale.cpu().numpy().astype(np.float64)
    for s in range(S):
        want = (A @ x[s].cpu().numpy().astype(np.float64)) * rs[:, None]
        err = np.abs(y[s].cpu().numpy() - want).max() / np.abs(want).max()
        assert err < 2e-6, "strand %d: scale-relative error %.2e" % (s, err)
    a = run_layer(g, S, d, 5, split=False)
    b = run_layer(g, S, d, 5, split=True)
    for nm, u, v in zip(["Xn", "gate", "spmm", "dX", "dW", "db", "dwg", "dcg"], a, b):
        err = float((u - v).abs().max()) / (float(u.abs().max()) + 1e-30)
        assert err < 2e-5, "%s: scale-relative difference %.2e" % (nm, err)


def test_tile_sorted_row_order_is_a_tile_local_permutation_and_changes_nothing_but_summation_order():
    """cgcn_graph_aux::row_order as the engine builds it (graph.tile_sorted_rows): the rows of every 64-row group sorted by
    length, the groups dealt to the tiles heaviest first.  The sliced aggregation gives the same sums with and without it up to fp32 re-association (a wave that walks
    its rows cooperatively adds a row's neighbours in butterfly order, and which waves do depends on their rows)."""
    lib = _lib.load()
    P = _lib.ptr
    rng = np.random.RandomState(11)
    for n in (130, 1000, 9000):
        i = rng.randint(0, n, 20 * n); j = rng.randint(0, n, 20 * n)
        for deg in (300, 900):
            deg = min(deg, n - 1)
            i = np.concatenate([i, np.full(deg, int(rng.randint(n)))]); j = np.concatenate([j, rng.choice(n, deg, replace=False)])
        keep = i != j
        m = sp.coo_matrix((np.ones(int(keep.sum()), dtype=np.float32), (i[keep], j[keep])), shape=(n, n)).tocsr()
        m = m + m.T
        m.data[:] = 1.0
        g = G.upload(G.normalize_graph("hic", m, n), DEV)
        order = G.row_order(g.col)
        assert order is not None
        o = order.cpu().numpy()
        deg_rows = np.diff(g.rowptr.cpu().numpy())
        assert sorted(o.tolist()) == list(range(n))
        weights = []
        for t in range((n + 63) // 64):
            seg = o[64 * t:64 * t + 64]
            assert len(set((seg // 64).tolist())) == 1          # a tile holds ONE group of 64 neighbouring rows
            assert np.all(np.diff(deg_rows[seg]) <= 0)           # longest first
            weights.append(int(deg_rows[seg].sum()))
        full = n // 64
        assert np.all(np.diff(weights[:full]) <= 0)              # full tiles heaviest first
        if n % 64:
            assert o[64 * full] // 64 == full                    # the partial group stays last
        S, d = 2, 128
        x = torch.randn(S, n, d, device=DEV)
        plain = G.GraphAux(G.col16_ptr(g.col), None, G.max_row_len(g.col))
        lib.cgcn_debug_set_fwd_split_bytes(0)
        try:
            ys = []
            for aux in (ctypes.addressof(plain), G.aux_ptr(g.col)):
                y = torch.empty_like(x)
                _lib.check(lib.cgcn_spmm(_lib.stream_ptr(), n, n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(y), aux), "spmm")
                ys.append(y)
        finally:
            lib.cgcn_debug_set_fwd_split_bytes(-1)
        err = float((ys[0] - ys[1]).abs().max()) / float(ys[0].abs().max())
        assert err < 2e-6, "n=%d: %.2e" % (n, err)
