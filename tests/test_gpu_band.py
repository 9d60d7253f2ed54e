"""The band route (adj_type 'constant': utils/util_methods.py:137-150, the +-7 diagonals plus I, row-normalised).
A band graph is recognised from its CSR arrays (graph.band_halfwidth -> cgcn_graph_aux::band_halfwidth) and its
aggregations run as a sliding-window stream over the feature table (k_band_aggregate / k_bwd_band) instead of the CSR walk.
Checked here: the recognition; the band kernels against a float64 scipy product and -- bit for bit, they sum in the CSR's
list order -- against the CSR kernels of the same library on the same graph (hint withheld); the whole layer backward incl.
input-dropout mask, head mode riders being covered by the full-size oracle cases (tests/test_gpu_fullsize_oracle.py
'*_constant') and the golden vectors G1/G2 (tests/test_gpu_parity.py, 'constant' cases), which now take this route."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from chromegcn_amd import _lib, graph as G, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
SIZES = [1, 2, 7, 8, 15, 31, 32, 33, 64, 257, 1000, 5776, 16264]


def _band(n):
    return G.upload(G.normalize_graph("constant", None, n), DEV)


def _without_hint(g):
    """the same graph's cgcn_graph_aux with the band / band-plus hints withheld (-> the plain CSR kernels)"""
    return G.GraphAux(G.col16_ptr(g.col), None if G.row_order(g.col) is None else G.row_order(g.col).data_ptr(), G.max_row_len(g.col), 0)


def _both_graph(n, pairs, seed, hubs=()):
    """process_graph's 'both' graph of a symmetric {0,1} contact matrix (optionally with hub rows of the given degrees)"""
    rng = np.random.RandomState(seed)
    i, j = rng.randint(0, n, pairs), rng.randint(0, n, pairs)
    for deg in hubs:
        if n > deg + 5:
            i = np.concatenate([i, np.full(deg, int(rng.randint(n)))]); j = np.concatenate([j, rng.choice(n, deg, replace=False)])
    keep = i != j
    m = sp.coo_matrix((np.ones(int(keep.sum()), dtype=np.float32), (i[keep], j[keep])), shape=(n, n)).tocsr()
    m = m + m.T
    m.data[:] = 1.0
    return G.upload(G.normalize_graph("both", m, n), DEV)


def test_band_graphs_are_recognised_and_nothing_else_is():
    for n in SIZES:
        g = _band(n)
        assert G.is_band(g.col) == (n >= 1), n
        assert g.val is None
    hic = synth.contact_graph(3000, 20000, 3)
    for adj in ("hic", "both", "none"):
        assert not G.is_band(G.upload(G.normalize_graph(adj, hic, 3000), DEV).col), adj
    # a band with one entry missing, one entry moved, explicit values: not a band
    h = G.normalize_graph("constant", None, 500)
    a = sp.csr_matrix((np.ones(h.col.size, np.float32), h.col, h.rowptr), shape=(500, 500)).tolil()
    a[100, 103] = 0
    b = sp.csr_matrix(a)
    b.eliminate_zeros()
    g = G.ChromGraph(n=500, nnz=b.nnz, rowptr=torch.from_numpy(b.indptr.astype(np.int32)).to(DEV), col=torch.from_numpy(b.indices.astype(np.int32)).to(DEV),
                     val=None, row_scale=None, rowptr_t=None, col_t=None, val_t=None, symmetric=True)
    assert not G.is_band(g.col)
    a[100, 103] = 1
    a[100, 93] = 0
    a[100, 300] = 1      # same length, same first column of the row, other last column
    b = sp.csr_matrix(a)
    b.eliminate_zeros()
    g2 = G.ChromGraph(n=500, nnz=b.nnz, rowptr=torch.from_numpy(b.indptr.astype(np.int32)).to(DEV), col=torch.from_numpy(b.indices.astype(np.int32)).to(DEV),
                      val=None, row_scale=None, rowptr_t=None, col_t=None, val_t=None, symmetric=True)
    assert not G.is_band(g2.col)
    gv = G.upload(h, DEV)
    gv2 = G.ChromGraph(n=500, nnz=gv.nnz, rowptr=gv.rowptr.clone(), col=gv.col.clone(), val=torch.ones(gv.nnz, device=DEV), row_scale=gv.row_scale,
                       rowptr_t=None, col_t=None, val_t=None, symmetric=True)
    assert not G.is_band(gv2.col)


def test_reference_style_coo_of_the_constant_graph_takes_the_band_route():
    """a caller that hands ChromeGCN.forward the torch sparse COO tensor the REFERENCE's process_graph('constant') returns
    (D^-1 (band + I), fp32 values 1 / deg_i): recognised as an implicit-value graph on the device and then as a band"""
    from oracle import chromegcn_oracle as O
    n = 3000
    coo = O.process_graph("constant", None, n, "chrX").to(DEV)
    g = G.graph_from_torch_sparse(coo, DEV)
    assert g.val is None and G.is_band(g.col)
    x = torch.randn(2, n, 128, device=DEV)
    y = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.cgcn_spmm(_lib.stream_ptr(), n, n, 2, 128, _lib.ptr(g.rowptr), _lib.ptr(g.col), None, _lib.ptr(g.row_scale), _lib.ptr(x),
                             _lib.ptr(y), G.aux_ptr(g.col)), "spmm")
    want = torch.stack([torch.sparse.mm(coo, x[s]) for s in range(2)])
    torch.testing.assert_close(y, want, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("S,d", [(2, 128), (1, 128), (2, 256), (1, 256)])
def test_band_aggregation_same_bits_as_the_csr_route_and_right(S, d):
    lib = _lib.load()
    P = _lib.ptr
    for n in SIZES:
        g = _band(n)
        x = torch.randn(S, n, d, device=DEV, generator=torch.Generator(device=DEV).manual_seed(n + d + S))
        yb, yc = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
        _lib.check(lib.cgcn_spmm(_lib.stream_ptr(), n, n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(yb), G.aux_ptr(g.col)), "spmm band")
        plain = _without_hint(g)
        lib.cgcn_debug_set_fwd_split_bytes(0)   # the feature-sliced CSR route at every size (list-order sums, like the band's)
        try:
            _lib.check(lib.cgcn_spmm(_lib.stream_ptr(), n, n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(yc), ctypes.addressof(plain)), "spmm csr")
        finally:
            lib.cgcn_debug_set_fwd_split_bytes(-1)
        assert torch.equal(yb, yc), (S, d, n)
        a = g.host.ahat().astype(np.float64)
        want = np.stack([np.asarray(sp.diags(g.host.row_scale.astype(np.float64)) @ (a @ x[s].double().cpu().numpy())) for s in range(S)])
        np.testing.assert_allclose(yb.cpu().numpy(), want, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("S,d,p", [(2, 128, 0.0), (2, 128, 0.3), (1, 256, 0.0), (2, 256, 0.25)])
def test_layer_forward_and_backward_same_bits_with_and_without_the_band_route(S, d, p):
    """cgcn_layer_fwd (H wanted: k_band_aggregate + k_layer_dense) and cgcn_layer_bwd (k_bwd_band with the second-stage
    sums riding) against the same calls with the hint withheld (forward: the sliced route forced)."""
    lib = _lib.load()
    P = _lib.ptr
    for n in (9, 64, 1000, 5776, 16264):
        g = _band(n)
        gen = torch.Generator(device=DEV).manual_seed(n + d)
        r = lambda *s: torch.randn(*s, device=DEV, generator=gen)
        x, W, b, wg, cg = r(S, n, d), r(d, d) / d ** 0.5, 0.1 * r(d), r(d) / d ** 0.5, torch.zeros(1, device=DEV)
        rng = torch.tensor([99, 5], dtype=torch.int64, device=DEV)
        plain = _without_hint(g)
        outs = []
        for aux in (G.aux_ptr(g.col), ctypes.addressof(plain)):
            if aux != G.aux_ptr(g.col):
                lib.cgcn_debug_set_fwd_split_bytes(0)
            try:
                xn, z, h, gate = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x), torch.empty(S, n, device=DEV)
                _lib.check(lib.cgcn_layer_fwd(_lib.stream_ptr(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(W), P(b), P(wg), P(cg),
                                              P(xn), P(z), P(h), P(gate), 0.0, None, 0, None, None, 0, aux), "fwd")
            finally:
                lib.cgcn_debug_set_fwd_split_bytes(-1)
            dxn = torch.randn(S, n, d, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
            dx, dhs = torch.empty_like(x), torch.empty_like(x)
            dW, db, dwg, dcg = torch.empty(d, d, device=DEV), torch.empty(d, device=DEV), torch.empty(d, device=DEV), torch.empty(1, device=DEV)
            ws_b = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
            ws = torch.empty(ws_b, dtype=torch.uint8, device=DEV)
            _lib.check(lib.cgcn_layer_bwd(_lib.stream_ptr(), n, S, d, P(g.rowptr_t), P(g.col_t), None, P(g.row_scale), P(x), P(z), P(h), P(gate),
                                          P(W), P(wg), P(dxn), None, P(dx), P(dhs), P(dW), P(db), P(dwg), P(dcg), 0, p, P(rng) if p else None, 1,
                                          None, P(ws), ws_b, None, None, aux), "bwd")
            outs.append((xn, z, h, gate, dx, dhs, dW, db, dwg, dcg))
        for nm, u, v in zip(("Xn", "Z", "H", "gate", "dX", "dHs", "dW", "db", "dwg", "dcg"), *outs):
            assert torch.equal(u, v), (nm, S, d, n, p)
    g5 = _band(5776)
    assert lib.cgcn_debug_layer_fwd_route(5776, S, d, G.aux_ptr(g5.col), 0) == 2


# ---- 'both' (Hi-C + band + I, values 1 / 2): the band-plus route of the feature-sliced kernels (cgcn_graph_aux::bp_*) -------
def test_both_graphs_carry_a_band_plus_decomposition_and_others_do_not():
    for n in (1, 7, 8, 64, 65, 1000):
        g = _both_graph(n, 4 * n, n)
        assert G.has_band_plus(g.col) == (g.val is not None), n     # (n = 1: the graph is I + I = one entry of value 2)
    hic = synth.contact_graph(3000, 20000, 3)
    for adj in ("hic", "constant", "none"):
        assert not G.has_band_plus(G.upload(G.normalize_graph(adj, hic, 3000), DEV).col), adj
    # a value outside {1, 2} (a Hi-C matrix that is not {0,1}): the explicit-value kernels keep the graph
    m = sp.csr_matrix(hic, dtype=np.float64)
    m.data[::7] = 2.0
    m = m + m.T
    assert not G.has_band_plus(G.upload(G.normalize_graph("both", m, 3000), DEV).col)


@pytest.mark.parametrize("S,d", [(2, 128), (1, 128), (2, 256), (1, 256)])
def test_band_plus_aggregation_matches_the_merged_csr_and_float64(S, d):
    lib = _lib.load()
    P = _lib.ptr
    for n, pairs, hubs in ((1, 1, ()), (7, 20, ()), (8, 30, ()), (64, 300, ()), (65, 300, ()), (1000, 8000, (200,)), (5776, 60000, (900, 2500)),
                           (16264, 250000, ())):
        g = _both_graph(n, pairs, n + d, hubs)
        if g.val is None:
            continue
        assert G.has_band_plus(g.col)
        x = torch.randn(S, n, d, device=DEV, generator=torch.Generator(device=DEV).manual_seed(n + d + S))
        yb, yc = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
        _lib.check(lib.cgcn_spmm(_lib.stream_ptr(), n, n, S, d, P(g.rowptr), P(g.col), P(g.val), P(g.row_scale), P(x), P(yb), G.aux_ptr(g.col)), "spmm bp")
        plain = _without_hint(g)
        lib.cgcn_debug_set_fwd_split_bytes(0)   # the explicit-value sliced kernel on the merged CSR
        try:
            _lib.check(lib.cgcn_spmm(_lib.stream_ptr(), n, n, S, d, P(g.rowptr), P(g.col), P(g.val), P(g.row_scale), P(x), P(yc), ctypes.addressof(plain)), "spmm csr")
        finally:
            lib.cgcn_debug_set_fwd_split_bytes(-1)
        a = g.host.ahat().astype(np.float64)
        want = np.stack([np.asarray(sp.diags(g.host.row_scale.astype(np.float64)) @ (a @ x[s].double().cpu().numpy())) for s in range(S)])
        scale = np.abs(want).max()
        assert np.abs(yb.cpu().numpy() - want).max() <= 3e-6 * scale, (S, d, n)
        assert float((yb - yc).abs().max()) <= 3e-6 * scale, (S, d, n)


@pytest.mark.parametrize("S,d,p", [(2, 128, 0.0), (2, 128, 0.3), (2, 256, 0.25)])
def test_layer_forward_and_backward_on_the_band_plus_route(S, d, p):
    """cgcn_layer_fwd (k_aggregate_sliced<BP> + k_layer_dense) and cgcn_layer_bwd (k_bwd_sliced<BP> with its riders) against the
    same calls on the merged CSR with explicit values (hints withheld, sliced route forced): same results up to fp32
    re-association of a row's sum."""
    lib = _lib.load()
    P = _lib.ptr
    for n, pairs, hubs in ((9, 20, ()), (1000, 8000, (200,)), (5776, 250000, (1200,)), (16264, 250000, ())):
        g = _both_graph(n, pairs, n + d, hubs)
        assert G.has_band_plus(g.col)
        gen = torch.Generator(device=DEV).manual_seed(n + d)
        r = lambda *s: torch.randn(*s, device=DEV, generator=gen)
        x, W, b, wg, cg = r(S, n, d), r(d, d) / d ** 0.5, 0.1 * r(d), r(d) / d ** 0.5, torch.zeros(1, device=DEV)
        rng = torch.tensor([99, 5], dtype=torch.int64, device=DEV)
        plain = _without_hint(g)
        outs = []
        for aux in (G.aux_ptr(g.col), ctypes.addressof(plain)):
            if aux != G.aux_ptr(g.col):
                lib.cgcn_debug_set_fwd_split_bytes(0)
            try:
                xn, z, h, gate = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x), torch.empty(S, n, device=DEV)
                _lib.check(lib.cgcn_layer_fwd(_lib.stream_ptr(), n, S, d, P(g.rowptr), P(g.col), P(g.val), P(g.row_scale), P(x), P(W), P(b), P(wg), P(cg),
                                              P(xn), P(z), P(h), P(gate), 0.0, None, 0, None, None, 0, aux), "fwd")
            finally:
                lib.cgcn_debug_set_fwd_split_bytes(-1)
            dxn = torch.randn(S, n, d, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
            dx, dhs = torch.empty_like(x), torch.empty_like(x)
            dW, db, dwg, dcg = torch.empty(d, d, device=DEV), torch.empty(d, device=DEV), torch.empty(d, device=DEV), torch.empty(1, device=DEV)
            ws_b = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
            ws = torch.empty(ws_b, dtype=torch.uint8, device=DEV)
            _lib.check(lib.cgcn_layer_bwd(_lib.stream_ptr(), n, S, d, P(g.rowptr_t), P(g.col_t), P(g.val_t), P(g.row_scale), P(x), P(z), P(h), P(gate),
                                          P(W), P(wg), P(dxn), None, P(dx), P(dhs), P(dW), P(db), P(dwg), P(dcg), 0, p, P(rng) if p else None, 1,
                                          None, P(ws), ws_b, None, None, aux), "bwd")
            outs.append((xn, z, h, gate, dx, dhs, dW, db, dwg, dcg))
        for nm, u, v in zip(("Xn", "Z", "H", "gate", "dX", "dHs", "dW", "db", "dwg", "dcg"), *outs):
            scale = float(v.abs().max()) + 1e-30
            assert float((u - v).abs().max()) <= 2e-5 * scale, (nm, S, d, n, p)
        # the dropout mask is a function of the element index alone: exactly the same elements are dropped on both routes
        assert torch.equal(outs[0][4] == 0, outs[1][4] == 0)
