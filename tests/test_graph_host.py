"""Host logic (no GPU): the product's own graph normaliser against the golden vectors recorded
from the reference's process_graph and against the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

from chromegcn_amd import graph as G
from oracle import chromegcn_oracle as O
from helpers import coo_to_csr, csr_from


def test_normalize_graph_matches_reference_golden(golden):
    z = golden("g1_process_graph.npz")
    for name in z["cases"]:
        a_in = csr_from(z, "%s_in" % name)
        n = a_in.shape[0]
        for adj_type in ["hic", "constant", "both", "none"]:
            key = "%s_%s" % (name, adj_type)
            if key + "_row" not in z.files:
                continue
            h = G.normalize_graph(adj_type, a_in, n)
            assert h.rowptr.dtype == np.int32 and h.col.dtype == np.int32
            ref = coo_to_csr(z, key, n)
            got = h.to_scipy()
            # same sparsity pattern wherever the reference stores a non-zero
            ref.eliminate_zeros()
            assert (got != 0).astype(int).sum() == (ref != 0).astype(int).sum()
            np.testing.assert_allclose(got.toarray(), ref.toarray(), rtol=2e-7, atol=0)
            # implicit-value graphs really are uniform per row
            if h.val is None:
                deg = np.diff(h.rowptr)
                np.testing.assert_array_equal(h.row_scale[deg > 0], (1.0 / deg[deg > 0]).astype(np.float32))
            assert h.symmetric


def test_both_graph_carries_values():
    a = O.random_symmetric_graph(40, 60, 3)
    h = G.normalize_graph("both", a, 40)
    assert h.val is not None and set(np.unique(h.val)).issubset({1.0, 2.0, 3.0})
    np.testing.assert_allclose(h.to_scipy().toarray(), O.normalized_adjacency("both", a, 40).toarray(), rtol=2e-7)


def test_small_n_and_bad_args():
    h = G.normalize_graph("constant", None, 3)
    np.testing.assert_allclose(h.to_scipy().toarray(), np.full((3, 3), 1 / 3, np.float32), rtol=1e-7)
    h1 = G.normalize_graph("hic", sp.csr_matrix((1, 1)), 1)
    assert h1.nnz == 1 and h1.row_scale[0] == 1.0
    with pytest.raises(ValueError):
        G.normalize_graph("random", None, 4)
    with pytest.raises(ValueError):
        G.normalize_graph("hic", sp.csr_matrix((3, 3)), 4)


def test_negative_diagonal_gives_empty_row():
    # hic_ii = -1 cancels the added identity: the reference keeps a zero there and the row
    # normaliser maps 1/0 -> 0 (utils/util_methods.py:103)
    a = sp.csr_matrix(np.array([[-1.0, 0, 0], [0, 0, 1.0], [0, 1.0, 0]]))
    h = G.normalize_graph("hic", a, 3)
    assert h.rowptr[1] - h.rowptr[0] == 0 and h.row_scale[0] == 0.0
    np.testing.assert_allclose(h.to_scipy().toarray(), O.normalized_adjacency("hic", a, 3).toarray())


def test_asymmetric_matrix_detected():
    m = sp.csr_matrix(np.array([[0.5, 0.5, 0], [0, 1.0, 0], [0.2, 0.3, 0.5]], dtype=np.float32))
    h = G.host_csr_from_matrix(m)
    assert not h.symmetric and h.row_scale is None
    np.testing.assert_array_equal(h.to_scipy().toarray(), m.toarray())


def test_binary_csr_cache_roundtrip(tmp_path):
    for adj_type, hic in [("hic", O.random_symmetric_graph(50, 120, 1)), ("both", O.random_symmetric_graph(40, 60, 2)),
                          ("constant", None), ("none", None)]:
        n = 50 if adj_type == "hic" else 40
        h = G.normalize_graph(adj_type, hic, n)
        p = str(tmp_path / ("g_%s.cgcsr" % adj_type))
        G.save_csr_cache(p, h)
        h2 = G.load_csr_cache(p)
        assert h2.n == h.n and h2.symmetric == h.symmetric
        np.testing.assert_array_equal(h2.rowptr, h.rowptr)
        np.testing.assert_array_equal(h2.col, h.col)
        np.testing.assert_array_equal(h2.row_scale, h.row_scale)
        assert (h2.val is None) == (h.val is None)
        if h.val is not None:
            np.testing.assert_array_equal(h2.val, h.val)
    with open(str(tmp_path / "bad.cgcsr"), "wb") as f:
        f.write(b"not a cache file at all")
    with pytest.raises(ValueError):
        G.load_csr_cache(str(tmp_path / "bad.cgcsr"))


def test_binary_csr_cache_rejects_invalid_structure(tmp_path):
    """the kernels trust the CSR: a corrupted cache file must fail in load_csr_cache, not read out of bounds on the GPU"""
    import dataclasses
    h = G.normalize_graph("hic", O.random_symmetric_graph(30, 60, 3), 30)
    bad_col = h.col.copy(); bad_col[5] = 30                      # column index == n
    neg_col = h.col.copy(); neg_col[0] = -1
    bad_ptr = h.rowptr.copy(); bad_ptr[3], bad_ptr[4] = bad_ptr[4] + 2, bad_ptr[3]   # non-monotonic
    for i, broken in enumerate([dataclasses.replace(h, col=bad_col), dataclasses.replace(h, col=neg_col),
                                dataclasses.replace(h, rowptr=bad_ptr)]):
        p = str(tmp_path / ("broken%d.cgcsr" % i))
        G.save_csr_cache(p, broken)
        with pytest.raises(ValueError):
            G.load_csr_cache(p)
    p = str(tmp_path / "short.cgcsr")
    G.save_csr_cache(p, h)
    data = open(p, "rb").read()
    open(p, "wb").write(data[:-20])                               # truncated file
    with pytest.raises(ValueError):
        G.load_csr_cache(p)


def test_convert_graph_pickle(tmp_path):
    """the reference's on-disk graph contract (data/7create_graph_new.py:197-202) -> flat per-chromosome files"""
    import pickle
    graphs = {"chr21": O.random_symmetric_graph(30, 80, 3), "chr22": O.random_symmetric_graph(17, 30, 4)}
    pkl = str(tmp_path / "train_graphs_500000_SQRTVCnorm.pkl")
    with open(pkl, "wb") as f:
        pickle.dump(graphs, f)
    out = G.convert_graph_pickle(pkl, str(tmp_path / "csr"), "hic")
    assert sorted(out) == ["chr21", "chr22"]
    for c, path in out.items():
        h = G.load_csr_cache(path)
        np.testing.assert_allclose(h.to_scipy().toarray(), O.normalized_adjacency("hic", graphs[c], graphs[c].shape[0]).toarray(), rtol=2e-7)


def test_band_recognition_from_the_csr_arrays():
    """graph.band_halfwidth (-> cgcn_graph_aux::band_halfwidth, the library's sliding-window route): exactly the 'constant'
    graph of utils/util_methods.py:137-150 qualifies"""
    import torch
    for n in (1, 2, 7, 8, 15, 100, 1001):
        h = G.normalize_graph("constant", None, n)
        rp, c = torch.from_numpy(h.rowptr), torch.from_numpy(h.col)
        assert G.band_halfwidth(rp, c, None, n) == 7, n
        assert G.band_halfwidth(rp, c, torch.ones(c.numel()), n) == 0          # explicit values
        assert G.band_halfwidth(rp, c, None, n + 1) == 0                        # not square
    rng = np.random.RandomState(0)
    a = sp.random(300, 300, 0.02, random_state=rng, format="csr")
    a = a + a.T
    a.data[:] = 1
    for adj in ("hic", "both", "none"):
        h = G.normalize_graph(adj, a, 300)
        assert G.band_halfwidth(torch.from_numpy(h.rowptr), torch.from_numpy(h.col), None if h.val is None else torch.from_numpy(h.val), 300) == 0
    # same row lengths and row ends as the band, one interior column replaced by a duplicate-free other one
    h = G.normalize_graph("constant", None, 50)
    c = h.col.copy()
    k = h.rowptr[20] + 3
    c[k] = c[k] - 0   # unchanged: still a band
    assert G.band_halfwidth(torch.from_numpy(h.rowptr), torch.from_numpy(c), None, 50) == 7
    c2 = h.col.copy()
    c2[h.rowptr[20] + 3] = c2[h.rowptr[20] + 2]   # a repeated column inside a row
    assert G.band_halfwidth(torch.from_numpy(h.rowptr), torch.from_numpy(c2), None, 50) == 0


def test_band_plus_decomposition_of_both_graphs_is_exact():
    """graph.band_plus_part (-> cgcn_graph_aux::bp_*): for process_graph's 'both' graph of a {0,1} matrix, merged = unit
    entries + band + I exactly; anything else is refused"""
    import torch
    rng = np.random.RandomState(1)
    for n in (1, 5, 8, 40, 700):
        a = sp.random(n, n, min(1.0, 6.0 / n), random_state=rng, format="csr")
        a = a + a.T
        a.data[:] = 1
        if n > 3:
            a = a.tolil(); a[2, 2] = 1; a = sp.csr_matrix(a)      # a self-loop in the input: merged diagonal value 2
        h = G.normalize_graph("both", a, n)
        if h.val is None:
            continue
        bp = G.band_plus_part(torch.from_numpy(h.rowptr), torch.from_numpy(h.col), torch.from_numpy(h.val), n)
        assert bp is not None, n
        merged = sp.csr_matrix((h.val, h.col, h.rowptr), shape=(n, n))
        unit = sp.csr_matrix((np.ones(bp[1].numel(), np.float32), bp[1].numpy(), bp[0].numpy()), shape=(n, n))
        b = G.normalize_graph("constant", None, n)
        band = sp.csr_matrix((np.ones(b.col.size, np.float32), b.col, b.rowptr), shape=(n, n))
        assert abs(merged - (unit + band)).max() == 0
    # a value 3 (Hi-C entry of value 2 inside the band) / a value 2 outside the band: not of the form
    a = sp.random(300, 300, 0.02, random_state=rng, format="csr"); a = a + a.T; a.data[:] = 1
    a2 = a.copy(); a2.data[::5] = 2
    h = G.normalize_graph("both", a2, 300)
    assert G.band_plus_part(torch.from_numpy(h.rowptr), torch.from_numpy(h.col), torch.from_numpy(h.val), 300) is None
    # implicit-value graphs have nothing to decompose
    h = G.normalize_graph("hic", a, 300)
    assert G.band_plus_part(torch.from_numpy(h.rowptr), torch.from_numpy(h.col), None, 300) is None
