"""Device-side process_graph (cgcn_graph_count / cgcn_graph_fill) against the golden vectors recorded
from the reference and against the host normaliser: CSR structure bit-exact, 1/deg bit-exact."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from chromegcn_amd import graph as G
from oracle import chromegcn_oracle as O
from helpers import coo_to_csr, csr_from

pytestmark = pytest.mark.gpu
DEV = "cuda"


def same(g, h):
    assert g.n == h.n and g.nnz == h.nnz and g.symmetric == h.symmetric
    np.testing.assert_array_equal(g.rowptr.cpu().numpy(), h.rowptr)
    np.testing.assert_array_equal(g.col.cpu().numpy(), h.col)
    np.testing.assert_array_equal(g.row_scale.cpu().numpy(), h.row_scale)
    assert (g.val is None) == (h.val is None)
    if h.val is not None:
        np.testing.assert_array_equal(g.val.cpu().numpy(), h.val)


def test_device_normaliser_matches_reference_golden(golden):
    z = golden("g1_process_graph.npz")
    for name in z["cases"]:
        a_in = csr_from(z, "%s_in" % name)
        n = a_in.shape[0]
        for adj_type in ["hic", "constant", "both", "none"]:
            key = "%s_%s" % (name, adj_type)
            g = G.normalize_graph_device(adj_type, a_in, n, DEV)
            same(g, G.normalize_graph(adj_type, a_in, n))
            if key + "_row" in z.files:
                ref = coo_to_csr(z, key, n)
                got = G.to_host(g).to_scipy()
                np.testing.assert_allclose(got.toarray(), ref.toarray(), rtol=2e-7, atol=0)


def test_device_normaliser_random_and_edge_cases():
    cases = [("hic", O.random_symmetric_graph(5776, 250000, 21), 5776),
             ("both", O.random_symmetric_graph(700, 3000, 2), 700),
             ("hic", sp.csr_matrix((1, 1)), 1),
             ("constant", None, 5), ("none", None, 9), ("constant", None, 1000)]
    neg = sp.csr_matrix(np.array([[-1.0, 0, 0], [0, 0, 1.0], [0, 1.0, 0]]))
    cases.append(("hic", neg, 3))           # hic_ii = -1 cancels the identity -> empty row, scale 0
    cases.append(("both", neg, 3))
    w = sp.csr_matrix(np.array([[0, 2.0, 0, 0], [2.0, 0, 0.5, 0], [0, 0.5, 0, 0], [0, 0, 0, 0]]))
    cases.append(("both", w, 4))            # non-unit weights carried through
    for adj_type, a, n in cases:
        same(G.normalize_graph_device(adj_type, a, n, DEV), G.normalize_graph(adj_type, a, n))


def test_asymmetric_input_gets_an_explicit_transpose():
    a = sp.random(60, 60, 0.1, format="csr", random_state=5)
    a.data[:] = 1.0
    g = G.normalize_graph_device("hic", a, 60, DEV)
    h = G.normalize_graph("hic", a, 60)
    assert not g.symmetric and not h.symmetric
    at = sp.csr_matrix(h.ahat().T); at.sort_indices()
    np.testing.assert_array_equal(g.rowptr_t.cpu().numpy(), at.indptr)
    np.testing.assert_array_equal(g.col_t.cpu().numpy(), at.indices)


def test_process_graph_uses_device_path_and_bad_types_raise():
    a = O.random_symmetric_graph(40, 100, 1)
    g = G.process_graph("hic", {"c": a}, 40, "c", device=DEV)
    assert g.host is None and g.rowptr.is_cuda
    with pytest.raises(ValueError):
        G.normalize_graph_device("random", a, 40, DEV)
    with pytest.raises(ValueError):
        G.normalize_graph_device("hic", a, 41, DEV)


def test_reference_style_coo_adjacency_is_converted_on_the_device():
    """finetune.py:36 hands `process_graph(...).cuda()` -- a torch sparse COO of the row-normalised adjacency -- to the
    model for every chromosome of every epoch.  graph_from_torch_sparse converts it without a host round trip and
    recognises the reference's own normalisation: 'hic' / 'constant' / 'none' come back in the kernels' implicit form
    (no values, row_scale = fp32(1/deg), one CSR for both directions), identical to normalize_graph's result; 'both'
    (values {1,2,3}/rowsum) keeps explicit values and gets the CSR of its transpose from a device sort."""
    from chromegcn_amd import ops
    n = 700
    a = O.random_symmetric_graph(n, 5000, 4)
    x = torch.randn(2, n, 128, device=DEV)
    for adj_type in ("hic", "constant", "none", "both"):
        coo = O.process_graph(adj_type, {"c": a}, n, "c").to(DEV)
        assert coo.is_cuda and coo.layout == torch.sparse_coo
        g = G.graph_from_torch_sparse(coo, DEV)
        h = G.normalize_graph(adj_type, a, n)
        np.testing.assert_array_equal(g.rowptr.cpu().numpy(), h.rowptr)
        np.testing.assert_array_equal(g.col.cpu().numpy(), h.col)
        if adj_type != "both":
            assert g.val is None and g.symmetric and g.rowptr_t is g.rowptr
            np.testing.assert_array_equal(g.row_scale.cpu().numpy(), h.row_scale)
        else:
            assert g.val is not None and g.row_scale is None and not g.symmetric    # row-normalised values: A != A^T
            ref = sp.csr_matrix(O.normalized_adjacency("both", a, n)); ref.sort_indices()
            np.testing.assert_array_equal(g.val.cpu().numpy(), ref.data.astype(np.float32))
            rt = sp.csr_matrix(ref.T); rt.sort_indices()
            np.testing.assert_array_equal(g.rowptr_t.cpu().numpy(), rt.indptr)
            np.testing.assert_array_equal(g.col_t.cpu().numpy(), rt.indices)
            np.testing.assert_array_equal(g.val_t.cpu().numpy(), rt.data.astype(np.float32))
        # and the aggregation over it equals the oracle's A X
        want = np.stack([O.normalized_adjacency(adj_type, a, n).astype(np.float64) @ x[s].cpu().numpy().astype(np.float64) for s in range(2)])
        np.testing.assert_allclose(ops.spmm(x, g).cpu().numpy(), want, atol=1e-5, rtol=1e-5)
    # an adjacency without its diagonal / with unequal row values is NOT mistaken for the reference form
    m = sp.csr_matrix(a, dtype=np.float32)
    coo = torch.sparse_coo_tensor(torch.from_numpy(np.vstack(m.tocoo().coords)).long(), torch.from_numpy(m.tocoo().data), (n, n)).to(DEV)
    g = G.graph_from_torch_sparse(coo, DEV)
    assert g.val is not None and g.row_scale is None and g.symmetric
    # the cache returns the same object for the same live tensor
    assert G.graph_from_torch_sparse(coo, DEV) is g
