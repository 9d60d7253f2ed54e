"""Shared test helpers (CPU side)."""
import numpy as np
import scipy.sparse as sp


def csr_from(npz, prefix, n=None):
    indptr = npz[prefix + "_indptr"]
    n = len(indptr) - 1 if n is None else n
    return sp.csr_matrix((npz[prefix + "_data"], npz[prefix + "_indices"], indptr), shape=(n, n))


def coo_to_csr(npz, prefix, n):
    """the reference's COO output -> canonical CSR; explicit zeros preserved."""
    a = sp.coo_matrix((npz[prefix + "_val"], (npz[prefix + "_row"], npz[prefix + "_col"])), shape=(n, n))
    # COO->CSR sums duplicates; the reference output has none (asserted in the test)
    return a.tocsr()


def state_from(npz, prefix):
    import torch
    out = {}
    for k in npz.files:
        if k.startswith(prefix + "_"):
            out[k[len(prefix) + 1:]] = torch.from_numpy(np.asarray(npz[k]))
    return out
