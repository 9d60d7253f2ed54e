#!/usr/bin/env python3
"""Ablation experiments for the fused forward kernel (tuning tool): regular vs Poisson degrees."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from chromegcn_amd import _lib, graph as G, synth
from tools.kbench import timeit

def main():
    dev = torch.device("cuda"); lib = _lib.load()
    n, k, d, S = 5776, 87, 128, 2
    rng = np.random.RandomState(0)
    graphs = {}
    # regular: every row has exactly k distinct random neighbours (not symmetric; timing only)
    cols = np.stack([np.sort(rng.choice(n, k, replace=False)) for _ in range(n)]).astype(np.int32)
    graphs["regular"] = (np.arange(0, n * k + 1, k, dtype=np.int32), cols.ravel())
    h = G.normalize_graph("hic", synth.contact_graph(n, 250000, 21), n)
    graphs["poisson"] = (h.rowptr, h.col)
    # sorted-by-degree variant of the poisson graph: rows of a tile have similar lengths
    deg = np.diff(h.rowptr); order = np.argsort(deg, kind="stable")
    rp = np.concatenate([[0], np.cumsum(deg[order])]).astype(np.int32)
    cc = np.concatenate([h.col[h.rowptr[i]:h.rowptr[i + 1]] for i in order]).astype(np.int32)
    graphs["poisson_sorted_rows"] = (rp, cc)
    W = torch.randn(d, d, device=dev) / d ** 0.5; b = torch.zeros(d, device=dev)
    wg = torch.randn(d, device=dev) / d ** 0.5; cg = torch.zeros(1, device=dev)
    x = torch.randn(S, n, d, device=dev)
    xn, z, hh, y = (torch.empty_like(x) for _ in range(4)); gate = torch.empty(S, n, device=dev)
    P = _lib.ptr; st = _lib.stream_ptr
    for name, (rowptr, col) in graphs.items():
        rpt = torch.from_numpy(rowptr).to(dev); ct = torch.from_numpy(col).to(dev)
        rs = torch.ones(n, device=dev)
        t_sp = timeit(lambda: lib.cgcn_spmm(st(), n, n, S, d, P(rpt), P(ct), None, P(rs), P(x), P(y), None))
        t_f = timeit(lambda: lib.cgcn_layer_fwd(st(), n, S, d, P(rpt), P(ct), None, P(rs), P(x), P(W), P(b), P(wg), P(cg), P(xn), P(z), P(hh), P(gate), 0.0, None, 0, None, None, 0, None))
        t_i = timeit(lambda: lib.cgcn_layer_fwd(st(), n, S, d, P(rpt), P(ct), None, P(rs), P(x), P(W), P(b), P(wg), P(cg), P(xn), None, None, P(gate), 0.0, None, 0, None, None, 0, None))
        print(json.dumps({"graph": name, "nnz": int(col.shape[0]), "spmm_us": round(t_sp, 1), "fwd_train_us": round(t_f, 1), "fwd_infer_us": round(t_i, 1)}))

if __name__ == "__main__":
    main()
