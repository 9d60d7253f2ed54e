#!/usr/bin/env python3
"""Kernel micro-benchmark (GPU box): times each C-ABI entry point with HIP events on synthetic
chromosomes, prints one JSON line per (shape, kernel).  Used for tuning; not part of the product."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from chromegcn_amd import _lib, graph as G, synth  # noqa: E402


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    dev = torch.device("cuda")
    lib = _lib.load()
    shapes = [("cfg1", 5000, 125000), ("chr21", 5776, 250000), ("chr10", 16264, 250000), ("chr1", 29910, 250000)]
    only = sys.argv[1:] or None
    for hic_like in (False, True):
        for name, n, pairs in shapes:
            if only and name not in only:
                continue
            h = G.normalize_graph("hic", synth.contact_graph(n, pairs, 7, hic_like), n)
            g = G.upload(h, dev)
            for d in (128,):
                W = torch.randn(d, d, device=dev) / d ** 0.5
                b = torch.zeros(d, device=dev); wg = torch.randn(d, device=dev) / d ** 0.5; cg = torch.zeros(1, device=dev)
                for S in (1, 2):
                    x = torch.randn(S, n, d, device=dev)
                    xn, z, hh, y = (torch.empty_like(x) for _ in range(4))
                    gate = torch.empty(S, n, device=dev)
                    dxn = torch.randn_like(x); dx = torch.empty_like(x); dhs = torch.empty_like(x)
                    dW = torch.empty_like(W); db = torch.empty(d, device=dev); dwg = torch.empty(d, device=dev); dcg = torch.empty(1, device=dev)
                    wsb = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
                    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
                    st = _lib.stream_ptr
                    P = _lib.ptr
                    t_spmm = timeit(lambda: lib.cgcn_spmm(st(), n, n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(y), None))
                    t_fwd = timeit(lambda: lib.cgcn_layer_fwd(st(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(W), P(b), P(wg), P(cg), P(xn), P(z), P(hh), P(gate), 0.0, None, 0, None, None, 0, None))
                    t_inf = timeit(lambda: lib.cgcn_layer_fwd(st(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(W), P(b), P(wg), P(cg), P(xn), None, None, P(gate), 0.0, None, 0, None, None, 0, None))
                    t_bwd = timeit(lambda: lib.cgcn_layer_bwd(st(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(z), P(hh), P(gate), P(W), P(wg), P(dxn), None, P(dx), P(dhs), P(dW), P(db), P(dwg), P(dcg), 0, 0.0, None, 0, None, P(ws), wsb, None, None, None))
                    gb = 4.0 * h.nnz * S * d
                    print(json.dumps({"shape": name, "hic_like": hic_like, "n": n, "nnz": h.nnz, "S": S, "d": d,
                                      "spmm_us": round(t_spmm, 1), "fwd_us": round(t_fwd, 1), "fwd_infer_us": round(t_inf, 1),
                                      "bwd_us": round(t_bwd, 1), "spmm_gather_TBps": round(gb / t_spmm / 1e6, 2),
                                      "fwd_gather_TBps": round(gb / t_fwd / 1e6, 2)}))
                    sys.stdout.flush()


if __name__ == "__main__":
    main()
