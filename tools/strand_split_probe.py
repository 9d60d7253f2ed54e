#!/usr/bin/env python3
"""Probe (tuning tool): one strand-batched k_layer_fwd launch ([2,n,128], 1 KiB per gathered neighbour, table 2*n*512 B)
vs two single-strand launches ([1,n,128] each: table n*512 B, half the bytes per launch) on uniform / hic-like graphs.
The question: above which table size does halving the gathered table (L2 hit rate) beat sharing one pass over the CSR?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chromegcn_amd import _lib, graph as G, synth


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda"); lib = _lib.load(); d = 128
    P, st = _lib.ptr, _lib.stream_ptr
    for hic in (False, True):
        for n in (5776, 7563, 9369, 12304, 16264, 20534, 29184):
            g = G.upload(G.normalize_graph("hic", synth.contact_graph(n, 250000, n, hic), n), dev)
            x = torch.randn(2, n, d, device=dev)
            W = torch.randn(d, d, device=dev) / d ** 0.5; b = torch.zeros(d, device=dev)
            wg = torch.randn(d, device=dev) / d ** 0.5; cg = torch.zeros(1, device=dev)
            xn, z, h = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            gate = torch.empty(2, n, device=dev)

            def fwd(S, xs, xns, zs, hs, gs):
                rc = lib.cgcn_layer_fwd(st(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), xs.data_ptr(), P(W), P(b), P(wg),
                                        P(cg), xns.data_ptr(), zs.data_ptr(), hs.data_ptr(), gs.data_ptr(), 0.0, None, 1, None, None, None)
                assert rc == 0
            both = lambda: fwd(2, x, xn, z, h, gate)
            split = lambda: (fwd(1, x[0], xn[0], z[0], h[0], gate[0]), fwd(1, x[1], xn[1], z[1], h[1], gate[1]))
            both(); ref = xn.clone(); split()
            same = torch.equal(ref, xn)
            print("%s n=%5d table %5.1f MB: batched %6.1f us | split 2 x S=1 %6.1f us  (identical=%s)" %
                  ("hic-like" if hic else "uniform ", n, 2 * n * d * 4 / 1e6, timeit(both), timeit(split), same))


if __name__ == "__main__":
    main()
