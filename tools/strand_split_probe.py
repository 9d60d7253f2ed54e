#!/usr/bin/env python3
"""d = 256 layer forward: both strands in one launch against one launch per strand (VERDICT r2 #5b) (tuning tool).
Routes timed with HIP events, same inputs, results compared bitwise:
  fused S=2          k_layer_fwd<2,256>                       (the product route at d = 256)
  fused 2 x S=1      k_layer_fwd<1,256> on X[0], then on X[1]  (table per launch = n KiB instead of 2n KiB)
  split S=2          k_aggregate_sliced<2,256> (16 slices, strand = pass) + k_layer_dense<2,256>
  split 2 x S=1      k_aggregate_sliced<1,256> + k_layer_dense<1,256> per strand
python tools/strand_split_probe.py [n ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chromegcn_amd import _lib, graph as G, synth
from tools.kbench import timeit


def main():
    dev = torch.device("cuda"); lib = _lib.load()
    d, S = 256, 2
    P = _lib.ptr; st = _lib.stream_ptr
    gen = os.environ.get("GEN", "uniform")
    for n in [int(a) for a in (sys.argv[1:] or ["5776", "16264", "29910"])]:
        g = G.upload(G.normalize_graph("hic", synth.contact_graph(n, 250000, 7, gen), n), dev)
        x = torch.randn(S, n, d, device=dev)
        W = torch.randn(d, d, device=dev) / d ** 0.5; b = torch.randn(d, device=dev) * 0.1
        wg = torch.randn(d, device=dev) / d ** 0.5; cg = torch.zeros(1, device=dev)
        c16 = G.aux_ptr(g.col)

        def outs():
            return [torch.empty_like(x), torch.empty_like(x), torch.empty_like(x), torch.empty(S, n, device=dev)]

        def both(o):
            xn, z, h, gate = o
            return lambda: lib.cgcn_layer_fwd(st(), n, 2, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(W), P(b), P(wg), P(cg),
                                              P(xn), P(z), P(h), P(gate), 0.0, None, 0, None, None, 0, c16)

        def per_strand(o):
            xn, z, h, gate = o
            def f():
                rc = 0
                for s in range(2):
                    rc |= lib.cgcn_layer_fwd(st(), n, 1, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x[s]), P(W), P(b), P(wg), P(cg),
                                             P(xn[s]), P(z[s]), P(h[s]), P(gate[s]), 0.0, None, 0, None, None, 0, c16)
                return rc
            return f

        res = {"n": n, "gen": gen}
        ref = None
        for split in (False, True):
            lib.cgcn_debug_set_fwd_split_bytes(0 if split else 1 << 60)
            for name, mk in (("S2", both), ("2xS1", per_strand)):
                o = outs()
                f = mk(o)
                assert f() == 0
                torch.cuda.synchronize()
                if ref is None:
                    ref = [t.clone() for t in o]
                else:
                    res["maxdiff_%s_%s" % ("split" if split else "fused", name)] = max(float((a - r).abs().max()) for a, r in zip(o, ref))
                res["%s_%s_us" % ("split" if split else "fused", name)] = round(timeit(f, reps=100), 1)
        lib.cgcn_debug_set_fwd_split_bytes(-1)
        print(json.dumps(res)); sys.stdout.flush()


if __name__ == "__main__":
    main()
