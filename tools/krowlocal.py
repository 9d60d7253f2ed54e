#!/usr/bin/env python3
"""Times the row-local launch of cgcn_layer_bwd alone (phase 1 of cgcn_debug_layer_bwd_phases) on random inputs
(tuning tool).  CHROMEGCN_LIB selects a variant.  python tools/krowlocal.py [n ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chromegcn_amd import _lib, graph as G, synth
from tools.kbench import timeit


def main():
    dev = torch.device("cuda"); lib = _lib.load()
    S = 2
    d = int([a[4:] for a in sys.argv[1:] if a.startswith("--d=")][0]) if any(a.startswith("--d=") for a in sys.argv[1:]) else 128
    for n in [int(a) for a in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["5776", "16264", "29910"])]:
        g = G.upload(G.normalize_graph("hic", synth.contact_graph(n, 250000, 7), n), dev)
        x, z, h, dxn = (torch.randn(S, n, d, device=dev) for _ in range(4))
        z = torch.tanh(z)
        gate = torch.rand(S, n, device=dev)
        W = torch.randn(d, d, device=dev) / d ** 0.5; wg = torch.randn(d, device=dev) / d ** 0.5
        dx, dhs = torch.empty_like(x), torch.empty_like(x)
        dW = torch.empty_like(W); db = torch.empty(d, device=dev); dwg = torch.empty(d, device=dev); dcg = torch.empty(1, device=dev)
        wsb = lib.cgcn_layer_bwd_workspace_bytes(n, S, d); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        P = _lib.ptr; st = _lib.stream_ptr
        def run(ph, with_dhs=True):
            return lambda: lib.cgcn_debug_layer_bwd_phases(st(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(z), P(h), P(gate), P(W), P(wg),
                                                           P(dxn), None, P(dx) if with_dhs else None, P(dhs) if with_dhs else None, P(dW), P(db), P(dwg), P(dcg), 0, 0.0, None, 0, None, P(ws), wsb, ph, G.aux_ptr(g.col))
        assert run(3)() == 0
        torch.cuda.synchronize()
        mb = (5 * S * n * d * 4) / 1e6
        t = timeit(run(1), reps=100)
        t_nodh = timeit(run(1, False), reps=100)
        if os.environ.get("KT"):   # -DKT_TIMING build: phase stamps of a few workgroups (100 MHz ticks -> us)
            import ctypes, numpy as np
            run(1)(); torch.cuda.synchronize()
            buf = np.zeros(8 * 16, dtype=np.uint64)
            raw = ctypes.CDLL(os.environ["CHROMEGCN_LIB"])
            assert raw.cgcn_debug_kt_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
            tt = buf.reshape(8, 16).astype(np.int64)
            t0 = tt[tt > 0].min()
            for b_ in range(8):
                print("wg", b_ * 32, " ".join("%7.2f" % ((v - t0) / 100.0) if v > 0 else "      -" for v in tt[b_]))
        print(json.dumps({"n": n, "d": d, "rowlocal_us": round(t, 1), "rowlocal_without_dHs_us": round(t_nodh, 1), "stream_MB": round(mb, 1), "TBps": round(mb / t, 2), "sliced_us": round(timeit(run(2), reps=100), 1)}))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
