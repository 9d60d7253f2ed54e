#!/usr/bin/env python3
"""Times cgcn_head_train (and its phases) at several chromosome sizes (tuning tool).  CHROMEGCN_LIB selects a variant."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chromegcn_amd import _lib
from tools.kbench import timeit


def main():
    dev = torch.device("cuda"); lib = _lib.load()
    d, S = 128, 2
    C = int(os.environ.get("KH_C", 103))
    pdrop = float(os.environ.get("KH_P", 0.2))
    sizes = [a for a in sys.argv[1:] if not a.startswith("--")]
    for n in [int(a) for a in (sizes or ["5776", "15182", "29910"])]:
        x = torch.randn(S, n, d, device=dev)
        bn_w = torch.rand(d, device=dev) + 0.5; bn_b = torch.randn(d, device=dev) * 0.1
        rm = torch.zeros(d, device=dev); rv = torch.ones(d, device=dev); nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        W = torch.randn(C, d, device=dev) / d ** 0.5; b = torch.zeros(C, device=dev)
        tgt = (torch.rand(n, C, device=dev) < 0.05).float()
        rng = torch.tensor([1, 0], dtype=torch.int64, device=dev)
        wsb = lib.cgcn_head_workspace_bytes(n, S, d, C); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        probs = torch.empty(n, C, device=dev); loss = torch.empty(1, device=dev)
        sm = torch.empty(S, d, device=dev); si = torch.empty(S, d, device=dev)
        P = _lib.ptr; st = _lib.stream_ptr
        def run(ph):
            return lambda: lib.cgcn_debug_head_train_phases(st(), n, S, d, C, P(x), P(bn_w), P(bn_b), P(rm), P(rv), P(nbt), 0.1, 1e-5, P(W), P(b), P(tgt), pdrop, P(rng), P(probs), P(loss), P(sm), P(si), None, 0, 0, P(ws), wsb, ph)
        assert run(7)() == 0
        torch.cuda.synchronize()
        if "--stamps" in sys.argv:   # needs a -DRS_TIMING build loaded through CHROMEGCN_LIB
            import numpy as np
            buf = np.zeros(8 * 2 * 8, dtype=np.uint64)
            raw = ctypes.CDLL(os.environ["CHROMEGCN_LIB"])
            assert raw.cgcn_debug_rs_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
            t = buf.reshape(8, 2, 8).astype(np.int64)
            for wg in range(8):
                for role in range(2):
                    v = t[wg, role]
                    if v[0] == 0: continue
                    print("n %d wg %3d %s  S1 %.2f  bar %.2f  S2 %.2f  bar %.2f  S3 %.2f   (S1 start -> S3 end %.2f us)" % (
                        n, wg * 32, "PQ"[role], (v[1] - v[0]) / 100., (v[2] - v[1]) / 100., (v[3] - v[2]) / 100., (v[4] - v[3]) / 100.,
                        (v[5] - v[4]) / 100., (v[5] - v[0]) / 100.))
        out = {"n": n, "C": C, "all_us": round(timeit(run(7)), 1), "stats_us": round(timeit(run(1)), 1), "fused_us": round(timeit(run(2)), 1), "finish_us": round(timeit(run(4)), 1), "loss": float(loss.item())}
        print(json.dumps(out)); sys.stdout.flush()


if __name__ == "__main__":
    main()
