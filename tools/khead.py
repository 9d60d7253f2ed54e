#!/usr/bin/env python3
"""Times the head / row-local entry points in isolation (tuning tool)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chromegcn_amd import _lib, graph as G, synth
from tools.kbench import timeit

def main():
    dev = torch.device("cuda"); lib = _lib.load()
    n, d, S, C = int(os.environ.get("KS_N", 5776)), int(os.environ.get("KS_D", 128)), 2, 103
    x = torch.randn(S, n, d, device=dev)
    bn_w = torch.rand(d, device=dev) + 0.5; bn_b = torch.randn(d, device=dev) * 0.1
    rm = torch.zeros(d, device=dev); rv = torch.ones(d, device=dev); nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    W = torch.randn(C, d, device=dev) / d ** 0.5; b = torch.zeros(C, device=dev)
    tgt = (torch.rand(n, C, device=dev) < 0.05).float()
    rng = torch.tensor([1, 0], dtype=torch.int64, device=dev)
    wsb = lib.cgcn_head_workspace_bytes(n, S, d, C); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    probs = torch.empty(n, C, device=dev); loss = torch.empty(1, device=dev); dpred = torch.empty(n, C, device=dev)
    sm = torch.empty(S, d, device=dev); si = torch.empty(S, d, device=dev)
    dX = torch.empty_like(x); dW = torch.empty_like(W); db = torch.empty(C, device=dev); dgw = torch.empty(d, device=dev); dgb = torch.empty(d, device=dev)
    P = _lib.ptr; st = _lib.stream_ptr
    fwd = lambda: lib.cgcn_head_fwd(st(), n, S, d, C, P(x), P(bn_w), P(bn_b), P(rm), P(rv), P(nbt), 0.1, 1e-5, 1, P(W), P(b), P(tgt), 0.2, P(rng), P(probs), P(loss), P(dpred), P(sm), P(si), P(ws), wsb)
    bwd_full = lambda: lib.cgcn_head_bwd(st(), n, S, d, C, P(x), P(bn_w), P(bn_b), P(sm), P(si), P(W), P(dpred), None, 0.2, P(rng), P(dX), P(dW), P(db), P(dgw), P(dgb), 0, P(ws), wsb)
    bwd_def = lambda: lib.cgcn_head_bwd(st(), n, S, d, C, P(x), P(bn_w), P(bn_b), P(sm), P(si), P(W), P(dpred), None, 0.2, P(rng), None, P(dW), P(db), P(dgw), P(dgb), 0, P(ws), wsb)
    if "--stamps" in sys.argv:  # needs a -DHF_TIMING build (CGCN_EXTRA_FLAGS) loaded through CHROMEGCN_LIB
        import numpy as np
        train = lambda: lib.cgcn_head_train(st(), n, S, d, C, P(x), P(bn_w), P(bn_b), P(rm), P(rv), P(nbt), 0.1, 1e-5, P(W), P(b), P(tgt), 0.2, P(rng), P(probs), P(loss), P(sm), P(si), None, 0, 0, P(ws), wsb)
        for _ in range(5): assert train() == 0
        torch.cuda.synchronize()
        buf = np.zeros(8 * 16, dtype=np.uint64)
        raw = ctypes.CDLL(os.environ["CHROMEGCN_LIB"])
        assert raw.cgcn_debug_hf_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        t = buf.reshape(8, 16).astype(np.int64)
        t0 = t[:, 0].min()
        for b_ in range(8):
            print("wg", b_ * 32, "start+%.2fus" % ((t[b_, 0] - t0) / 100.0), " phases(us):", " ".join("%.2f" % ((t[b_, i + 1] - t[b_, i]) / 100.0) for i in range(12)), " total %.2f" % ((t[b_, 12] - t[b_, 0]) / 100.0))
        print("train_us", round(timeit(train), 1))
        return
    fwd()
    print(json.dumps({"head_fwd_us(4 launches)": round(timeit(fwd), 1), "head_bwd_full_us(3 launches)": round(timeit(bwd_full), 1),
                      "head_bwd_deferred_us(2 launches)": round(timeit(bwd_def), 1)}))

if __name__ == "__main__":
    main()
