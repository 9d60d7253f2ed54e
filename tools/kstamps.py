#!/usr/bin/env python3
"""Phase timestamps of k_bwd_rowlocal (tuning tool).  Needs a -DKT_TIMING build loaded through CHROMEGCN_LIB."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from chromegcn_amd import _lib, graph as G, synth
from tools.kbench import timeit


def main():
    dev = torch.device("cuda"); lib = _lib.load()
    n, pairs, d, S = int(os.environ.get("KS_N", 5776)), 250000, 128, 2
    h = G.normalize_graph("hic", synth.contact_graph(n, pairs, 7, False), n); g = G.upload(h, dev)
    W = torch.randn(d, d, device=dev) / d ** 0.5; wg = torch.randn(d, device=dev) / d ** 0.5
    x = torch.randn(S, n, d, device=dev); z = torch.tanh(torch.randn_like(x)); hh = torch.randn_like(x)
    gate = torch.rand(S, n, device=dev); dxn = torch.randn_like(x); dx = torch.empty_like(x); dhs = torch.empty_like(x)
    dW = torch.empty_like(W); db = torch.empty(d, device=dev); dwg = torch.empty(d, device=dev); dcg = torch.empty(1, device=dev)
    wsb = lib.cgcn_layer_bwd_workspace_bytes(n, S, d); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st, P = _lib.stream_ptr, _lib.ptr
    bwd = lambda: lib.cgcn_layer_bwd(st(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(z), P(hh), P(gate), P(W), P(wg), P(dxn), None, P(dx), P(dhs), P(dW), P(db), P(dwg), P(dcg), 0, 0.0, None, 0, None, P(ws), wsb, None, None, None)
    for _ in range(5): assert bwd() == 0
    torch.cuda.synchronize()
    buf = np.zeros(8 * 16, dtype=np.uint64)
    raw = ctypes.CDLL(os.environ["CHROMEGCN_LIB"])
    assert raw.cgcn_debug_kt_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(8, 16).astype(np.int64)
    t0 = t[t[:, 0] > 0, 0].min()
    names = "prologue loads rowpass sync mfma partial colsums"
    print("phases:", names)
    for b_ in range(8):
        if t[b_, 0] == 0: continue
        print("wg", b_ * 32, "start+%.2fus" % ((t[b_, 0] - t0) / 100.0), " ".join("%.2f" % ((t[b_, i + 1] - t[b_, i]) / 100.0) for i in range(7)), " total %.2f" % ((t[b_, 7] - t[b_, 0]) / 100.0))
    print("layer_bwd_us", round(timeit(bwd), 1))
    b = torch.zeros(d, device=dev); cg = torch.zeros(1, device=dev); xn = torch.empty_like(x)
    fwd = lambda: lib.cgcn_layer_fwd(st(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(W), P(b), P(wg), P(cg), P(xn), P(z), P(hh), P(gate), 0.0, None, 0, None, None, 0, None)
    for _ in range(5): assert fwd() == 0
    torch.cuda.synchronize()
    assert raw.cgcn_debug_kt_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.reshape(8, 16).astype(np.int64)
    t0 = t[t[:, 8] > 0, 8].min()
    print("k_layer_fwd phases: wload gather xres+sync mfma+sync tanh epilogue")
    for b_ in range(8):
        if t[b_, 8] == 0: continue
        print("wg", b_ * int(os.environ.get("KT_STRIDE", 32)), "start+%.2fus" % ((t[b_, 8] - t0) / 100.0), " ".join("%.2f" % ((t[b_, i + 1] - t[b_, i]) / 100.0) for i in range(8, 14)), " total %.2f" % ((t[b_, 14] - t[b_, 8]) / 100.0))
    print("layer_fwd_us", round(timeit(fwd), 1))


if __name__ == "__main__":
    main()
