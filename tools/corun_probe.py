#!/usr/bin/env python3
"""Can the parameter-independent aggregation of the NEXT chromosome (H1 = Ahat X0, k_aggregate_sliced) hide under a kernel
of the CURRENT chromosome's step?    For every kernel K of a train step, two HIP graphs of R
repetitions each are captured and replayed:
    serial : K ; agg ; K ; agg ; ...                      (one stream: what the epoch graph does today)
    forked : (K || agg) ; (K || agg) ; ...                (agg on a forked branch, joined before the next repetition)
and the time per repetition is compared with the two kernels' own times.  `hidden` = (serial - forked) / t_agg: the share
of the aggregation that ran under K.  agg works on ANOTHER chromosome's buffers (same size), as it would in the epoch.
    python tools/corun_probe.py [n, default 16264]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chromegcn_amd as C  # noqa: E402
from chromegcn_amd import _lib, graph as G, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16264
    R = 20
    dev = torch.device("cuda")
    lib = _lib.load()
    P, st = _lib.ptr, _lib.stream_ptr
    S, d, Cn = 2, 128, synth.N_LABELS
    torch.manual_seed(0)
    m = C.ChromeGCN(d, d, Cn, 0.2, True, 2).to(dev)
    g = G.upload(G.normalize_graph("hic", synth.contact_graph(n, 250000, 7), n), dev)
    g2 = G.upload(G.normalize_graph("hic", synth.contact_graph(n, 250000, 8), n), dev)   # the next chromosome
    x = torch.randn(S, n, d, device=dev)
    x2, h2 = torch.randn(S, n, d, device=dev), torch.empty(S, n, d, device=dev)
    tgt = (torch.rand(n, Cn, device=dev) < 0.05).float()
    xn, z, h, dx, dhs = (torch.empty_like(x) for _ in range(5))
    gate = torch.empty(S, n, device=dev)
    rng = m._rng_state
    rows = ctypes.c_int(0)
    tiles = lib.cgcn_layer_fwd_colstats_plan(n, S, d, _lib.COLSTATS_RECORDS, ctypes.byref(rows))   # this probe builds its cgcn_head_grad by hand: records mode
    colstats = torch.empty((tiles, S, d, 2), device=dev)
    gc1, w1, gc2, w2, bn, out = m.GC1, m.W1, m.GC2, m.W2, m.batch_norm, m.out
    aux, aux2 = G.aux_ptr(g.col), G.aux_ptr(g2.col)

    def agg():
        return lib.cgcn_spmm(st(), n, n, S, d, P(g2.rowptr), P(g2.col), None, P(g2.row_scale), x2.data_ptr(), h2.data_ptr(), aux2)

    def spmm1():
        return lib.cgcn_spmm(st(), n, n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), x.data_ptr(), h.data_ptr(), aux)

    def dense(last):
        gc, wk = (gc2, w2) if last else (gc1, w1)
        return lib.cgcn_layer_fwd(st(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), x.data_ptr(), gc.weight.data_ptr(), gc.bias.data_ptr(),
                                  wk.weight.data_ptr(), wk.bias.data_ptr(), xn.data_ptr(), z.data_ptr(), None, gate.data_ptr(),
                                  0.0 if last else 0.2, None if last else P(rng), 2 if last else 1, P(h), colstats.data_ptr() if last else None, rows.value if last else 0, aux)
    hws_b = lib.cgcn_head_workspace_bytes(n, S, d, Cn)
    hws = torch.empty(hws_b, dtype=torch.uint8, device=dev)
    probs, loss = torch.empty(n, Cn, device=dev), torch.empty(1, device=dev)
    sm, si = torch.empty(S, d, device=dev), torch.empty(S, d, device=dev)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()

    def head(ph):
        return lib.cgcn_debug_head_train_phases(st(), n, S, d, Cn, xn.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                                None, 0.1, 1e-5, out.weight.data_ptr(), out.bias.data_ptr(), tgt.data_ptr(), 0.2, P(rng), probs.data_ptr(),
                                                loss.data_ptr(), sm.data_ptr(), si.data_ptr(), colstats.data_ptr(), tiles, rows.value, hws.data_ptr(), hws_b, ph)
    o_dym, o_bnc, o_part = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(lib.cgcn_head_workspace_layout(n, S, d, Cn, ctypes.byref(o_dym), ctypes.byref(o_bnc), ctypes.byref(o_part)), "layout")
    one = torch.ones(1, device=dev)
    dW_out, db_out, dbn_w, dbn_b = torch.empty_like(out.weight), torch.empty(Cn, device=dev), torch.empty(d, device=dev), torch.empty(d, device=dev)
    hg = _lib.HeadGrad(hws.data_ptr() + o_dym.value, hws.data_ptr() + o_bnc.value, sm.data_ptr(), si.data_ptr(), bn.weight.data_ptr(), 0.2, P(rng),
                       hws.data_ptr() + o_part.value, lib.cgcn_head_bwd_partials(n), Cn, dW_out.data_ptr(), db_out.data_ptr(), 0, one.data_ptr(),
                       dbn_w.data_ptr(), dbn_b.data_ptr())
    ws_b = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
    ws = torch.empty(ws_b, dtype=torch.uint8, device=dev)
    dW, db, dwg, dcg = torch.empty(d, d, device=dev), torch.empty(d, device=dev), torch.empty(d, device=dev), torch.empty(1, device=dev)
    dxn = torch.randn_like(x) * 1e-6

    def bwd(ph, hm):
        gc, wk = (gc2, w2) if hm else (gc1, w1)
        return lib.cgcn_debug_layer_bwd_phases(st(), n, S, d, P(g.rowptr_t), P(g.col_t), None, P(g.row_scale), x.data_ptr(), z.data_ptr(), h.data_ptr(),
                                               gate.data_ptr(), gc.weight.data_ptr(), wk.weight.data_ptr(), None if hm else dxn.data_ptr(), None, dx.data_ptr(),
                                               dhs.data_ptr(), dW.data_ptr(), db.data_ptr(), dwg.data_ptr(), dcg.data_ptr(), 0, 0.2 if hm else 0.0, P(rng),
                                               1 if hm else 0, ctypes.byref(hg) if hm else None, ws.data_ptr(), ws_b, ph, G.aux_ptr(g.col_t))
    for f in (spmm1, lambda: dense(False), lambda: dense(True), lambda: head(7), lambda: bwd(3, True), lambda: bwd(3, False), agg):
        _lib.check(f(), "setup")
    torch.cuda.synchronize()
    kernels = {"k_layer_dense": lambda: dense(False), "k_layer_dense(colstats)": lambda: dense(True), "k_head_bn_finalize": lambda: head(1),
               "k_head_fused_rs": lambda: head(2), "k_head_train_finish": lambda: head(4), "k_bwd_rowlocal_ring(head)": lambda: bwd(1, True),
               "k_bwd_rowlocal_ring": lambda: bwd(1, False), "k_bwd_sliced": lambda: bwd(2, False), "k_aggregate_sliced(own)": spmm1}

    def graph_time(body):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body(side)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(R):
                body(side)
        for _ in range(3):
            gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / (10 * R) * 1e3   # us per repetition

    def eager_time(body):
        side = torch.cuda.Stream()
        for _ in range(3):
            body(side)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10 * R):
            body(side)
        torch.cuda.current_stream().wait_stream(side)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / (10 * R) * 1e3

    t_agg = graph_time(lambda side: agg())
    print(json.dumps({"n": n, "k_aggregate_sliced(next chromosome)_us": round(t_agg, 2)}))
    for name, K in kernels.items():
        t_k = graph_time(lambda side: K())

        def serial(side):
            K()
            agg()

        def forked(side):
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                agg()
            K()
            cur.wait_stream(side)
        t_s, t_f = graph_time(serial), graph_time(forked)
        # the same two orders issued EAGERLY (two HIP streams = two hardware queues, no graph): separates "the graph runs its
        # branches one after the other" from "the two kernels do not fit on a CU together"
        e_s, e_f = eager_time(serial), eager_time(forked)
        print(json.dumps({"K": name, "K_us": round(t_k, 2), "serial_us": round(t_s, 2), "forked_us": round(t_f, 2),
                          "hidden_share_of_agg": round((t_s - t_f) / t_agg, 3), "eager_serial_us": round(e_s, 2),
                          "eager_two_streams_us": round(e_f, 2), "eager_hidden_share_of_agg": round((e_s - e_f) / t_agg, 3)}))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
