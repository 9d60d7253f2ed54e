#!/usr/bin/env python3
"""Row-local backward launch (phase 1 of cgcn_debug_layer_bwd_phases) of several library builds side by side (tuning
tool): every variant is checked against the first one on the same random inputs (dHs, dW, db, dwg, dcg after the full
call, max |a - b| / max |b|) and timed alone (HIP events, 100 launches).
    python tools/kring.py base=chromegcn_amd/libchromegcn_hip.so old=variants/libcgcn_old.so ... [--n 5776,16264,29910]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chromegcn_amd import _lib, graph as G, synth
from tools.kbench import timeit


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ns = [5776, 16264, 29910]
    for a in sys.argv[1:]:
        if a.startswith("--n="):
            ns = [int(v) for v in a[4:].split(",")]
    reps = 100
    libs = []
    for a in args:   # name=path[:fp32|:split] -- the suffix sets the products form (cgcn_debug_set_products) for that entry
        name, path = a.split("=")
        path, _, form = path.partition(":")
        libs.append((name, _lib.open_library(os.path.join(ROOT, path)), {"fp32": 0, "split": 1}.get(form, -1)))
    dev = torch.device("cuda")
    d, S = 128, 2
    P = _lib.ptr; st = _lib.stream_ptr
    for n in ns:
        torch.manual_seed(n)
        g = G.upload(G.normalize_graph("hic", synth.contact_graph(n, 250000, 7), n), dev)
        x, z, h, dxn = (torch.randn(S, n, d, device=dev) for _ in range(4))
        z = torch.tanh(z)
        gate = torch.rand(S, n, device=dev)
        W = torch.randn(d, d, device=dev) / d ** 0.5; wg = torch.randn(d, device=dev) / d ** 0.5
        ref = None
        for name, lib, products in libs:
            if hasattr(lib, "cgcn_debug_set_products"):
                lib.cgcn_debug_set_products(products)
            dx, dhs = torch.zeros_like(x), torch.zeros_like(x)
            dW = torch.zeros_like(W); db = torch.zeros(d, device=dev); dwg = torch.zeros(d, device=dev); dcg = torch.zeros(1, device=dev)
            wsb = lib.cgcn_layer_bwd_workspace_bytes(n, S, d); ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
            def run(ph):
                return lambda: lib.cgcn_debug_layer_bwd_phases(st(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(z), P(h), P(gate), P(W), P(wg),
                                                               P(dxn), None, P(dx), P(dhs), P(dW), P(db), P(dwg), P(dcg), 0, 0.0, None, 0, None, P(ws), wsb, ph, G.aux_ptr(g.col))
            assert run(3)() == 0
            torch.cuda.synchronize()
            out = {"dHs": dhs.clone(), "dW": dW.clone(), "db": db.clone(), "dwg": dwg.clone(), "dcg": dcg.clone(), "dX": dx.clone()}
            # run to run reproducibility of the whole call
            assert run(3)() == 0
            torch.cuda.synchronize()
            repro = all(torch.equal(out[k], v) for k, v in {"dHs": dhs, "dW": dW, "db": db, "dwg": dwg, "dcg": dcg, "dX": dx}.items())
            err = None
            if ref is None:
                ref = out
            else:
                err = {k: float((out[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-30)) for k in out}
            t = timeit(run(1), reps=reps)
            print(json.dumps({"n": n, "lib": name, "rowlocal_us": round(t, 2), "bit_reproducible": repro,
                              "rel_err_vs_first": None if err is None else {k: float("%.2e" % v) for k, v in err.items()}}))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
