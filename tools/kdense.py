#!/usr/bin/env python3
"""The row-local forward launch (cgcn_layer_fwd with an H_in: k_layer_dense / k_layer_dense_ring) of several library
builds side by side (tuning tool): Xn, Z, gate and the batch statistics the column-statistics records add up to are
checked against a float64 torch restatement on the same random inputs, every build is timed alone (HIP events) in the
two forms a train step uses (inter-layer dropout; column statistics).
    python tools/kdense.py base=chromegcn_amd/libchromegcn_hip.so old=variants/libcgcn_dold.so ... [--n=16264,29910]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chromegcn_amd import _lib
from tools.kbench import timeit


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ns = [16264, 29910]
    d = 128
    for a in sys.argv[1:]:
        if a.startswith("--n="):
            ns = [int(v) for v in a[4:].split(",")]
        if a.startswith("--d="):
            d = int(a[4:])
    # name=path[:fp32|:split] -- the suffix sets the products form (cgcn_debug_set_products) for that entry's runs
    libs = [(a.split("=")[0], _lib.open_library(os.path.join(ROOT, a.split("=")[1].split(":")[0])),
             {"fp32": 0, "split": 1}.get((a.split("=")[1].split(":") + [""])[1], -1)) for a in args]
    dev = torch.device("cuda")
    S = 2
    P = _lib.ptr; st = _lib.stream_ptr
    for n in ns:
        torch.manual_seed(n)
        x, h = torch.randn(S, n, d, device=dev), torch.randn(S, n, d, device=dev)
        W = torch.randn(d, d, device=dev) / d ** 0.5; b = torch.randn(d, device=dev) * 0.1
        wg = torch.randn(d, device=dev) / d ** 0.5; cg = torch.randn(1, device=dev) * 0.1
        # float64 restatement (SubLayers.py:43-50 on the aggregated H, ChromeModels.py:37-40)
        z64 = torch.tanh(h.double() @ W.double() + b.double())
        g64 = torch.sigmoid(z64 @ wg.double() + cg.double())
        xn64 = (1 - g64)[..., None] * x.double() + g64[..., None] * z64
        r64 = torch.relu(xn64)
        rng = torch.tensor([1234, 5], dtype=torch.int64, device=dev)
        for name, lib, products in libs:
            if hasattr(lib, "cgcn_debug_set_products"):
                lib.cgcn_debug_set_products(products)
            xn, z = torch.empty_like(x), torch.empty_like(x)
            gate = torch.empty(S, n, device=dev)
            rows = ctypes.c_int(0)
            lib.cgcn_debug_set_fwd_split_bytes(0)   # H_in route at every size
            tiles = lib.cgcn_layer_fwd_colstats_plan(n, S, d, _lib.COLSTATS_RECORDS, ctypes.byref(rows))   # records (this tool decodes them)
            cs = torch.zeros(tiles, S, d, 2, device=dev)
            # rowptr / col are not touched on the H_in route, but the entry point checks them for NULL
            dummy = torch.zeros(4, dtype=torch.int32, device=dev)
            def run(stats, drop):
                return lambda: lib.cgcn_layer_fwd(st(), n, S, d, P(dummy), P(dummy), None, None, P(x), P(W), P(b), P(wg), P(cg), P(xn), P(z), None, P(gate),
                                                  0.2 if drop else 0.0, P(rng) if drop else None, 1, P(h), P(cs) if stats else None, rows.value if stats else 0, None)
            rc = run(True, False)()
            assert rc == 0, rc
            torch.cuda.synchronize()
            rel = lambda a, t: float((a.double() - t).abs().max() / t.abs().max())
            cnt = torch.tensor([max(0, min(n, (t + 1) * rows.value) - t * rows.value) for t in range(tiles)], device=dev, dtype=torch.float64)
            mean = (cs[..., 0].double() * cnt[:, None, None]).sum(0) / n
            m2 = (cs[..., 1].double() + cnt[:, None, None] * (cs[..., 0].double() - mean) ** 2).sum(0)
            err = {"Xn": rel(xn, xn64), "Z": rel(z, z64), "gate": rel(gate, g64), "bn_mean": rel(mean, r64.mean(1)),
                   "bn_var": rel(m2 / n, r64.var(1, unbiased=False))}
            first = (xn.clone(), z.clone(), gate.clone(), cs.clone())
            run(True, False)(); torch.cuda.synchronize()
            repro = all(torch.equal(a, b_) for a, b_ in zip(first, (xn, z, gate, cs)))
            t_stats = timeit(run(True, False), reps=100)
            t_drop = timeit(run(False, True), reps=100)
            lib.cgcn_debug_set_fwd_split_bytes(-1)
            print(json.dumps({"n": n, "lib": name, "records": tiles, "nodes_per_record": rows.value, "colstats_form_us": round(t_stats, 2),
                              "dropout_form_us": round(t_drop, 2), "bit_reproducible": repro,
                              "rel_err_vs_float64": {k: float("%.2e" % v) for k, v in err.items()}}))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
