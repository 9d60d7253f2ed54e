#!/usr/bin/env python3
"""Stress test of the flag-synchronised LDS ring of k_bwd_rowlocal_ring (GPU box): the same launch, many times, at many
sizes; every launch's dHs / partial sums must be bit-identical to the first launch's (a missed flag, a slot reused too
early or a stale LDS read shows up as a different bit pattern) and match a float64 restatement of the row-local math.
Odd sizes on purpose: last slot not full, fewer slots than workgroups, strand boundary inside a slot, one strand.
The test suite's form of this -- three builds (slowed row team / slowed matrix team), head form through the captured step --
is tests/test_gpu_ring_stress.py; this tool is for long runs by hand (CHROMEGCN_LIB selects a build).
    python tools/ring_stress.py [launches per size, default 300]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chromegcn_amd import _lib, graph as G


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    lib = _lib.load()
    dev = torch.device("cuda")
    d = 128
    P = _lib.ptr; st = _lib.stream_ptr
    bad = 0
    for S, n in [(2, 1), (2, 7), (1, 16), (2, 129), (1, 2049), (2, 2047), (2, 4099), (2, 5776), (1, 16264), (2, 16264), (2, 29910), (2, 70001)]:
        torch.manual_seed(n * 3 + S)
        g = G.upload(G.normalize_graph("none", None, n), dev)
        x, z, h, dxn = (torch.randn(S, n, d, device=dev) for _ in range(4))
        z = torch.tanh(z)
        gate = torch.rand(S, n, device=dev)
        W = torch.randn(d, d, device=dev) / d ** 0.5; wg = torch.randn(d, device=dev) / d ** 0.5
        rs = torch.rand(n, device=dev) + 0.1
        dx, dhs = torch.zeros_like(x), torch.zeros_like(x)
        dW = torch.zeros_like(W); db = torch.zeros(d, device=dev); dwg = torch.zeros(d, device=dev); dcg = torch.zeros(1, device=dev)
        wsb = lib.cgcn_layer_bwd_workspace_bytes(n, S, d); ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)

        def run(ph, with_dhs=True):   # dX == NULL: the row-local launch + the second-stage sum only (no gather)
            return lib.cgcn_debug_layer_bwd_phases(st(), n, S, d, P(g.rowptr), P(g.col), None, P(rs), P(x), P(z), P(h), P(gate), P(W), P(wg),
                                                   P(dxn), None, None, P(dhs) if with_dhs else None, P(dW), P(db), P(dwg), P(dcg), 0, 0.0, None, 0,
                                                   None, P(ws), wsb, ph, None)
        assert run(3) == 0
        torch.cuda.synchronize()
        first = [t.clone() for t in (dhs, dW, db, dwg, dcg)]
        # the form without the dHs product (dHs == NULL: nobody differentiates the layer's input) must give the same sums
        assert run(3, with_dhs=False) == 0
        torch.cuda.synchronize()
        same_without = all(torch.equal(a, b) for a, b in zip(first[1:], (dW, db, dwg, dcg)))
        # float64 restatement
        X, Z, Hh, Gu = (t.double().reshape(S * n, d) for t in (x, z, h, dxn))
        gt = gate.double().reshape(S * n)
        dg = (Gu * (Z - X)).sum(1)
        gamma = gt * (1 - gt) * dg
        dU = (gt[:, None] * Gu + gamma[:, None] * wg.double()[None, :]) * (1 - Z * Z)
        want = {"dHs": (dU * rs.double().repeat(S)[:, None]) @ W.double().T, "dW": Hh.T @ dU, "db": dU.sum(0), "dwg": (gamma[:, None] * Z).sum(0), "dcg": gamma.sum().reshape(1)}
        err = {k: float((a.double().reshape(want[k].shape) - want[k]).abs().max() / want[k].abs().max().clamp_min(1e-30))
               for k, a in zip(("dHs", "dW", "db", "dwg", "dcg"), first)}
        bad_el = torch.zeros((), dtype=torch.int64, device=dev)   # EVERY launch is compared, on the device, without a host sync
        for i in range(reps):
            dhs.fill_(float("nan"))
            assert run(3) == 0
            for a, b in zip(first, (dhs, dW, db, dwg, dcg)):
                bad_el += (a != b).sum()
        diff = int(bad_el)
        ok = diff == 0 and same_without and max(err.values()) < 2e-5
        bad += 0 if ok else 1
        print("S=%d n=%6d  launches %d  differing elements %d  dW-only form identical %s  rel err vs float64 %s  %s"
              % (S, n, reps, diff, same_without, {k: "%.1e" % v for k, v in err.items()}, "ok" if ok else "FAIL"))
        sys.stdout.flush()
    print("RING STRESS", "ok" if bad == 0 else "FAILED (%d sizes)" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
