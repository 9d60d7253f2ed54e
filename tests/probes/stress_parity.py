#!/usr/bin/env python3
"""Randomised parity sweep of the fused layer (forward + backward) against the oracle's numpy restatement: random
sizes (1 .. 6000 nodes, ragged tiles), densities, adjacency types (implicit / explicit values), strands and widths.
A one-off confidence tool for the GPU box (python tests/probes/stress_parity.py [cases] [seed]); the fixed cases live in tests/."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from chromegcn_amd import graph as G, ops
from oracle import chromegcn_oracle as O  # checker only

DEV = "cuda"
TOL = dict(rtol=1e-4, atol=1e-4)


def one(rng, case):
    n = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, rng.randint(1, 400), rng.randint(400, 6000)]))
    S = int(rng.choice([1, 2])); d = int(rng.choice([128, 128, 256]))
    adj = str(rng.choice(["hic", "hic", "both", "constant", "none"]))
    pairs = int(rng.choice([0, n // 2, 3 * n, 20 * n, min(60 * n, n * n // 3)]))
    a = O.random_symmetric_graph(n, pairs, int(rng.randint(1 << 30))) if adj in ("hic", "both") else None
    if a is not None and n > 40 and rng.rand() < 0.3:   # a hub row
        a = a.tolil(); hub = int(rng.randint(n)); a[hub, :] = 1; a[:, hub] = 1; a[hub, hub] = 0; a = a.tocsr()
    h = G.normalize_graph(adj, a, n)
    g = G.upload(h, DEV)
    W = (rng.randn(d, d) / np.sqrt(d) * 1.5).astype(np.float32); b = (rng.randn(d) * 0.2).astype(np.float32)
    wg = (rng.randn(d) / np.sqrt(d) * 2).astype(np.float32); cg = np.float32(rng.randn() * 0.3)
    x = rng.randn(S, n, d).astype(np.float32)
    gup = (rng.randn(S, n, d) * 0.1).astype(np.float32); ggate = (rng.randn(S, n) * 0.1).astype(np.float32)
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(DEV)
    t = {k: dev(v).requires_grad_(True) for k, v in dict(x=x, W=W, b=b, wg=wg.reshape(1, d), cg=np.array([cg])).items()}
    xn, gate = ops.gated_layer(t["x"], t["W"], t["b"], t["wg"], t["cg"], g)
    (xn * dev(gup)).sum().add((gate * dev(ggate)).sum()).backward()
    sp = h.to_scipy()
    acc = {k: 0.0 for k in ["dW", "db", "dwg", "dcg"]}
    tag = "case %d n=%d S=%d d=%d adj=%s nnz=%d" % (case, n, S, d, adj, h.nnz)
    for s in range(S):
        f = O.layer_forward_np(sp, x[s], W, b, wg, float(cg))
        np.testing.assert_allclose(xn[s].detach().cpu().numpy(), f["Xn"], err_msg=tag, **TOL)
        np.testing.assert_allclose(gate[s].detach().cpu().numpy(), f["g"], err_msg=tag, **TOL)
        bw = O.layer_backward_np(sp, x[s], W, wg, f["Z"], f["g"], gup[s], ggate[s])
        np.testing.assert_allclose(t["x"].grad[s].cpu().numpy(), bw["dX"], err_msg=tag + " dX", **TOL)
        for k in acc:
            acc[k] = acc[k] + bw[k]
    scale = max(1.0, float(np.abs(acc["dW"]).max()))
    np.testing.assert_allclose(t["W"].grad.cpu().numpy(), acc["dW"], rtol=1e-4, atol=1e-4 * scale, err_msg=tag + " dW")
    np.testing.assert_allclose(t["b"].grad.cpu().numpy(), acc["db"], rtol=1e-4, atol=1e-4 * scale, err_msg=tag + " db")
    np.testing.assert_allclose(t["wg"].grad.cpu().numpy().ravel(), acc["dwg"], rtol=1e-4, atol=1e-4 * scale, err_msg=tag + " dwg")
    np.testing.assert_allclose(t["cg"].grad.cpu().numpy().ravel()[0], acc["dcg"], rtol=1e-4, atol=1e-4 * scale, err_msg=tag + " dcg")
    return tag


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0 = time.time()
    for c in range(cases):
        tag = one(rng, c)
        if c % 10 == 0:
            print(tag, "ok")
    print("%d cases ok in %.1f s" % (cases, time.time() - t0))


if __name__ == "__main__":
    main()
