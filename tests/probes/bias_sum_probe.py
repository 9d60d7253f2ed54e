"""Where does the error of the gate-bias gradient (dW2.bias = sum_i gamma_i) come from?  

Splits the HIP-vs-float64 error of the last layer's bias-type sums into
  (a) kernel error : cgcn_layer_bwd's db / dwg / dcg  vs  the same sums recomputed in float64 from the SAME fp32
                     inputs the kernel read (X, Z, gate, dXn) -- row math + summation inside k_bwd_rowlocal / reduce_slab
  (b) input error  : those float64-from-HIP-inputs sums  vs  the all-float64 oracle -- error the sums inherit from the
                     fp32 forward (activations, BatchNorm statistics, dL/dXn), amplified by the cancellation of the sum
Usage (GPU box): python tests/probes/bias_sum_probe.py [chr21|chr1]"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import chromegcn_amd as C  # noqa: E402
from chromegcn_amd import graph as G, synth, torch_ops  # noqa: E402,F401
from oracle import chromegcn_oracle as O  # noqa: E402  (checker)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "chr21"
    gen = sys.argv[2] if len(sys.argv) > 2 else ("hic_like" if which == "chr1" else "uniform")
    n = synth.chrom_nodes(which)
    seed = {"chr21": 21, "chr1": 1}[which] + (5 if gen == "hub" else 0)
    d, NC = 128, 103
    feats = synth.chrom_features(n, d, NC, 1000 + seed)
    hic = synth.contact_graph(n, 250000, seed, gen)
    torch.manual_seed(seed)
    orc = O.GatedGCNOracle(d, NC, 0.0, 2)
    with torch.no_grad():
        for k in (1, 2):
            getattr(orc, "GC%d" % k).weight.mul_(40)
            getattr(orc, "W%d" % k).weight.mul_(3)
    orc64 = copy.deepcopy(orc).double()
    # ---- float64 truth (reference op order), parameter gradients of one step
    f64 = {k: v.double() for k, v in feats.items()}
    adj64 = O.process_graph("hic", {"c": hic}, n, "c").double()
    orc64.train()
    xs = [f64["forward"], f64["backward"]]
    preds = [orc64(x, adj64, None)[1] for x in xs]
    loss64 = torch.nn.functional.binary_cross_entropy_with_logits((preds[0] + preds[1]) / 2, f64["target"])
    loss64.backward()
    truth = {k: p.grad.numpy().copy() for k, p in orc64.named_parameters()}

    dev = "cuda"
    g = G.upload(G.normalize_graph("hic", hic, n), dev)
    P = {k: v.to(dev) for k, v in orc.state_dict().items()}
    x0 = torch.stack([feats["forward"], feats["backward"]]).to(dev)
    tgt = feats["target"].to(dev)
    ga = (g.rowptr, g.col, g.val, g.row_scale, g.rowptr_t, g.col_t, g.val_t)
    xn1, g1, z1, h1 = torch.ops.chromegcn.gated_layer(x0, P["GC1.weight"], P["GC1.bias"], P["W1.weight"], P["W1.bias"], *ga, 0.0, 0.0, None, 1)
    xn2, g2, z2, h2 = torch.ops.chromegcn.gated_layer(xn1, P["GC2.weight"], P["GC2.bias"], P["W2.weight"], P["W2.bias"], *ga, 0.0, 0.0, None, 2)

    def head64(xn):   # dL/dXn2 in float64 from a given Xn2 (torch autograd on the device, float64)
        xn = xn.double().requires_grad_(True)
        ys = []
        for s in range(2):
            r = torch.relu(xn[s])
            mu, var = r.mean(0), r.var(0, unbiased=False)
            y = (r - mu) / torch.sqrt(var + 1e-5) * P["batch_norm.weight"].double() + P["batch_norm.bias"].double()
            ys.append(y @ P["out.weight"].double().t() + P["out.bias"].double())
        loss = torch.nn.functional.binary_cross_entropy_with_logits((ys[0] + ys[1]) / 2, tgt.double())
        loss.backward()
        return xn.grad

    G2 = head64(xn2)           # exact d loss / d Xn2 for the HIP forward's Xn2
    G2f = G2.float()
    dx, dw, db, dwg, dcg, dhs = torch.ops.chromegcn.gated_layer_backward(
        G2f, None, xn1, z2, h2, g2, P["GC2.weight"], P["W2.weight"], g.rowptr_t, g.col_t, g.val_t, g.row_scale, 0.0, None, 2, True)

    def sums64(Gup, x, z, gt, wg):   # Appendix A row math in float64 on the given inputs
        Gup, x, z, gt, wg = Gup.double(), x.double(), z.double(), gt.double(), wg.double().view(-1)
        dg = (Gup * (z - x)).sum(-1)
        gamma = gt * (1 - gt) * dg
        du = (gt.unsqueeze(-1) * Gup + gamma.unsqueeze(-1) * wg) * (1 - z * z)
        return du.sum((0, 1)).cpu().numpy(), (gamma.unsqueeze(-1) * z).sum((0, 1)).cpu().numpy(), gamma.sum().cpu().numpy(), gamma

    db64, dwg64, dcg64, gamma = sums64(G2f, xn1, z2, g2, P["W2.weight"])
    print("[%s] n=%d  layer 2 (last layer) bias-type sums" % (which, n))
    print("  cancellation of dcg: |sum gamma| / sum |gamma| = %.3e" % (abs(float(gamma.sum())) / float(gamma.abs().sum())))
    for nm, hipv, f64v, tk in (("GC2.bias", db, db64, "GC2.bias"), ("W2.weight", dwg, dwg64, "W2.weight"), ("W2.bias", dcg, dcg64, "W2.bias")):
        a = rel(hipv.cpu().numpy().reshape(-1), np.asarray(f64v).reshape(-1))
        b = rel(np.asarray(f64v).reshape(-1), truth[tk].reshape(-1))
        c = rel(hipv.cpu().numpy().reshape(-1), truth[tk].reshape(-1))
        print("  d%-10s kernel (HIP vs f64 on the HIP inputs) %.2e | inputs (f64 on the HIP inputs vs all-f64) %.2e | total %.2e" % (nm, a, b, c))
    # which upstream quantity carries the input error: replace one input at a time by its float64-oracle value
    with torch.no_grad():
        o = orc64
        a64 = adj64
        acts = []
        for x in xs:
            z1o = torch.tanh(o.GC1(x, a64)); g1o = torch.sigmoid(o.W1(z1o)); x1o = (1 - g1o) * x + g1o * z1o
            z2o = torch.tanh(o.GC2(x1o, a64)); g2o = torch.sigmoid(o.W2(z2o)); x2o = (1 - g2o) * x1o + g2o * z2o
            acts.append((x1o, z2o, g2o.view(-1), x2o))
        X1o = torch.stack([a[0] for a in acts]).to(dev); Z2o = torch.stack([a[1] for a in acts]).to(dev)
        G2o_ = torch.stack([a[2] for a in acts]).to(dev); X2o = torch.stack([a[3] for a in acts]).to(dev)
    Gexact = head64(X2o)
    print("  max |HIP - f64| : Xn1 %.2e  Z2 %.2e  gate2 %.2e  Xn2 %.2e  dL/dXn2(rel) %.2e" % (
        float((xn1.double() - X1o).abs().max()), float((z2.double() - Z2o).abs().max()), float((g2.double() - G2o_).abs().max()),
        float((xn2.double() - X2o).abs().max()), rel(G2.cpu().numpy(), Gexact.cpu().numpy())))
    for label, args in (("all f64 inputs", (Gexact, X1o, Z2o, G2o_)), ("HIP dXn only", (G2, X1o, Z2o, G2o_)), ("HIP Xn1 only", (Gexact, xn1, Z2o, G2o_)),
                        ("HIP Z2 only", (Gexact, X1o, z2, G2o_)), ("HIP gate only", (Gexact, X1o, Z2o, g2))):
        _, _, dcgv, _ = sums64(args[0], args[1], args[2], args[3], P["W2.weight"])
        print("  dW2.bias with %-16s: rel err vs all-f64 truth %.2e" % (label, rel(np.asarray(dcgv).reshape(-1), truth["W2.bias"].reshape(-1))))


    # ---- layer 1: its bias-type sums inherit dL/dXn1 = the dX of layer 2's backward (gather over Ahat^T included)
    A64 = torch.sparse_csr_tensor(g.rowptr.long(), g.col.long(), torch.ones(g.col.numel(), dtype=torch.float64, device=dev), size=(n, n))
    rs64 = g.row_scale.double()
    gam2 = g2.double() * (1 - g2.double()) * (G2f.double() * (z2.double() - xn1.double())).sum(-1)
    du2 = (g2.double().unsqueeze(-1) * G2f.double() + gam2.unsqueeze(-1) * P["W2.weight"].double().view(-1)) * (1 - z2.double() ** 2)
    dhs64 = (du2 @ P["GC2.weight"].double().t()) * rs64.view(1, -1, 1)
    dx64 = torch.stack([(1 - g2[s].double()).unsqueeze(-1) * G2f[s].double() + torch.sparse.mm(A64.t().to_sparse_csr(), dhs64[s]) for s in range(2)])
    deg = (g.rowptr[1:] - g.rowptr[:-1]).cpu().numpy()
    e = (dx.double() - dx64).abs().amax((0, 2)).cpu().numpy()
    worst = np.argsort(-e)[:5]
    print("  layer-2 backward dX (= dL/dXn1): HIP vs f64 on the HIP inputs, scale-relative %.2e; rows with the largest error (row, degree): %s; max degree %d" % (
        rel(dx.cpu().numpy(), dx64.cpu().numpy()), [(int(r), int(deg[r])) for r in worst], int(deg.max())))
    print("  dHs (MFMA, fp32) vs f64: scale-relative %.2e" % rel(dhs.cpu().numpy(), dhs64.cpu().numpy()))
    _, dw1, db1, dwg1, dcg1, _ = torch.ops.chromegcn.gated_layer_backward(
        dx, None, x0, z1, h1, g1, P["GC1.weight"], P["W1.weight"], g.rowptr_t, g.col_t, g.val_t, g.row_scale, 0.0, None, 1, True)
    for label, Gup in (("HIP dXn1 (kernel error)", dx), ("f64 dXn1 (what layer 2's dX costs)", dx64)):
        b64, w64, c64, gam = sums64(Gup, x0, z1, g1, P["W1.weight"])
        print("  layer 1 with %-36s: dGC1.bias %.2e  dW1.weight %.2e  dW1.bias %.2e   (HIP kernel outputs vs these f64 sums; cancellation %.2e)" % (
            label, rel(db1.cpu().numpy(), b64), rel(dwg1.cpu().numpy(), w64), rel(dcg1.cpu().numpy().reshape(-1), np.asarray(c64).reshape(-1)),
            abs(float(gam.sum())) / float(gam.abs().sum())))
    print("  layer 1 all-f64 truth vs HIP outputs: dGC1.bias %.2e  dW1.weight %.2e  dW1.bias %.2e" % (
        rel(db1.cpu().numpy(), truth["GC1.bias"]), rel(dwg1.cpu().numpy().reshape(-1), truth["W1.weight"].reshape(-1)), rel(dcg1.cpu().numpy().reshape(-1), truth["W1.bias"].reshape(-1))))


if __name__ == "__main__":
    main()
