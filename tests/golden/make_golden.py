#!/usr/bin/env python3
"""
Golden-vector generator.  Runs ONLY in the authoring container: it imports the real
reference (QData/ChromeGCN mounted read-only at /root/reference), drives the hot-path
entry points on seeded inputs and records inputs + outputs as .npz next to this file.
The reference source never leaves that container; only these vectors are committed.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Entry points exercised (reference file:line):
  G1  utils/util_methods.py:146  process_graph            (all four adj_type branches)
  G2  models/SubLayers.py:42     GraphConvolution.forward + gate math of
      models/ChromeModels.py:37-40, with autograd backward for a fixed upstream gradient
  G3  models/ChromeModels.py:34  ChromeGCN.forward (eval; train w/ dropout=0 + one SGD step)
  G4  finetune.py:29-53          the per-chromosome loop, re-driven on CPU (the original
      hard-codes .cuda(), finetune.py:30-36), 2 train epochs + 1 eval pass
  G5  utils/metrics.py:148,168,238,25  fdr / aupr / auroc / mean_average_precision per label
  G6  utils/util_methods.py:183-199    save_feats: encoder outputs regrouped per chromosome (the
      chrom_feature_dict_<split>.pt contract between the window encoder and the GCN stage)
  G7  models/WindowModels.py:9-87 Expecto + models/NonStrandSpecific.py:81-94 strand wrapper: the encoder whose
      x_feat outputs are the GCN stage's node features (config 5).  The weights (tens of MB) are not recorded:
      the reference encoder is built under a recorded torch seed, and the fixture holds the tokens, the outputs
      and a few parameter checksums; the build's own encoder must reproduce them when built under the same seed.
"""
import json
import os
import sys
import warnings

import numpy as np
import scipy
import scipy.sparse as sp
import torch
import torch.nn.functional as F

REF = os.environ.get("CHROMEGCN_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

from models.ChromeModels import ChromeGCN  # noqa: E402  (reference)
from models.SubLayers import GraphConvolution  # noqa: E402  (reference)
from utils import util_methods  # noqa: E402  (reference)

HERE = os.path.dirname(os.path.abspath(__file__))
META = {
    "torch": torch.__version__, "numpy": np.__version__, "scipy": scipy.__version__,
    "reference": "QData/ChromeGCN @ v0 (mounted at /root/reference)",
}


def sym_graph(n, pairs, seed, dense_row=None, isolated=None, self_loop=None):
    """symmetric {0,1} CSR, zero diagonal unless self_loop is given (input-side edge case)."""
    rng = np.random.RandomState(seed)
    m = np.zeros((n, n))
    if n > 1 and pairs > 0:
        i = rng.randint(0, n, pairs); j = rng.randint(0, n, pairs)
        k = i != j
        m[i[k], j[k]] = 1; m[j[k], i[k]] = 1
    if dense_row is not None and n > 1:
        m[dense_row, :] = 1; m[:, dense_row] = 1; m[dense_row, dense_row] = 0
    if isolated is not None:
        m[isolated, :] = 0; m[:, isolated] = 0
    if self_loop is not None:
        m[self_loop, self_loop] = 1
    return sp.csr_matrix(m)


def csr_fields(prefix, a):
    a = sp.csr_matrix(a); a.sort_indices()
    return {prefix + "_indptr": a.indptr.astype(np.int64), prefix + "_indices": a.indices.astype(np.int64),
            prefix + "_data": a.data.astype(np.float64)}


def coo_fields(prefix, t):
    return {prefix + "_row": t._indices()[0].numpy().astype(np.int64),
            prefix + "_col": t._indices()[1].numpy().astype(np.int64),
            prefix + "_val": t._values().numpy().astype(np.float32)}


# ----------------------------------------------------------------------------- G1
def make_g1():
    out = {}
    cases = []
    specs = [
        (1, 0, 11, None, None, None),
        (7, 6, 12, None, 3, 5),        # isolated node 3, self loop on 5 in the INPUT
        (64, 150, 13, 10, 20, 33),     # dense row 10, isolated 20, input self loop 33
        (257, 900, 14, 100, 7, None),  # dense row 100, isolated 7
    ]
    for n, pairs, seed, dense, iso, loop in specs:
        a = sym_graph(n, pairs, seed, dense, iso, loop)
        name = "n%d" % n
        out.update(csr_fields(name + "_in", a))
        # the reference's band builder raises for n < 7 (np.ones of a negative size,
        # utils/util_methods.py:141), so the n=1 case only has the 'hic' and 'none' branches
        for adj_type in (["hic", "none"] if n < 7 else ["hic", "constant", "both", "none"]):
            t = util_methods.process_graph(adj_type, {"c": a.copy()}, n, "c")
            out.update(coo_fields("%s_%s" % (name, adj_type), t))
        cases.append(name)
    out["cases"] = np.array(cases)
    out["meta"] = np.array(json.dumps(META))
    np.savez_compressed(os.path.join(HERE, "g1_process_graph.npz"), **out)


# ----------------------------------------------------------------------------- G2
def rand_layer_params(d, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(d, d, generator=g) / np.sqrt(d) * 1.5
    b = torch.randn(d, generator=g) * 0.2
    wg = torch.randn(1, d, generator=g) / np.sqrt(d) * 2.0
    cg = torch.randn(1, generator=g) * 0.3
    return w, b, wg, cg


def make_g2():
    out = {}
    cases = []
    for (n, d, pairs, adj_type, seed) in [(64, 128, 200, "hic", 21), (257, 128, 1500, "hic", 22),
                                          (130, 256, 500, "hic", 23), (97, 128, 300, "both", 24),
                                          (61, 128, 0, "constant", 25)]:
        name = "n%d_d%d_%s" % (n, d, adj_type)
        a = sym_graph(n, pairs, seed, dense_row=(n // 3 if n > 100 else None))
        adj = util_methods.process_graph(adj_type, {"c": a.copy()}, n, "c")
        gc = GraphConvolution(d, d, bias=True, init="xavier")
        gate = torch.nn.Linear(d, 1)
        w, b, wg, cg = rand_layer_params(d, seed)
        with torch.no_grad():
            gc.weight.copy_(w); gc.bias.copy_(b); gate.weight.copy_(wg); gate.bias.copy_(cg)
        g = torch.Generator().manual_seed(seed + 1000)
        x = torch.randn(n, d, generator=g).requires_grad_(True)
        gup = torch.randn(n, d, generator=g) * 0.1
        # models/ChromeModels.py:37-40 driven with the reference's own modules
        u = gc(x, adj, None)
        z = F.tanh(u)
        gt = F.sigmoid(gate(z))
        xn = (1 - gt) * x + gt * z
        xn.backward(gup)
        out.update(csr_fields(name + "_in", a))
        out.update(coo_fields(name + "_adj", adj))
        for k, v in [("X", x), ("Gup", gup), ("W", w), ("b", b), ("wg", wg), ("cg", cg), ("U", u), ("Z", z),
                     ("g", gt), ("Xn", xn), ("dX", x.grad), ("dW", gc.weight.grad), ("db", gc.bias.grad),
                     ("dwg", gate.weight.grad), ("dcg", gate.bias.grad)]:
            out["%s_%s" % (name, k)] = v.detach().numpy().astype(np.float32)
        cases.append(name)
    out["cases"] = np.array(cases)
    out["meta"] = np.array(json.dumps(META))
    np.savez_compressed(os.path.join(HERE, "g2_gated_layer.npz"), **out)


# ----------------------------------------------------------------------------- G3
def randomize_model(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("GC1.weight") or name.endswith("GC2.weight"):
                p.copy_(torch.randn(p.shape, generator=g) / np.sqrt(p.shape[0]) * 1.5)
            elif "batch_norm.weight" in name:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=g) / np.sqrt(p.shape[1]) * 1.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        model.batch_norm.running_mean.copy_(0.05 * torch.randn(model.batch_norm.running_mean.shape, generator=g))
        model.batch_norm.running_var.copy_(1.0 + 0.2 * torch.rand(model.batch_norm.running_var.shape, generator=g))


def sd_fields(prefix, sd):
    return {"%s_%s" % (prefix, k): v.detach().numpy().copy() for k, v in sd.items()}  # copy: state_dict aliases live params


def make_g3():
    out = {}
    cases = []
    for (n, d, c, layers, pairs, seed) in [(257, 128, 103, 2, 1500, 31), (130, 128, 19, 1, 400, 32),
                                           (90, 256, 11, 2, 300, 33)]:
        name = "n%d_d%d_L%d" % (n, d, layers)
        a = sym_graph(n, pairs, seed)
        adj = util_methods.process_graph("hic", {"c": a.copy()}, n, "c")
        model = ChromeGCN(d, d, c, 0.0, True, layers)
        randomize_model(model, seed)
        out.update(sd_fields(name + "_init", model.state_dict()))
        g = torch.Generator().manual_seed(seed + 1000)
        x_f = torch.randn(n, d, generator=g); x_r = torch.randn(n, d, generator=g)
        tgt = (torch.rand(n, c, generator=g) < 0.1).float()
        out.update(csr_fields(name + "_in", a))
        out[name + "_xf"] = x_f.numpy(); out[name + "_xr"] = x_r.numpy(); out[name + "_tgt"] = tgt.numpy()
        # eval forward (ChromeModels.py:34-52)
        model.eval()
        with torch.no_grad():
            _, lo_f, (g1, g2), _ = model(x_f, adj, None)
            _, lo_r, _, _ = model(x_r, adj, None)
        out[name + "_eval_logits_f"] = lo_f.numpy(); out[name + "_eval_logits_r"] = lo_r.numpy()
        out[name + "_eval_g1"] = g1.numpy()
        if g2 is not None:
            out[name + "_eval_g2"] = g2.numpy()
        # train step, dropout = 0  (finetune.py:38-49; optimizer utils/util_methods.py:14-19)
        model.train()
        opt = torch.optim.SGD(model.parameters(), lr=0.25, weight_decay=1e-6, momentum=0.9)
        xf = x_f.clone().requires_grad_(True); xr = x_r.clone().requires_grad_(True)
        opt.zero_grad()
        _, pf, _, _ = model(xf, adj, None)
        _, pr, _, _ = model(xr, adj, None)
        pred = (pf + pr) / 2
        loss = F.binary_cross_entropy_with_logits(pred, tgt)
        loss.backward()
        out[name + "_train_loss"] = np.array(loss.item(), dtype=np.float64)
        out[name + "_train_pred"] = pred.detach().numpy()
        out[name + "_train_dxf"] = xf.grad.numpy().copy(); out[name + "_train_dxr"] = xr.grad.numpy().copy()
        for k, p in model.named_parameters():
            out["%s_grad_%s" % (name, k)] = p.grad.numpy().copy()
        opt.step()
        out.update(sd_fields(name + "_post", model.state_dict()))
        # second step (exercises the momentum buffer)
        opt.zero_grad()
        _, pf, _, _ = model(xf, adj, None); _, pr, _, _ = model(xr, adj, None)
        loss2 = F.binary_cross_entropy_with_logits((pf + pr) / 2, tgt)
        loss2.backward(); opt.step()
        out[name + "_train_loss2"] = np.array(loss2.item(), dtype=np.float64)
        out.update(sd_fields(name + "_post2", model.state_dict()))
        cases.append(name)
    out["cases"] = np.array(cases)
    out["meta"] = np.array(json.dumps(META))
    np.savez_compressed(os.path.join(HERE, "g3_model.npz"), **out)


# ----------------------------------------------------------------------------- G4
def drive_loop(model, feats, graphs, optimizer, split, adj_type):
    """finetune.py:29-53 on CPU: the reference's own model + process_graph, its loop order."""
    model.train() if split == "train" else model.eval()
    preds = []; losses = []
    for chrom in feats:
        x_f = feats[chrom]["forward"].clone().requires_grad_(True)
        x_r = feats[chrom]["backward"].clone().requires_grad_(True)
        tgt = feats[chrom]["target"]
        adj = util_methods.process_graph(adj_type, graphs, x_f.size(0), chrom)
        if split == "train":
            optimizer.zero_grad()
        _, pf, _, _ = model(x_f, adj, None)
        _, pr, _, _ = model(x_r, adj, None)
        pred = (pf + pr) / 2
        loss = F.binary_cross_entropy_with_logits(pred, tgt.float())
        if split == "train":
            loss.backward(); optimizer.step()
        losses.append(loss.item())
        preds.append(F.sigmoid(pred).detach())
    return torch.cat(preds, 0), losses


def make_g4():
    out = {}
    d, c, layers, seed = 128, 19, 2, 41
    sizes = {"chr5": 150, "chr9": 97, "chr20": 211}
    feats, graphs = {}, {}
    g = torch.Generator().manual_seed(seed)
    for i, (chrom, n) in enumerate(sizes.items()):
        graphs[chrom] = sym_graph(n, 6 * n, seed + i)
        feats[chrom] = {"forward": torch.randn(n, d, generator=g), "backward": torch.randn(n, d, generator=g),
                        "target": (torch.rand(n, c, generator=g) < 0.15).float()}
        out.update(csr_fields(chrom + "_in", graphs[chrom]))
        out[chrom + "_xf"] = feats[chrom]["forward"].numpy(); out[chrom + "_xr"] = feats[chrom]["backward"].numpy()
        out[chrom + "_tgt"] = feats[chrom]["target"].numpy()
    model = ChromeGCN(d, d, c, 0.0, True, layers)
    randomize_model(model, seed)
    out.update(sd_fields("init", model.state_dict()))
    opt = torch.optim.SGD(model.parameters(), lr=0.25, weight_decay=1e-6, momentum=0.9)
    trace = []
    for epoch in range(2):
        p, l = drive_loop(model, feats, graphs, opt, "train", "hic")
        trace += l
        out["train_preds_e%d" % epoch] = p.numpy()
    p, l = drive_loop(model, feats, graphs, opt, "valid", "hic")
    out["eval_preds"] = p.numpy(); out["eval_losses"] = np.array(l, dtype=np.float64)
    out["train_losses"] = np.array(trace, dtype=np.float64)
    out.update(sd_fields("final", model.state_dict()))
    out["chroms"] = np.array(list(sizes.keys()))
    out["meta"] = np.array(json.dumps(META))
    np.savez_compressed(os.path.join(HERE, "g4_finetune_loop.npz"), **out)


# ----------------------------------------------------------------------------- G5
def make_g5():
    """utils/metrics.py auroc / aupr / fdr / mean_average_precision (the per-label arrays compute_metrics
    aggregates, utils/evals.py:89-92) on scores with ties, a label without positives and one without negatives."""
    from utils import metrics as ref_metrics
    rng = np.random.RandomState(51)
    n, c = 3000, 9
    tg = (rng.rand(n, c) < np.linspace(0.02, 0.6, c)).astype(np.float64)
    tg[:, 3] = 0.0
    tg[:, 6] = 1.0
    pr = rng.rand(n, c) * 0.6 + 0.4 * tg * rng.rand(n, c)
    pr[:, 1] = np.round(pr[:, 1], 2)      # heavy ties
    pr[:, 2] = np.round(pr[:, 2], 1)
    pr[:, 8] = 0.5                        # all scores equal
    pr = pr.astype(np.float32)
    out = {"targets": tg.astype(np.float32), "preds": pr}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        per_label = {"auroc": [], "aupr": [], "fdr": [], "ap": []}
        for j in range(c):  # the reference helpers drop undefined labels from their arrays; record per label
            t1, p1 = tg[:, j:j + 1], pr[:, j:j + 1]
            a = ref_metrics.auroc(t1, p1)[3]
            per_label["auroc"].append(a[0] if len(a) else np.nan)
            a = ref_metrics.aupr(t1, p1)[3]
            per_label["aupr"].append(a[0] if len(a) else np.nan)
            a = ref_metrics.fdr(t1, p1)[3]
            per_label["fdr"].append(a[0] if len(a) else np.nan)
            try:
                per_label["ap"].append(ref_metrics.mean_average_precision(t1[:, 0], p1[:, 0]))
            except Exception:
                per_label["ap"].append(np.nan)
    for k, v in per_label.items():
        out["ref_" + k] = np.array(v, dtype=np.float64)
    out["meta"] = np.array(json.dumps(META))
    np.savez_compressed(os.path.join(HERE, "g5_metrics.npz"), **out)


def make_g6():
    """save_feats on interleaved chromosomes: rows keep their order of appearance inside each chromosome, the
    dict iterates chromosomes in order of first appearance."""
    import tempfile
    g = torch.Generator().manual_seed(66)
    n, d, c = 157, 16, 7
    chroms = ["chr8", "chr21", "chrX", "chr1"]
    which = torch.randint(0, len(chroms), (n,), generator=g).tolist()
    locs = [(chroms[w], 1000 * i, 1000 * i + 1000) for i, w in enumerate(which)]
    x_f = torch.randn(n, d, generator=g); x_r = torch.randn(n, d, generator=g)
    targs = (torch.rand(n, c, generator=g) < 0.3).float()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "run"))  # save_feats writes into <model_name before '.finetune'>/
        util_methods.save_feats(os.path.join(tmp, "run.finetune.x"), "train", targs, locs, x_f, x_r)
        saved = torch.load(os.path.join(tmp, "run", "chrom_feature_dict_train.pt"))
    out = {"meta": np.array(json.dumps(META)), "chrom_of_row": np.array([l[0] for l in locs]), "x_f": x_f.numpy(),
           "x_r": x_r.numpy(), "targs": targs.numpy(), "order": np.array(list(saved.keys()))}
    for ch, v in saved.items():
        for k in ("forward", "backward", "target"):
            out["%s_%s" % (ch, k)] = v[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g6_save_feats.npz"), **out)


def make_g7():
    from models.WindowModels import Expecto  # reference
    from models.NonStrandSpecific import GraphNonStrandSpecific  # reference
    seed, L, C, B = 77, 600, 11, 4
    torch.manual_seed(seed)
    enc = Expecto(C, L)
    g = torch.Generator().manual_seed(78)
    with torch.no_grad():  # non-trivial BatchNorm statistics, drawn AFTER construction from a separate generator
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    enc.eval()
    tokens = torch.randint(0, 5, (B, L), generator=g)
    src_dict = {"a": 0, "c": 1, "g": 2, "t": 3, "n": 4}
    with torch.no_grad():
        x_f, x_r, y, _, _ = GraphNonStrandSpecific(enc)(tokens, src_dict)
    sd = enc.state_dict()
    out = {"meta": np.array(json.dumps(dict(META, seed=seed, bn_seed=78, seq_length=L, nclass=C))),
           "tokens": tokens.numpy(), "x_f": x_f.numpy(), "x_r": x_r.numpy(), "logits": y.numpy(),
           "keys": np.array(list(sd.keys())),
           "param_sums": np.array([float(v.double().sum()) for v in sd.values()]),
           "param_numel": np.array([v.numel() for v in sd.values()])}
    np.savez_compressed(os.path.join(HERE, "g7_encoder.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(1)  # fixed reduction order for the recorded values
    if sys.argv[1:]:
        for name in sys.argv[1:]:
            globals()["make_" + name]()
    else:
        make_g1(); make_g2(); make_g3(); make_g4(); make_g5(); make_g6(); make_g7()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
