#!/usr/bin/env python3
"""Whole-genome epoch on one GPU (BASELINE.json configs[2] shape, synthetic): 16 train chromosomes, one SGD
step each in the reference's order, then valid/test forward passes and the device metrics.  Prints one JSON
line.  Tuning / reporting tool, not part of the product."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chromegcn_amd as C  # noqa: E402
from chromegcn_amd import metrics as M, synth  # noqa: E402
from chromegcn_amd.finetune import GCNStage  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--hic-like", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = C.ChromeGCN(args.d, args.d, synth.N_LABELS, 0.2, True, args.layers).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    stage = GCNStage(model, opt, "hic", dev)
    splits = {"train": [], "valid": [], "test": []}
    t0 = time.perf_counter()
    for chrom in synth.HG19_LEN:
        feats, hic = synth.synthetic_chromosome(chrom, d=args.d, hic_like=args.hic_like)
        stage.add_chromosome(chrom, feats, hic)
        splits[synth.split_of(chrom)].append(chrom)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    n_train = sum(stage.chroms[c].n for c in splits["train"])
    n_eval = sum(stage.chroms[c].n for c in splits["valid"] + splits["test"])
    times = []
    for e in range(args.epochs + 1):  # epoch 0 = capture
        torch.cuda.synchronize(); t0 = time.perf_counter()
        preds, targets, loss = stage.run_split("train", splits["train"], to_cpu=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        pv, tv, lv = stage.run_split("valid", splits["valid"], to_cpu=False)
        pt, tt, lt = stage.run_split("test", splits["test"], to_cpu=False)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        mv = M.compute_metrics(pv, tv, lv, None, 0.0)
        mt = M.compute_metrics(pt, tt, lt, None, 0.0)
        mtr = M.compute_metrics(preds, targets, loss, None, 0.0)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        times.append((t1 - t0, t2 - t1, t3 - t2))
    tr, ev, me = (min(t[i] for t in times[1:]) for i in range(3))
    out = {"workload": "synthetic GM12878-shaped genome, 22 chromosomes, %d train / %d eval windows, d=%d L=%d C=%d" %
                       (n_train, n_eval, args.d, args.layers, synth.N_LABELS),
           "generator": "hic_like" if args.hic_like else "uniform",
           "load_and_normalise_s": t_load, "capture_epoch_s": sum(times[0]),
           "train_epoch_ms": tr * 1e3, "train_windows_per_s": n_train / tr,
           "eval_ms": ev * 1e3, "eval_windows_per_s": n_eval / ev,
           "metrics_3_splits_ms": me * 1e3, "train_loss": loss, "valid_meanAUC": mv["meanAUC"]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
