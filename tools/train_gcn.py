#!/usr/bin/env python3
"""End-to-end GCN-stage training on the synthetic GM12878-shaped genome through the reference-shaped driver
(chromegcn_amd.runner.run_model): train / valid / test every epoch, device metrics, checkpoint + .log files."""
import argparse
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chromegcn_amd as C  # noqa: E402
from chromegcn_amd import runner, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--chroms", default="chr19,chr20,chr22,chr17,chr21", help="comma list; split by data/create_data.py:44-45")
    ap.add_argument("--out", default="gpurun_out/train_gcn")
    ap.add_argument("--lr", type=float, default=0.25)
    args = ap.parse_args()
    dev = torch.device("cuda")
    data = {"train": {}, "valid": {}, "test": {}}
    graphs = {"train": {}, "valid": {}, "test": {}}
    for c in args.chroms.split(","):
        feats, hic = synth.synthetic_chromosome(c)
        sp = synth.split_of(c)
        data[sp][c] = feats
        graphs[sp][c] = hic
    torch.manual_seed(0)
    model = C.ChromeGCN(128, 128, synth.N_LABELS, 0.2, True, 2).to(dev)
    optim = torch.optim.SGD(model.parameters(), lr=args.lr, momentum=0.9, weight_decay=1e-6)
    opt = types.SimpleNamespace(epochs=args.epochs, adj_type="hic", model_name=args.out, lr_decay2=0, load_gcn=False,
                                test_only=False, hip_graphs=True)
    t0 = time.time()
    hist = runner.run_model(None, model, data["train"], data["valid"], data["test"], None, optim, None, opt, None, graphs=graphs)
    print("total %.2f s for %d epochs; files in %s: %s" % (time.time() - t0, args.epochs, args.out, sorted(os.listdir(args.out))))


if __name__ == "__main__":
    main()
