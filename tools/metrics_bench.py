#!/usr/bin/env python3
"""Device multi-label metrics on the train split's shape: ms per call, and (under rocprofv3 --kernel-trace
--stats) the kernels behind it.  Tuning tool."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from chromegcn_amd import metrics as M


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 242908
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 103
    torch.manual_seed(0)
    dev = torch.device("cuda")
    probs = torch.sigmoid(torch.randn(n, C, device=dev))
    targ = (torch.rand(n, C, device=dev) < 0.05).float()
    for _ in range(3):
        out = M.compute_metrics(probs, targ, 0.0, None, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = M.compute_metrics(probs, targ, 0.0, None, 0.0)
    torch.cuda.synchronize()
    print(json.dumps({"n": n, "C": C, "ms_per_call": (time.perf_counter() - t0) / 10 * 1e3, "meanAUC": out["meanAUC"]}))


if __name__ == "__main__":
    main()
