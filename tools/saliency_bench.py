#!/usr/bin/env python3
"""The adjacency saliency (scripts/visualize.py:29-55 on the CSR pattern) of a chr21-size and a chr1-size
chromosome, 5 times each: the program tools/final_profiles.sh runs under rocprofv3 --kernel-trace --stats for
profiles/r0N_saliency_kernel_stats.csv (k_sddmm, k_saliency_rows and the forward / backward kernels around them)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import chromegcn_amd as C  # noqa: E402
from chromegcn_amd import synth  # noqa: E402
from chromegcn_amd.saliency import adjacency_saliency  # noqa: E402


def main():
    dev = "cuda"
    torch.manual_seed(0)
    model = C.ChromeGCN(128, 128, synth.N_LABELS, 0.0, True, 2).to(dev).eval()
    for name in ("chr21", "chr1"):
        feats, hic = synth.synthetic_chromosome(name, d=128)
        n = feats["forward"].shape[0]
        g = C.process_graph("hic", {name: hic}, n, name, device=dev)
        xf, xr, t = feats["forward"].to(dev), feats["backward"].to(dev), feats["target"].float().to(dev)
        adjacency_saliency(model, xf, xr, g, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            _, sal = adjacency_saliency(model, xf, xr, g, t)
        torch.cuda.synchronize()
        print("%s n=%d nnz=%d saliency %.3f ms per call, max %.3f" % (name, n, g.nnz, (time.perf_counter() - t0) / 5 * 1e3, float(sal.max())))


if __name__ == "__main__":
    main()
