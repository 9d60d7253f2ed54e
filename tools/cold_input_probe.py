#!/usr/bin/env python3
"""Upper bound for prefetching the next chromosome's inputs (tuning tool): an epoch over 16 chromosomes of the SAME shape
(n = 15 182, the genome's mean) whose features / graph / targets are 16 distinct device buffers (each last touched an epoch
ago: cold beyond L2, like the genome) against the same epoch with all 16 pointing at ONE set of buffers (touched 0.27 ms
ago: Infinity-Cache resident).  Everything else -- kernels, order, per-chromosome workspaces -- is the same."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import chromegcn_amd as C
from chromegcn_amd import synth
from chromegcn_amd.finetune import GCNStage


def build(shared):
    torch.manual_seed(0)
    dev = torch.device("cuda")
    model = C.ChromeGCN(128, 128, synth.N_LABELS, 0.2, True, 2).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    stage = GCNStage(model, opt, "hic", dev, input_grad=True, cache_input_aggregation=False)
    n = 15182
    names = []
    for k in range(16):
        feats = synth.chrom_features(n, 128, synth.N_LABELS, 7 if shared else 7 + k)
        hic = synth.contact_graph(n, 250000, 7 if shared else 7 + k)
        nm = "c%d" % k
        stage.add_chromosome(nm, feats, hic)
        names.append(nm)
    if shared:
        first = stage.chroms[names[0]]
        for nm in names[1:]:
            c = stage.chroms[nm]
            c.x, c.graph, c.target = first.x, first.graph, first.target
    return stage, names


def main():
    out = {}
    stages = {"distinct": build(False), "shared": build(True)}
    for rep in range(4):
        for k, (st, names) in stages.items():
            for _ in range(3):
                st.run_split("train", names, to_cpu=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                st.run_split("train", names, to_cpu=False)
            torch.cuda.synchronize()
            out.setdefault(k, []).append(round((time.perf_counter() - t0) / 20 * 1e3, 4))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
