#!/usr/bin/env python3
"""A/B of GCNStage.prefetch_input_aggregation (tuning tool): trains the synthetic genome for a few epochs in both orders,
checks parameters bit for bit, and times epochs of each order alternately.  Needs profiles/patches/prefetch_agg.patch applied
(the switch was measured neutral and is not in the product: profiles/r05_prefetch_agg_experiment.txt)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import chromegcn_amd as C
from chromegcn_amd import synth
from chromegcn_amd.finetune import GCNStage


def make(prefetch):
    torch.manual_seed(0)
    dev = torch.device("cuda")
    model = C.ChromeGCN(128, 128, synth.N_LABELS, 0.2, True, 2).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    stage = GCNStage(model, opt, "hic", dev, input_grad=True, cache_input_aggregation=False, prefetch_input_aggregation=prefetch)
    names = []
    for chrom in synth.HG19_LEN:
        if synth.split_of(chrom) != "train":
            continue
        feats, hic = synth.synthetic_chromosome(chrom, d=128)
        stage.add_chromosome(chrom, feats, hic)
        names.append(chrom)
    return model, stage, names


def main():
    out = {}
    stages = {pf: make(pf) for pf in (False, True)}
    losses = {}
    for pf, (m, st, names) in stages.items():
        ls = []
        for e in range(4):
            _, _, loss = st.run_split("train", names, to_cpu=False)
            ls.append(float(loss))
        losses[pf] = ls
    out["losses"] = {str(k): v for k, v in losses.items()}
    pa = torch.cat([p.detach().flatten() for p in stages[False][0].parameters()])
    pb = torch.cat([p.detach().flatten() for p in stages[True][0].parameters()])
    out["params_bitwise_equal"] = bool(torch.equal(pa, pb))
    out["losses_equal"] = losses[False] == losses[True]
    times = {False: [], True: []}
    for rep in range(6):
        for pf, (m, st, names) in stages.items():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for e in range(20):
                st.run_split("train", names, to_cpu=False)
            torch.cuda.synchronize(); times[pf].append((time.perf_counter() - t0) / 20 * 1e3)
    out["epoch_ms_serial"] = times[False]
    out["epoch_ms_prefetch"] = times[True]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
