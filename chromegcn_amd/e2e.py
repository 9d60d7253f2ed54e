"""BASELINE.json config 5, end to end on the device: window encoder (stock PyTorch-ROCm convolutions) -> per-window
features of both strands -> device hand-off (handoff.FeatureCollector) -> GCN stage (the hand-written path) ->
multi-label head.  The reference does this in two program runs with a `chrom_feature_dict_<split>.pt` file between
them (pretrain.py:57-63 + utils/util_methods.py:183-199, then main.py:30-32 + finetune.py); here the features never
leave HBM.

`bench` is `bench.py --workload e2e`: encoder windows/s and GCN windows/s are reported SEPARATELY (SURVEY.md 8d,
config 5).  The encoder costs ~17 GFLOP per window (fp32, two strands), three orders of magnitude more than the GCN
stage per window, so the workload is a scaled-down genome: `--e2e-windows` windows (default 4096) on each of the three
smallest train chromosomes' slots, 250 000 / (n_c / windows) contact pairs each (same mean degree as the full
chromosome), tokens uniform in {0..4}, length 2000 (config_args.py:39), encoder batch 64."""
from __future__ import annotations

import time

import numpy as np
import torch

from . import synth
from .encoder import StrandPair, WindowEncoder, extract_features
from .finetune import GCNStage
from .handoff import FeatureCollector
from .layers import ChromeGCN

E2E_CHROMS = ("chr19", "chr20", "chr22")


def synthetic_windows(chroms, windows, seq_length, n_labels, seed=0):
    """tokens [N, L] int64 in {0..4}, targets [N, C], locs [(chrom, start, end)] in file order (chromosome by
    chromosome, ascending start: data/5merge_seqs_and_labels.py:70)"""
    g = torch.Generator().manual_seed(seed)
    tokens = torch.randint(0, 5, (len(chroms) * windows, seq_length), generator=g)
    targets = (torch.rand(len(chroms) * windows, n_labels, generator=g) < 0.05).float()
    locs = [(c, 1000 * i, 1000 * i + 1000) for c in chroms for i in range(windows)]
    return tokens, targets, locs


def run_pipeline(dev, windows=4096, seq_length=2000, d=128, layers=2, dropout=0.2, epochs=5, warmup=2, hic_like=False,
                 hip_graphs=True, chroms=E2E_CHROMS, batch_size=64):
    """returns (timings dict, stage, names).  Encoder in eval mode (the -save_feats pass, pretrain.py:9-12)."""
    n_labels = synth.N_LABELS
    tokens, targets, locs = synthetic_windows(chroms, windows, seq_length, n_labels)
    torch.manual_seed(0)
    enc = StrandPair(WindowEncoder(n_labels, seq_length)).to(dev)
    model = ChromeGCN(d, d, n_labels, dropout, True, layers).to(dev)
    with torch.no_grad():  # main.py:78-81: the GCN's head starts from the encoder's classifier + BatchNorm affine
        model.out.load_state_dict(enc.model.classifier.state_dict())
        model.batch_norm.load_state_dict(enc.model.batch_norm.state_dict())
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    tokens_d, targets_d = tokens.to(dev), targets.to(dev)
    # ---- encoder: features of every window, both strands, collected on the device
    extract_features(enc, tokens_d[:2 * batch_size], targets_d[:2 * batch_size], locs[:2 * batch_size], FeatureCollector())  # warm-up (MIOpen)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    col = extract_features(enc, tokens_d, targets_d, locs, FeatureCollector(), batch_size)
    torch.cuda.synchronize(dev)
    t_enc = time.perf_counter() - t0
    # ---- hand-off: regroup per chromosome (device), normalise + upload the graphs (one-time, like every stage load)
    graphs = {}
    for c in chroms:
        pairs = max(1, int(round(synth.PAIRS_PER_CHROM * windows / synth.chrom_nodes(c))))
        graphs[c] = synth.contact_graph(windows, pairs, synth.chrom_seed(c), hic_like)
    stage = GCNStage(model, opt, "hic", dev, hip_graphs=hip_graphs, input_grad=True, cache_input_aggregation=False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    feats = col.finish()
    torch.cuda.synchronize(dev)
    t_regroup = time.perf_counter() - t0
    t0 = time.perf_counter()
    names = col.to_stage(stage, graphs)
    torch.cuda.synchronize(dev)
    t_load = time.perf_counter() - t0
    # ---- GCN stage: train epochs in reference semantics
    for _ in range(max(warmup, 1)):
        stage.run_split("train", names, to_cpu=False)
    torch.cuda.synchronize(dev)
    per = []
    for _ in range(epochs):
        t0 = time.perf_counter()
        _, _, loss = stage.run_split("train", names, to_cpu=False)
        per.append(time.perf_counter() - t0)
    n_win = len(chroms) * windows
    return {"windows": n_win, "encoder_s": t_enc, "regroup_s": t_regroup, "stage_load_s": t_load,
            "gcn_epoch_s": float(np.median(per)), "gcn_epoch_p10_s": float(np.percentile(per, 10)),
            "gcn_epoch_p90_s": float(np.percentile(per, 90)), "epochs": epochs, "final_loss": loss,
            "feat_device": str(feats[names[0]]["forward"].device)}, stage, names


def bench(args, dev, world, rank):
    if world != 1:
        raise SystemExit("--workload e2e is a single-GPU line (the sharded GCN stage is --workload genome)")
    steps = args.steps if args.steps is not None else 10
    warmup = args.warmup if args.warmup is not None else 2
    t, stage, names = run_pipeline(dev, windows=args.e2e_windows, d=args.d, layers=args.layers, dropout=args.dropout,
                                   epochs=steps, warmup=warmup, hic_like=args.hic_like, hip_graphs=not args.no_hip_graph)
    n = t["windows"]
    enc_rate, gcn_rate = n / t["encoder_s"], n / t["gcn_epoch_s"]
    total = t["encoder_s"] + t["regroup_s"] + t["gcn_epoch_s"]
    return {
        "metric": "end-to-end windows/sec: Expecto-shaped encoder (stock torch-ROCm) -> device hand-off -> 2-layer gated GCN train epoch",
        "value": n / total, "unit": "windows/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": t["gcn_epoch_s"] * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config 5 scaled down: %d chromosomes x %d windows (tokens in {0..4}, length 2000, encoder batch 64), "
                               "features f/r handed to the GCN stage on the device; GCN stage = train epoch in reference "
                               "semantics, d=%d, L=%d, C=%d" % (len(names), args.e2e_windows, args.d, args.layers, synth.N_LABELS),
                   "generator": "hic_like" if args.hic_like else "uniform"},
        "encoder_windows_per_s": enc_rate, "gcn_windows_per_s": gcn_rate,
        "encoder_s": t["encoder_s"], "handoff_regroup_ms": t["regroup_s"] * 1e3, "stage_load_ms": t["stage_load_s"] * 1e3,
        "gcn_epoch_ms": {"median": t["gcn_epoch_s"] * 1e3, "p10": t["gcn_epoch_p10_s"] * 1e3, "p90": t["gcn_epoch_p90_s"] * 1e3},
        "note": "value = windows / (encoder pass + hand-off + one GCN train epoch); the encoder (~17 GFLOP/window, MIOpen fp32) "
                "dominates by orders of magnitude, which is why the two rates are reported separately",
        "final_loss": t["final_loss"],
    }
