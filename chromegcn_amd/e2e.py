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


def windows_per_chrom(chroms, windows):
    """{chrom: window count}: `windows` is one count for every chromosome, a {chrom: count} dict, or "full" = the
    chromosome's real size in the synthetic genome (synth.chrom_nodes: chr21 = 5 776 windows with peaks)"""
    if isinstance(windows, str):
        if windows != "full":
            raise ValueError("windows must be an int, a dict or 'full'")
        return {c: synth.chrom_nodes(c) for c in chroms}
    if isinstance(windows, dict):
        return {c: int(windows[c]) for c in chroms}
    return {c: int(windows) for c in chroms}


def synthetic_windows(chroms, windows, seq_length, n_labels, seed=0):
    """tokens [N, L] int64 in {0..4}, targets [N, C], locs [(chrom, start, end)] in file order (chromosome by
    chromosome, ascending start: data/5merge_seqs_and_labels.py:70)"""
    per = windows_per_chrom(chroms, windows)
    total = sum(per.values())
    g = torch.Generator().manual_seed(seed)
    tokens = torch.randint(0, 5, (total, seq_length), generator=g)
    targets = (torch.rand(total, n_labels, generator=g) < 0.05).float()
    locs = [(c, 1000 * i, 1000 * i + 1000) for c in chroms for i in range(per[c])]
    return tokens, targets, locs


def run_pipeline(dev, windows=4096, seq_length=2000, d=128, layers=2, dropout=0.2, epochs=5, warmup=2, hic_like=False,
                 hip_graphs=True, chroms=E2E_CHROMS, batch_size=64, group=None, return_feats=False):
    """returns (timings dict, stage, names).  Encoder in eval mode (the -save_feats pass, pretrain.py:9-12).

    group (a torch.distributed process group): the multi-rank form.  The reference runs the encoder under
    nn.DataParallel (main.py:92-94: replicas of the module, batches scattered over the GPUs, outputs gathered on
    GPU 0) and the GCN stage on one GPU.  Here the unit of sharding is the chromosome in BOTH stages: every rank holds
    a replica of the encoder and pushes the windows of the chromosomes the shard plan (dist.plan_shards) gives it;
    their features never leave that rank's HBM -- they are collected there (handoff.FeatureCollector) and registered
    with its GCNStage, which then runs the same plan (finetune.GCNStage.run_split: one chromosome per rank per step
    group, one all-reduce of the flat gradient arena).  No feature tensor crosses a link."""
    import torch.distributed as dist
    from .dist import plan_shards
    world = dist.get_world_size(group) if group is not None else 1
    rank = dist.get_rank(group) if group is not None else 0
    n_labels = synth.N_LABELS
    tokens, targets, locs = synthetic_windows(chroms, windows, seq_length, n_labels)
    torch.manual_seed(0)   # identical replicas on every rank
    enc = StrandPair(WindowEncoder(n_labels, seq_length)).to(dev)
    model = ChromeGCN(d, d, n_labels, dropout, True, layers).to(dev)
    with torch.no_grad():  # main.py:78-81: the GCN's head starts from the encoder's classifier + BatchNorm affine
        model.out.load_state_dict(enc.model.classifier.state_dict())
        model.batch_norm.load_state_dict(enc.model.batch_norm.state_dict())
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    wpc = windows_per_chrom(chroms, windows)
    graphs = {}
    for c in chroms:   # the chromosome's own contact budget, scaled with the share of its windows that is used
        pairs = max(1, int(round(synth.PAIRS_PER_CHROM * wpc[c] / synth.chrom_nodes(c))))
        graphs[c] = synth.contact_graph(wpc[c], pairs, synth.chrom_seed(c), hic_like)
    stage = GCNStage(model, opt, "hic", dev, hip_graphs=hip_graphs, input_grad=True, cache_input_aggregation=False,
                     group=group)
    # ---- who encodes what: the stage's own shard plan (every rank registers every chromosome -- size, labels, cost --
    # without data: feature placeholders on the meta device, the real targets, the graph)
    rows_of = {c: [i for i, l in enumerate(locs) if l[0] == c] for c in chroms}
    if world > 1:
        for c in chroms:
            stub = torch.empty((len(rows_of[c]), d), device="meta")
            stage.add_chromosome(c, {"forward": stub, "backward": stub, "target": targets[rows_of[c]]}, graphs[c], defer=True)
        plan = plan_shards({c: stage._meta[c][2] for c in chroms}, world)
        mine = [c for c in chroms if plan.owner[c] == rank]
    else:
        mine = list(chroms)
    my_rows = torch.tensor([i for c in mine for i in rows_of[c]], dtype=torch.long)
    tokens_d, targets_d = tokens[my_rows].to(dev), targets[my_rows].to(dev)
    my_locs = [locs[i] for i in my_rows.tolist()]

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier(group)
    # ---- encoder: features of this rank's windows, both strands, collected on the device
    if len(my_locs):
        nb = min(2 * batch_size, len(my_locs))
        extract_features(enc, tokens_d[:nb], targets_d[:nb], my_locs[:nb], FeatureCollector())  # warm-up (MIOpen)
    fence()
    t0 = time.perf_counter()
    col = extract_features(enc, tokens_d, targets_d, my_locs, FeatureCollector(), batch_size)
    fence()
    t_enc = time.perf_counter() - t0
    # ---- hand-off: regroup per chromosome (device), normalise + upload the graphs (one-time, like every stage load)
    t0 = time.perf_counter()
    feats = col.finish()
    fence()
    t_regroup = time.perf_counter() - t0
    t0 = time.perf_counter()
    if world > 1:
        for c in mine:   # materialise what this rank owns; the other chromosomes stay registered-only
            stage.add_chromosome(c, feats[c], graphs[c])
        names = list(chroms)
    else:
        names = col.to_stage(stage, graphs)
    fence()
    t_load = time.perf_counter() - t0
    # ---- GCN stage: train epochs in reference semantics
    for _ in range(max(warmup, 1)):
        stage.run_split("train", names, to_cpu=False)
    fence()
    per = []
    for _ in range(epochs):
        t0 = time.perf_counter()
        _, _, loss = stage.run_split("train", names, to_cpu=False)
        per.append(time.perf_counter() - t0)
    n_win = sum(wpc.values())
    out = {"windows": n_win, "windows_per_chrom": wpc, "encoder_s": t_enc, "regroup_s": t_regroup, "stage_load_s": t_load,
           "gcn_epoch_s": float(np.median(per)), "gcn_epoch_p10_s": float(np.percentile(per, 10)),
           "gcn_epoch_p90_s": float(np.percentile(per, 90)), "epochs": epochs, "final_loss": loss,
           "feat_device": str(feats[mine[0]]["forward"].device) if mine else str(dev), "owned": mine,
           "encoded_windows_this_rank": len(my_locs)}
    if return_feats:
        out["feats"] = feats
    return out, stage, names


def bench(args, dev, world, rank):
    import torch.distributed as dist
    steps = args.steps if args.steps is not None else 10
    warmup = args.warmup if args.warmup is not None else 2
    # 8 ranks want at least 8 chromosomes: the 3 smallest train chromosomes at N = 1 (the round-2 line), the 8 smallest beyond
    chroms = E2E_CHROMS if world <= 3 else tuple(sorted((c for c in synth.HG19_LEN if synth.split_of(c) == "train"),
                                                         key=synth.chrom_nodes)[:max(8, world)])
    full = args.e2e_windows <= 0   # --e2e-windows 0: REAL chromosome sizes -- chr21 (5 776 windows x 2 000 tokens, its own
    if full:                       # 250 000 contact pairs) on one rank, the `world` smallest train chromosomes beyond
        chroms = ("chr21",) if world == 1 else tuple(sorted((c for c in synth.HG19_LEN if synth.split_of(c) == "train"),
                                                            key=synth.chrom_nodes)[:world])
    t, stage, names = run_pipeline(dev, windows="full" if full else args.e2e_windows, d=args.d, layers=args.layers, dropout=args.dropout,
                                   epochs=steps, warmup=warmup, hic_like=args.hic_like, hip_graphs=not args.no_hip_graph,
                                   chroms=chroms, group=dist.group.WORLD if world > 1 else None)
    n = t["windows"]
    enc_rate, gcn_rate = n / t["encoder_s"], n / t["gcn_epoch_s"]
    total = t["encoder_s"] + t["regroup_s"] + t["gcn_epoch_s"]
    return {
        "metric": "end-to-end windows/sec: Expecto-shaped encoder (stock torch-ROCm) -> device hand-off -> 2-layer gated GCN train epoch",
        "value": n / total, "unit": "windows/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": t["gcn_epoch_s"] * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("config 5 at real chromosome size: %s (tokens in {0..4}, length 2000, encoder batch 64), " % ", ".join(
                                   "%s = %d windows" % (c, k) for c, k in t["windows_per_chrom"].items()) if full else
                                "config 5 scaled down: %d chromosomes x %d windows (tokens in {0..4}, length 2000, encoder batch 64), "
                                % (len(names), args.e2e_windows)) +
                               "features f/r handed to the GCN stage on the device; GCN stage = train epoch in reference "
                               "semantics, d=%d, L=%d, C=%d" % (args.d, args.layers, synth.N_LABELS),
                   "generator": getattr(args, "generator", "hic_like" if args.hic_like else "uniform"),
                   "parallelism": "chromosomes sharded over %d rank(s): each rank encodes and trains the chromosomes it owns" % world},
        "encoder_windows_per_s": enc_rate, "gcn_windows_per_s": gcn_rate,
        "encoder_s": t["encoder_s"], "handoff_regroup_ms": t["regroup_s"] * 1e3, "stage_load_ms": t["stage_load_s"] * 1e3,
        "gcn_epoch_ms": {"median": t["gcn_epoch_s"] * 1e3, "p10": t["gcn_epoch_p10_s"] * 1e3, "p90": t["gcn_epoch_p90_s"] * 1e3},
        "note": "value = windows / (encoder pass + hand-off + one GCN train epoch); the encoder (~17 GFLOP/window, MIOpen fp32) "
                "dominates by orders of magnitude, which is why the two rates are reported separately",
        "final_loss": t["final_loss"],
    }
