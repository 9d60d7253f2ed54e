#!/usr/bin/env python3
"""bench.py -- GCN windows/sec of the 2-layer gated GCN over the whole GM12878-shaped Hi-C genome
(BASELINE.json metric), in the REFERENCE's semantics.

One "step" = one training epoch of the GCN stage (finetune.py:29-53 over every chromosome of the train split,
data/create_data.py:40-45): for each of the 16 train chromosomes, in dict order, both strands forward,
BCE-with-logits, backward INCLUDING d loss / d features (finetune.py:33-34), one SGD(momentum .9, wd 1e-6) step,
dropout 0.2 -- every one of the four sparse aggregations redone every step (nothing cached across steps except the
normalised CSR and the device-resident inputs).  The epoch returns what finetune.py:67 returns (all predictions in
chromosome order, targets, summed loss) -- predictions stay in HBM; one host sync per epoch (the loss).
`value` = (sum of train windows) x steps / wall time.  Graphs: SURVEY.md 8(d) config 3 (22 synthetic chromosomes,
N_c = round(0.12 hg19_len_c / 1 kb), 250 000 contact pairs each, seed = chromosome number; uniform generator).

N = 1: all 16 train chromosomes on one GPU (they fit: < 3 GB resident).
N > 1: STRONG scaling of the same epoch: chromosomes sharded across ranks by GCNStage.run_split / dist.plan_shards
(LPT), one step group = one chromosome per rank + ONE all-reduce of the flat gradient buffer (RCCL) + the same
optimizer step everywhere; predictions gathered once at the end of the epoch.  (k ranks => 16/k optimizer steps
per epoch instead of 16: the documented semantics change of data-parallel chromosomes)

    python bench.py                                   # N = 1, genome, 30 epochs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
    python bench.py --workload chr21|chr1|config1     # one chromosome per rank (weak scaling), secondary lines
    python bench.py --workload e2e                    # config 5: Expecto-shaped encoder -> hand-off -> GCN stage
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); harmless at N=1
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 30 epochs / 50 single-chromosome steps)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="genome", choices=["genome", "chr21", "config1", "chr1", "e2e"])
    ap.add_argument("--hic-like", action="store_true", help="distance-decay contact generator instead of uniform")
    ap.add_argument("--generator", default=None, choices=["uniform", "hic_like", "hub"],
                    help="contact generator (synth.contact_graph): uniform (headline), hic_like = distance decay, hub = top-K-style "
                         "heavy-tailed degrees with hubs of 2-10 k neighbours (data/7create_graph_new.py:93-104)")
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--no-hip-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (profiling runs)")
    ap.add_argument("--no-roofline", action="store_true", help="tuning runs only: skip the isolated layer-forward timing (roofline = null)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the host baseline sample")
    ap.add_argument("--backend", default=os.environ.get("CGCN_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU functional tests)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--p2p-allreduce", action="store_true",
                    help="N > 1: one-shot peer-to-peer gradient all-reduce over symmetric memory instead of RCCL's (SURVEY section 5)")
    ap.add_argument("--no-group-graph", action="store_true",
                    help="N > 1: do not capture the collective + optimizer step into the step's HIP graph")
    ap.add_argument("--e2e-windows", type=int, default=4096, help="e2e: windows per chromosome pushed through the encoder")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous check only: every rank joins the process group, rank 0 prints one JSON line "
                         "(n_gpus, ranks_seen_by_backend) and nothing touches a GPU")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment (README.md:34 runs the reference with ONE
    command): start the N ranks ourselves -- `python -m torch.distributed.run`, one process per GPU, rendezvous on
    127.0.0.1 -- BEFORE anything in this process touches the GPU (a process that has initialised HIP must neither fork
    GPU work nor exec), relay rank 0's JSON line and the job's exit code.  Under a launcher (WORLD_SIZE set) this is a
    no-op and the process is one rank."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    import socket
    import subprocess
    if not args.share_gpu and not args.dry_run:
        have = torch.cuda.device_count()   # counts devices without initialising the runtime
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (use --share-gpu --backend gloo for a "
                             "single-GPU functional run)" % (args.gpus, have))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:                 # rank 0's JSON line goes to our stdout; anything else to stderr
        if ln.startswith("{"):
            line = ln
            sys.stdout.write(ln)
            sys.stdout.flush()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited 0 without printing a result line\n")
        rc = 1
    return rc


def ranks_seen(world, dev):
    """how many ranks the backend actually connected: an all-reduce of ones over the job's process group"""
    if world <= 1:
        return 1
    t = torch.ones(1, device=dev, dtype=torch.float32)
    dist.all_reduce(t)
    return int(round(float(t.item())))


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------
def genome_train_names():
    from chromegcn_amd import synth
    return [c for c in synth.HG19_LEN if synth.split_of(c) == "train"]


def single_shape(name):
    from chromegcn_amd import synth
    if name == "chr21":
        return "chr21", synth.chrom_nodes("chr21"), synth.PAIRS_PER_CHROM
    if name == "chr1":
        return "chr1", synth.chrom_nodes("chr1"), synth.PAIRS_PER_CHROM
    return "cfg1", 5000, 125000


def layer_fwd_bytes(n, nnz, S, d):
    """Algorithmic HBM bytes of ONE fused-layer forward launch in training, SURVEY.md 8(d): CSR (rowptr + col) +
    1/deg + X read + parameters + X' write + gate write + the saved Z.  (The kernel also writes H = A X for the
    weight gradient -- a by-product of the (A X) W re-association that 8(d) does not list, so it is NOT counted.)"""
    return 4 * (n + 1) + 4 * nnz + 4 * n + S * 4 * n * d + (4 * d * d + 8 * d + 4) + S * 4 * n * d + S * 4 * n + S * 4 * n * d


def time_layer_fwd(stage, name, reps, dropout_p):
    """Average duration of one layer forward (cgcn_layer_fwd, training form, both strands: the fused k_layer_fwd or the
    k_aggregate_sliced + k_layer_dense pair) on one chromosome, measured with HIP events
    on the stream the library launches on (torch's current stream), in the two forms a train step uses: layer 1
    (inter-layer dropout with the run's p and RNG state) and layer 2 (BatchNorm column statistics for the head)."""
    import ctypes
    from chromegcn_amd import _lib
    c = stage.chroms[name]
    m = stage.model
    g = c.graph
    S, n, d = c.x.shape
    xn, z, h = torch.empty_like(c.x), torch.empty_like(c.x), torch.empty_like(c.x)
    gate = torch.empty(S, n, device=c.x.device)
    lib = _lib.load()
    rng = m._rng_state
    rows = ctypes.c_int(0)
    tiles = lib.cgcn_layer_fwd_colstats_tiles(n, S, d, ctypes.byref(rows))
    colstats = torch.empty((tiles, S, d, 2), device=c.x.device)

    def launch(layer):
        gc, wk = getattr(m, "GC%d" % layer), getattr(m, "W%d" % layer)
        last = layer == m.n_layers
        _lib.check(lib.cgcn_layer_fwd(_lib.stream_ptr(), n, S, d, _lib.ptr(g.rowptr), _lib.ptr(g.col), _lib.ptr(g.val),
                                      _lib.ptr(g.row_scale), c.x.data_ptr(), gc.weight.data_ptr(), gc.bias.data_ptr(),
                                      wk.weight.data_ptr(), wk.bias.data_ptr(), xn.data_ptr(), z.data_ptr(),
                                      h.data_ptr(), gate.data_ptr(), 0.0 if last else float(dropout_p),
                                      None if (last or dropout_p <= 0) else _lib.ptr(rng), layer, None,
                                      colstats.data_ptr() if last else None), "fwd")
    out = []
    for layer in (1, m.n_layers):
        for _ in range(3):
            launch(layer)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch(layer)
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e-3)
    return out  # seconds: [layer 1 form, last-layer form]


def stored_traffic(key):
    """HBM bytes per launch from the PMC passes of an earlier profiling run (profiles/traffic.json; FETCH_SIZE doubled
    per MI355X_MICROARCH.md + WRITE_SIZE).  A stored, offline value: the bench line says so and names the tag."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(tpath))
    except Exception:
        return None, None
    ent = t.get(key)
    if isinstance(ent, dict):
        return ent.get("bytes_per_launch"), ent.get("tag")
    return ent, t.get("_tag")


def host_info():
    cpu_model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return cpu_model


def cpu_baseline(args, chroms, budget_s):
    """The oracle (torch-CPU restatement of the reference ops, finetune.py:29-53) timed on this box's host cores on a
    BOUNDED sample of the same workload: train steps on the sample's chromosomes, adjacency cached (the reference
    re-normalises it every chromosome every epoch, finetune.py:36 -- reported separately)."""
    from oracle import chromegcn_oracle as O  # cpu_baseline leg: the oracle is the thing timed here, nowhere else
    from chromegcn_amd import synth
    ncpu = os.cpu_count() or 1
    data, graphs = {}, {}
    for nm, n, pairs, seed in chroms:
        data[nm] = synth.chrom_features(n, args.d, synth.N_LABELS, 1000 + seed)
        graphs[nm] = synth.contact_graph(n, pairs, seed, args.hic_like)
    n_sample = sum(c[1] for c in chroms)
    torch.manual_seed(0)
    model = O.GatedGCNOracle(args.d, synth.N_LABELS, args.dropout, args.layers)
    opt = O.make_sgd(model, 0.25)
    cache = {}

    def one(cached=True):
        t0 = time.perf_counter()
        O.finetune_epoch(model, data, graphs, opt, "train", "hic", adj_cache=cache if cached else None)
        return time.perf_counter() - t0

    # torch's CPU spmm does not scale to every core of a big host (256 threads measured 20x slower than 8-32): pick
    # the thread count that is fastest on this box, so the baseline is the CPU's best case.
    best = None
    for th in sorted({t for t in (4, 8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        one()  # warm-up at this thread count (the first call also builds the cached adjacency)
        dt = one()
        if best is None or dt < best[1]:
            best = (th, dt)
        if dt > 4.0 * best[1] or dt > budget_s / 3:
            break
    cores = best[0]
    torch.set_num_threads(cores)
    reps = max(2, min(20, int(budget_s / max(best[1], 1e-3))))
    ts = [one() for _ in range(reps)]
    t_cached = float(np.median(ts))
    t_full = one(cached=False)  # reference behaviour: process_graph every chromosome every epoch (finetune.py:36)
    return {"value": n_sample / t_cached, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": "%d passes over %s (%d windows; f+r fwd, BCE, bwd incl. d/dx, SGD step per chromosome), "
                      "adjacency cached; oracle = torch-CPU restatement of the reference ops; median" %
                      (reps, "+".join(c[0] for c in chroms), n_sample),
            "s_per_pass": t_cached, "p10_s": float(np.percentile(ts, 10)), "p90_s": float(np.percentile(ts, 90)),
            "with_process_graph_windows_per_s": n_sample / t_full, "host_cpus": ncpu,
            "cpu_model": host_info(), "torch": torch.__version__}


# ------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    args.generator = args.generator or ("hic_like" if args.hic_like else "uniform")
    args.hic_like = args.generator          # synth's generator argument (False / True / name)
    rc = self_launch(args)
    if rc is not None:
        sys.exit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.dry_run:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = ranks_seen(world, torch.device("cpu"))
        out = {"dry_run": True, "n_gpus": world, "ranks_seen_by_backend": seen, "backend": "gloo",
               "launcher": "self (torch.distributed.run child)" if os.environ.get("TORCHELASTIC_RUN_ID") else "none"}
        if rank == 0:
            print(json.dumps(out))
            sys.stdout.flush()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return out
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    seen = ranks_seen(world, dev)
    if args.workload == "e2e":
        from chromegcn_amd import e2e
        out = e2e.bench(args, dev, world, rank)
        if rank == 0:
            print(json.dumps(out))
            sys.stdout.flush()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return out

    import chromegcn_amd as C
    from chromegcn_amd import synth
    from chromegcn_amd.finetune import GCNStage

    genome = args.workload == "genome"
    steps = args.steps if args.steps is not None else (30 if genome else 50)
    warmup = args.warmup if args.warmup is not None else (5 if genome else 10)

    torch.manual_seed(0)  # identical initial parameters on every rank
    model = C.ChromeGCN(args.d, args.d, synth.N_LABELS, args.dropout, True, args.layers).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)  # README.md:45 flags
    # Reference semantics: EVERY step redoes all four sparse aggregations and produces d loss / d features, like the
    # reference.  (The engine's defaults cache A X of the first layer -- loop invariant, the features are fixed -- and
    # skip the unobservable input gradient; measured separately below, never as `value`.)
    stage = GCNStage(model, opt, "hic", dev, hip_graphs=not args.no_hip_graph, input_grad=True,
                     group=dist.group.WORLD if world > 1 else None, cache_input_aggregation=False,
                     group_graph=False if args.no_group_graph else None, p2p_allreduce=True if args.p2p_allreduce else None)

    if genome:
        names = genome_train_names()
        shapes = []
        for nm in names:  # N > 1: registered only -- a rank normalises and uploads the chromosomes the shard plan gives it
            feats, hic = synth.synthetic_chromosome(nm, d=args.d, hic_like=args.hic_like)
            stage.add_chromosome(nm, feats, hic, defer=world > 1)
            shapes.append((nm, feats["forward"].shape[0], int(hic.nnz) + feats["forward"].shape[0]))
        windows = sum(s[1] for s in shapes)

        def step():
            return stage.run_split("train", names, to_cpu=False)
    else:
        cname, n, pairs = single_shape(args.workload)
        seed = (synth.chrom_seed(cname) if cname.startswith("chr") else 0) + 100 * rank  # a different chromosome per rank
        feats = synth.chrom_features(n, args.d, synth.N_LABELS, 1000 + seed)
        hic = synth.contact_graph(n, pairs, seed, args.hic_like)
        name = "%s_r%d" % (cname, rank)
        stage.add_chromosome(name, feats, hic)
        names = [name]
        shapes = [(name, n, stage.chroms[name].graph.nnz)]
        windows = n * world

        def step():
            if world > 1:
                return stage.train_group(name, world)
            return stage.train_step(name)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, k):
        """k steps bracketed by barrier + synchronize; also the per-step host times (genome: every epoch ends with
        its own loss sync, so these are true per-epoch times)"""
        fence()
        per = []
        t0 = time.perf_counter()
        for _ in range(k):
            s0 = time.perf_counter()
            r = fn()
            per.append(time.perf_counter() - s0)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, per, r

    for _ in range(max(warmup, 1)):
        step()
    elapsed, per, last = timed(step, steps)
    final_loss = float(last[2]) if genome else float(last[0].item())

    extras = {}
    if not args.no_extras:
        # inference: eval-mode forward of both strands over the same chromosomes
        if genome:
            def ev():
                return stage.run_split("valid", names, to_cpu=False)
        else:
            def ev():
                return stage.eval_step(names[0])
        for _ in range(2):
            ev()
        ev_el, _, _ = timed(ev, max(steps // 2, 3))
        extras["inference_windows_per_s"] = windows * max(steps // 2, 3) / ev_el
        # the engine's default configuration: first-layer aggregation cached, no gradient w.r.t. the input features
        # (finetune.py:33-34 asks for it but nothing can observe it)
        stage.cache_input_aggregation = True
        stage.input_grad = False
        stage._drop_graphs()
        for _ in range(max(warmup, 1)):
            step()
        d_el, _, _ = timed(step, steps)
        extras["engine_default_windows_per_s"] = windows * steps / d_el
        extras["engine_default_note"] = ("cached first-layer aggregation (loop invariant) + no d loss / d features; "
                                         "not the headline")
        stage.cache_input_aggregation = False
        stage.input_grad = True
        stage._drop_graphs()

    out = None
    if rank == 0:
        # ---- roofline of the dominant operation (the layer forward: largest share of the epoch), measured live
        reps = 30 if genome else 200
        tot_t = tot_b = tot_g = tot_f = 0.0
        per_chrom = {}
        for nm, n, nnz in shapes:
            if nm not in stage.chroms or args.no_roofline:
                continue
            t1, t2 = time_layer_fwd(stage, nm, reps, args.dropout)
            b = layer_fwd_bytes(n, nnz, 2, args.d)
            tot_t += t1 + t2
            tot_b += 2 * b
            tot_g += 2 * 4.0 * nnz * 2 * args.d
            tot_f += 2 * 2.0 * 2 * n * args.d * args.d
            per_chrom[nm] = {"n": n, "nnz": nnz, "us_layer1": t1 * 1e6, "us_last": t2 * 1e6, "GBps": b / ((t1 + t2) / 2) / 1e9}
        wl_key = ("genome" if genome else args.workload) + {"uniform": "", "hic_like": "_hic", "hub": "_hub"}[args.generator] + "_d%d" % args.d
        traffic, ttag = stored_traffic(wl_key)
        nl = 2 * len(per_chrom)
        roof = None if not per_chrom else {"bound": "hbm",
                "kernel": "layer forward, training form (writes Z, H; layer 1 with dropout, last layer with BatchNorm column "
                          "statistics) = one cgcn_layer_fwd call: k_aggregate_sliced + k_layer_dense<S=2,D=%d> on feature tables "
                          ">= 8 MiB, the fused k_layer_fwd below; HIP events around the call, average over the %d calls of one %s" %
                          (args.d, nl, "train epoch" if genome else "train step"),
                "achieved": tot_b / tot_t / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": tot_b / tot_t / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                "traffic_source": None if traffic is None else "stored profile value (profiles/traffic.json, tag %s), not measured in this run" % ttag,
                "algorithmic_bytes_per_launch": tot_b / nl, "avg_kernel_us": tot_t / nl * 1e6,
                "bytes_formula": "SURVEY 8(d): 4(n+1)+4nnz+4n + S*4nd (X) + 4d^2+8d+4 + S*4nd (X') + S*4n (gate) + S*4nd (Z); H not counted",
                "gather_GBps": tot_g / tot_t / 1e9, "mfma_TFLOPs": tot_f / tot_t / 1e12,
                "per_chromosome": per_chrom if genome else None}
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the host baseline is reported at N=1 only
            if genome:  # bounded sample of the same genome: its smallest, a middle and its largest train chromosome
                by_n = sorted(shapes, key=lambda s: s[1])
                pick = [by_n[0], by_n[len(by_n) // 2], by_n[-1]]
                sample = [(nm, n, synth.PAIRS_PER_CHROM, synth.chrom_seed(nm)) for nm, n, _ in pick]
            else:
                sample = [(names[0], shapes[0][1], single_shape(args.workload)[2],
                           (synth.chrom_seed(single_shape(args.workload)[0]) if args.workload != "config1" else 0))]
            cpu = cpu_baseline(args, sample, args.cpu_seconds)
        per_ms = np.array(per) * 1e3
        if genome:
            wl = ("synthetic GM12878-shaped genome (SURVEY 8d config 3): %d train chromosomes, %d windows, 250000 contact "
                  "pairs each (nnz(A+I) %d..%d), d=%d, L=%d, C=%d, dropout=%.2f, SGD lr .25 m .9 wd 1e-6; step = one train "
                  "epoch in reference semantics: per chromosome f+r fwd, BCE, bwd incl. d/dx, optimizer step; all four "
                  "aggregations every step%s" %
                  (len(names), windows, min(s[2] for s in shapes), max(s[2] for s in shapes), args.d, args.layers,
                   synth.N_LABELS, args.dropout,
                   "; chromosomes sharded over %d ranks (LPT), one flat-gradient all-reduce per step group (RCCL)" % world if world > 1 else ""))
        else:
            wl = ("%s-like synthetic Hi-C chromosome per rank: n=%d windows, %d contact pairs (nnz(A+I)=%d), d=%d, L=%d, "
                  "C=%d, dropout=%.2f, SGD lr .25 m .9 wd 1e-6; train step = f+r fwd, BCE, bwd incl. d/dx, optimizer step%s"
                  % (args.workload, shapes[0][1], single_shape(args.workload)[2], shapes[0][2], args.d, args.layers,
                     synth.N_LABELS, args.dropout, "; grad all-reduce over RCCL" if world > 1 else ""))
        out = {
            "metric": "GCN windows/sec (2-layer, d_model=128) on GM12878 Hi-C graph" if genome else
                      "GCN windows/sec (2-layer, d_model=128) train step, one chromosome",
            "value": windows * steps / elapsed, "unit": "windows/s",
            "n_gpus": world, "ranks_seen_by_backend": seen, "backend": args.backend if world > 1 else None,
            "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if genome else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl, "generator": args.generator,
                       "hip_graph": not args.no_hip_graph,
                       "parallelism": ("chromosomes sharded over %d rank(s)" % world) if genome else "chromosome-per-rank x%d" % world,
                       "allreduce": stage.allreduce_kind if world > 1 else None,
                       "step_group_graph": bool(stage._group_graph_enabled()) if world > 1 else None},
            "step_ms": {"median": float(np.median(per_ms)), "p10": float(np.percentile(per_ms, 10)),
                        "p90": float(np.percentile(per_ms, 90)), "n": len(per),
                        "note": "per-step host time on rank 0" + (" (each epoch ends with its own loss sync)" if genome else " (launch only: steps are asynchronous)")},
            "value_at_median": windows / (float(np.median(per_ms)) * 1e-3) if genome else None,
            "roofline": roof, "cpu_baseline": cpu, "final_loss": final_loss,
        }
        out.update(extras)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
