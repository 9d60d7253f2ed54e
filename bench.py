#!/usr/bin/env python3
"""bench.py -- GCN windows/sec of the 2-layer gated GCN train step (BASELINE.json metric).

One "step" = the reference's per-chromosome train step (finetune.py:38-49): both strands forward,
BCE-with-logits, backward (incl. d/d features, finetune.py:33-34), SGD(momentum .9, wd 1e-6) step,
dropout 0.2 -- on one synthetic chromosome per rank, inputs already resident in HBM.
N = 1: BASELINE.json configs[1] stand-in, "chr21-like" (n = 5776 windows, 250k contact pairs, d = 128,
L = 2, C = 103).  N > 1 (weak scaling): every rank owns its own chr21-like
chromosome, and each step ends with one all-reduce of the flat gradient buffer (RCCL).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
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

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="chr21", choices=["chr21", "config1", "chr1"])
    ap.add_argument("--hic-like", action="store_true", help="distance-decay contact generator instead of uniform")
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--no-hip-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--backend", default=os.environ.get("CGCN_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU functional tests)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    return ap.parse_args()


def workload_shape(name):
    from chromegcn_amd import synth
    if name == "chr21":
        return "chr21", synth.chrom_nodes("chr21"), synth.PAIRS_PER_CHROM
    if name == "chr1":
        return "chr1", synth.chrom_nodes("chr1"), synth.PAIRS_PER_CHROM
    return "cfg1", 5000, 125000


def layer_fwd_bytes(n, nnz, S, d, training=True):
    """Algorithmic HBM bytes of ONE fused-layer forward launch ('k_layer_fwd'):
    CSR (rowptr + col) + 1/deg + X read + params + Xn write + gate write (+ Z, H saved when training)."""
    b = 4 * (n + 1) + 4 * nnz + 4 * n + S * 4 * n * d + (4 * d * d + 8 * d + 4) + S * 4 * n * d + S * 4 * n
    if training:
        b += 2 * S * 4 * n * d
    return b


def time_dominant_kernel(stage, name, reps):
    """Average duration of the dominant kernel (k_layer_fwd, training variant, both strands) measured
    with HIP events on the stream it is launched on (torch's current stream)."""
    from chromegcn_amd import _lib
    c = stage.chroms[name]
    m = stage.model
    g = c.graph
    S, n, d = c.x.shape
    xn, z, h = torch.empty_like(c.x), torch.empty_like(c.x), torch.empty_like(c.x)
    gate = torch.empty(S, n, device=c.x.device)
    lib = _lib.load()
    w, b = m.GC1.weight.detach(), m.GC1.bias.detach()
    wg, cg = m.W1.weight.detach().view(-1), m.W1.bias.detach()

    def launch():
        _lib.check(lib.cgcn_layer_fwd(_lib.stream_ptr(), n, S, d, _lib.ptr(g.rowptr), _lib.ptr(g.col), _lib.ptr(g.val),
                                      _lib.ptr(g.row_scale), c.x.data_ptr(), w.data_ptr(), b.data_ptr(), wg.data_ptr(),
                                      cg.data_ptr(), xn.data_ptr(), z.data_ptr(), h.data_ptr(), gate.data_ptr(), 0.0, None, 0, None, None), "fwd")
    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3  # seconds


def cpu_baseline(args, n, pairs, seed, steps):
    """The oracle (torch-CPU restatement of the reference ops) timed on this box's host cores on the
    same workload: a bounded sample of `steps` train steps."""
    from oracle import chromegcn_oracle as O  # cpu_baseline leg: the oracle is the thing timed here, nowhere else
    from chromegcn_amd import synth
    ncpu = os.cpu_count() or 1
    feats = synth.chrom_features(n, args.d, synth.N_LABELS, 1000 + seed)
    hic = synth.contact_graph(n, pairs, seed, args.hic_like)
    torch.manual_seed(0)
    model = O.GatedGCNOracle(args.d, synth.N_LABELS, args.dropout, args.layers)
    opt = O.make_sgd(model, 0.25)
    data = {"c": feats}
    cache = {}

    def one(cached=True):
        t0 = time.perf_counter()
        O.finetune_epoch(model, data, {"c": hic}, opt, "train", "hic", adj_cache=cache if cached else None)
        return time.perf_counter() - t0

    # torch's CPU spmm does not scale to every core of a big host (256 threads measured 20x slower than
    # 8-32): pick the thread count that is fastest on this box, so the baseline is the CPU's best case.
    best = None
    for th in sorted({t for t in (4, 8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        one()  # warm-up at this thread count (the first call also builds the cached adjacency)
        dt = min(one(), one())
        if best is None or dt < best[1]:
            best = (th, dt)
        if dt > 4.0 * best[1] or dt > 8.0:
            break
    cores = best[0]
    torch.set_num_threads(cores)
    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    t_cached = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(max(1, steps // 4)):
        one(cached=False)  # reference behaviour: process_graph every chromosome every epoch (finetune.py:36)
    t_full = (time.perf_counter() - t0) / max(1, steps // 4)
    cpu_model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": n / t_cached, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": "%d train steps (f+r fwd, BCE, bwd, SGD) on the same %d-window chromosome, adjacency cached; "
                      "oracle = torch-CPU restatement of the reference ops" % (steps, n),
            "s_per_step": t_cached, "with_process_graph_windows_per_s": n / t_full, "host_cpus": ncpu,
            "cpu_model": cpu_model, "torch": torch.__version__}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    import chromegcn_amd as C
    from chromegcn_amd import synth
    from chromegcn_amd.finetune import GCNStage

    cname, n, pairs = workload_shape(args.workload)
    seed = synth.chrom_seed(cname) if cname.startswith("chr") else 0
    seed += 100 * rank  # every rank owns a different chromosome of the same shape
    feats = synth.chrom_features(n, args.d, synth.N_LABELS, 1000 + seed)
    hic = synth.contact_graph(n, pairs, seed, args.hic_like)

    torch.manual_seed(0)  # identical initial parameters on every rank
    model = C.ChromeGCN(args.d, args.d, synth.N_LABELS, args.dropout, True, args.layers).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)  # README.md:45 flags
    # Headline configuration: EVERY step redoes all four sparse aggregations, like the reference.  (The engine's
    # default additionally caches A X of the first layer, which is loop invariant because the node features are
    # fixed -- measured separately below and reported as an extra field, never as `value`.)
    stage = GCNStage(model, opt, "hic", dev, hip_graphs=not args.no_hip_graph, input_grad=True,
                     group=dist.group.WORLD if world > 1 else None, cache_input_aggregation=False)
    name = "%s_r%d" % (cname, rank)
    stage.add_chromosome(name, feats, hic)
    nnz = stage.chroms[name].graph.nnz

    def step():
        if world > 1:
            return stage.train_group(name, world)
        return stage.train_step(name)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _, _ = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())

    # inference (eval-mode forward of both strands), reported alongside
    for _ in range(3):
        stage.eval_step(name)
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        stage.eval_step(name)
    fence()
    eval_elapsed = time.perf_counter() - t1

    # extra: the engine's default configuration -- first-layer aggregation cached, and no gradient w.r.t. the input
    # features (finetune.py:33-34 asks for it but nothing can observe it)
    stage.cache_input_aggregation = True
    stage.input_grad = False
    stage._drop_graphs()
    for _ in range(max(args.warmup, 1)):
        step()
    fence()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    cached_elapsed = time.perf_counter() - t2
    if world > 1:
        t = torch.tensor([cached_elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cached_elapsed = float(t.item())

    out = None
    if rank == 0:
        k_s = time_dominant_kernel(stage, name, 200)
        alg = layer_fwd_bytes(n, nnz, 2, args.d, True)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("%s_d%d" % (args.workload, args.d))
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": "k_layer_fwd<S=2,D=%d,MB=1,HAS_VAL=false,FROM_CACHE=false> (training variant: writes Z,H)" % args.d,
                "achieved": alg / k_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": alg / k_s / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg, "avg_kernel_us": k_s * 1e6,
                "gather_GBps": 4.0 * nnz * 2 * args.d / k_s / 1e9,
                "mfma_TFLOPs": 2.0 * 2 * n * args.d * args.d / k_s / 1e12}
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the host baseline is reported at N=1 only
            cpu = cpu_baseline(args, n, pairs, seed, args.cpu_steps)
        value = world * n * args.steps / elapsed
        out = {
            "metric": "GCN windows/sec (2-layer, d_model=128) train step", "value": value, "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s-like synthetic Hi-C chromosome per rank: n=%d windows, %d contact pairs "
                                   "(nnz(A+I)=%d), d=%d, L=%d, C=%d, dropout=%.2f, SGD lr .25 m .9 wd 1e-6; "
                                   "train step = f+r fwd, BCE, bwd incl. d/dx, optimizer step%s"
                                   % (args.workload, n, pairs, nnz, args.d, args.layers, synth.N_LABELS, args.dropout,
                                      "; grad all-reduce over RCCL" if world > 1 else ""),
                       "generator": "hic_like" if args.hic_like else "uniform",
                       "hip_graph": not args.no_hip_graph, "parallelism": "chromosome-per-rank x%d" % world},
            "roofline": roof, "cpu_baseline": cpu,
            "inference_windows_per_s": world * n * args.steps / eval_elapsed,
            "engine_default_windows_per_s": world * n * args.steps / cached_elapsed,
            "engine_default_note": "cached first-layer aggregation (loop invariant) + no d loss/d features; not the headline",
            "final_loss": final_loss,
        }
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
