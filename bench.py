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
    ap.add_argument("--adj-type", default="hic", choices=["hic", "both", "constant", "none"],
                    help="the reference's -adj_type (config_args.py:45, utils/util_methods.py:146-174): hic (headline), both = Hi-C + "
                         "the +-7 band + I with summed values, constant = the band + I alone (the library's sliding-window route), "
                         "none = I")
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--no-hip-graph", action="store_true")
    ap.add_argument("--no-epoch-graph", action="store_true", help="A/B: one HIP graph launch per chromosome instead of one per epoch (single rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (profiling runs)")
    ap.add_argument("--no-roofline", action="store_true", help="tuning runs only: skip the isolated layer-forward timing (roofline = null)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the host baseline sample")
    ap.add_argument("--backend", default=os.environ.get("CGCN_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU functional tests)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--force-collectives", action="store_true",
                    help="testing only: with ONE rank, still take the N > 1 code path -- process group (backend nccl = RCCL), shard "
                         "plan, the step group captured with its all-reduce, prediction gather -- the only way to drive that "
                         "path through RCCL on a one-GPU box")
    ap.add_argument("--p2p-allreduce", action="store_true",
                    help="N > 1: one-shot peer-to-peer gradient all-reduce over symmetric memory instead of RCCL's (SURVEY section 5)")
    ap.add_argument("--gather", default="rank0", choices=["rank0", "all", "none"],
                    help="N > 1: where an epoch's predictions are assembled -- rank0 (default; what nn.DataParallel does with "
                         "the replicas' outputs, reference main.py:92-94: direct sends to rank 0), all (all-gather: every "
                         "rank holds the whole split), none (each rank keeps its chromosomes' rows)")
    ap.add_argument("--no-group-graph", action="store_true",
                    help="N > 1: do not capture the collective + optimizer step into the step's HIP graph")
    ap.add_argument("--e2e-windows", type=int, default=4096, help="e2e: windows per chromosome pushed through the encoder")
    ap.add_argument("--probe", action="store_true",
                    help="internal (launch ladder): a short functional job -- the first two step groups of the epoch, one timed "
                         "step, no extras -- whose only purpose is to come back")
    ap.add_argument("--rung", type=int, default=None, help="internal (launch ladder): the rung this job runs on; no further probing")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous check only: every rank joins the process group, rank 0 prints one JSON line "
                         "(n_gpus, ranks_seen_by_backend) and nothing touches a GPU")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# N > 1: the fallback ladder.  The first multi-GPU run on a node exercises code a one-GPU box cannot (RCCL with more
# than one rank, the all-reduce captured into the step's HIP graph, point-to-point prediction sends), and a hang inside a
# captured collective cannot be recovered from inside the process that issued it.  So before the measured job starts, a
# process that has NOT touched a GPU -- the self-launching parent, or rank 0 of an external launcher's ranks -- runs a
# short PROBE job per rung as a child (its own torch.distributed.run, same N, two step groups of the real epoch with
# that rung's collectives) under a timeout, and the measured job runs with the first rung whose probe came back.  The
# JSON line records which rung ran and why the earlier ones did not (`launch_ladder`).  Children only, never an exec.
# ------------------------------------------------------------------------------------------------------------------
LADDER = [
    ("step group as one HIP graph (collective inside) + rows sent to rank 0", []),
    ("captured fwd+bwd, eager all-reduce, all-gather of predictions", ["--no-group-graph", "--gather", "all"]),
    ("no HIP graphs at all, eager all-reduce, all-gather of predictions", ["--no-group-graph", "--gather", "all", "--no-hip-graph"]),
]
_PASS_THROUGH = ("--backend", "--d", "--layers", "--dropout", "--generator", "--workload", "--e2e-windows", "--adj-type")
_PASS_FLAGS = ("--share-gpu", "--dry-run", "--hic-like", "--p2p-allreduce", "--no-hip-graph", "--no-group-graph")


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _alive(pid):
    try:
        with open("/proc/%d/stat" % pid) as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except OSError:
        return False


def _descendants(root):
    """pids of every descendant of `root` (exact process tree from /proc: no pattern matching, no third-party module --
    the timeout path is the one case the ladder exists for and must not depend on an optional import)"""
    kids = {}
    for ent in os.listdir("/proc"):
        if not ent.isdigit():
            continue
        try:
            with open("/proc/%s/stat" % ent) as f:
                ppid = int(f.read().rsplit(")", 1)[1].split()[1])
        except (OSError, ValueError, IndexError):
            continue
        kids.setdefault(ppid, []).append(int(ent))
    out, todo = [], [root]
    while todo:
        for k in kids.get(todo.pop(), []):
            out.append(k)
            todo.append(k)
    return out


def _child_job(gpus, argv, timeout_s, relay_stderr=True):
    """Run `bench.py argv` as a child torch.distributed.run job of `gpus` ranks in a process group of its own; returns
    (rc, rank 0's JSON line or None, seconds, why).  A job that outlives `timeout_s` is killed (the whole process group
    we started, by its id) and reported as a timeout."""
    import signal
    import subprocess
    import threading
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
                        "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
                        "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING",
                        "TORCHELASTIC_ERROR_FILE")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    t0 = time.perf_counter()
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    got = {"line": None}

    def pump():
        for ln in proc.stdout:                 # rank 0's JSON line is the result; anything else goes to stderr
            if ln.startswith("{"):
                got["line"] = ln
            elif relay_stderr:
                sys.stderr.write(ln)
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    why = None
    try:
        rc = proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        why = "no result within %.0f s (killed)" % timeout_s
        # torch.distributed.run puts every worker into a session of its own, so the job is not one process group: take
        # the exact descendants of the launcher we started (by pid), ask the launcher to stop (SIGTERM: its agent
        # terminates the workers), then kill whatever of them is still alive -- a rank stuck in a collective may
        # ignore SIGTERM, and a rank left behind would keep its GPU
        kids = _descendants(proc.pid)
        proc.send_signal(signal.SIGTERM)
        try:
            proc.wait(timeout=float(os.environ.get("CGCN_BENCH_TERM_GRACE_S", "10")))
        except subprocess.TimeoutExpired:
            pass
        for pid in kids:
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass
        if proc.poll() is None:
            proc.kill()
        rc = proc.wait()
        t_end = time.time() + 10
        while time.time() < t_end and any(os.path.exists("/proc/%d" % pid) and _alive(pid) for pid in kids):
            time.sleep(0.1)
    th.join(5)
    if why is None and rc != 0:
        why = "exit code %d" % rc
    if why is None and got["line"] is None:
        why, rc = "exited 0 without a result line", 1
    return rc, got["line"], time.perf_counter() - t0, why


def _carried_args():
    """the caller's arguments a probe / measured child must share (workload shape, backend, testing switches)"""
    out, av, i = [], sys.argv[1:], 0
    while i < len(av):
        a = av[i]
        if a in _PASS_THROUGH and i + 1 < len(av):
            out += [a, av[i + 1]]
            i += 2
            continue
        if a.split("=")[0] in _PASS_THROUGH or a in _PASS_FLAGS:
            out.append(a)
        i += 1
    return out


def choose_rung(args, first=0):
    """Probe the ladder from rung `first`; returns (rung index or None, record for the JSON line)."""
    probe_tmo = float(os.environ.get("CGCN_BENCH_PROBE_TIMEOUT_S", "150"))
    rec = {"rungs": [r[0] for r in LADDER], "tried": []}
    user = _carried_args()
    for i in range(first, len(LADDER)):
        flags = [f for f in LADDER[i][1] if f not in user or f in ("--gather", "all")]
        argv = ["--gpus", str(args.gpus), "--probe", "--rung", str(i), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                "--no-extras", "--no-roofline"] + user + flags
        rc, line, secs, why = _child_job(args.gpus, argv, probe_tmo)
        rec["tried"].append({"rung": i, "probe_s": round(secs, 1), "ok": why is None, "why": why})
        if why is not None and "killed" not in why and secs < 45.0:
            # a probe that DIES quickly (not one that hangs) may have lost a race for its rendezvous port or met a
            # transient start-up error: the same rung once more, on a fresh port, before the configuration is given up
            rc, line, secs, why = _child_job(args.gpus, argv, probe_tmo)
            rec["tried"].append({"rung": i, "probe_s": round(secs, 1), "ok": why is None, "why": why, "retry": True})
        if why is None:
            rec["rung"] = i
            return i, rec
    rec["rung"] = None
    return None, rec


def _rung_argv(args, rung):
    """the measured job's arguments: the caller's, plus the rung's switches"""
    av = [a for a in sys.argv[1:]]
    for f in LADDER[rung][1]:
        if f == "all":
            continue
        if f == "--gather":
            if "--gather" in av:
                k = av.index("--gather")
                del av[k:k + 2]
            av += ["--gather", "all"]
        elif f not in av:
            av.append(f)
    return av + ["--rung", str(rung)]


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment (README.md:34 runs the reference with ONE
    command): start the N ranks ourselves -- `python -m torch.distributed.run`, one process per GPU, rendezvous on
    127.0.0.1 -- BEFORE anything in this process touches the GPU (a process that has initialised HIP must neither fork
    GPU work nor exec), relay rank 0's JSON line and the job's exit code.  The measured job runs on the first rung of
    LADDER whose probe job came back; if the measured job itself fails or hangs, the next rung is tried.  Under a
    launcher (WORLD_SIZE set) this is a no-op and the process is one rank (see external_ladder)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    if not args.share_gpu and not args.dry_run:
        have = torch.cuda.device_count()   # counts devices without initialising the runtime
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (use --share-gpu --backend gloo for a "
                             "single-GPU functional run)" % (args.gpus, have))
    job_tmo = float(os.environ.get("CGCN_BENCH_JOB_TIMEOUT_S", "1500"))
    if os.environ.get("CGCN_BENCH_LADDER", "1") == "0" or args.rung is not None:
        rc, line, _, why = _child_job(args.gpus, sys.argv[1:], job_tmo)
        if line is not None:
            sys.stdout.write(line)
            sys.stdout.flush()
        if why:
            sys.stderr.write("bench.py: the job failed: %s\n" % why)
        return rc
    first, history = 0, []
    while first < len(LADDER):
        rung, rec = choose_rung(args, first)
        history += rec["tried"]
        if rung is None:
            break
        rc, line, secs, why = _child_job(args.gpus, _rung_argv(args, rung), job_tmo)
        if why is None:
            d = json.loads(line)
            d["launch_ladder"] = {"rung": rung, "ran": LADDER[rung][0], "tried": history, "measured_job_s": round(secs, 1),
                                  "launcher": "bench.py (self-launched torch.distributed.run children)"}
            sys.stdout.write(json.dumps(d) + "\n")
            sys.stdout.flush()
            return 0
        history.append({"rung": rung, "measured_job": True, "ok": False, "why": why})
        sys.stderr.write("bench.py: rung %d (%s) passed its probe but the measured job failed: %s; trying the next rung\n"
                         % (rung, LADDER[rung][0], why))
        first = rung + 1
    sys.stderr.write("bench.py: no rung of the launch ladder produced a result: %s\n" % json.dumps(history))
    return 1


_JOB_STORE = [None]


def job_store(rank, world):
    """The ONE c10d store of this process: a client of the launcher's TCP store (torch.distributed.run hosts it in its
    agent: TORCHELASTIC_USE_AGENT_STORE=True), or -- ranks started by hand with MASTER_ADDR / MASTER_PORT -- hosted by
    rank 0.  The launch ladder hands its choice over it and init_group builds the process group on a prefix of it."""
    if _JOB_STORE[0] is None:
        import datetime
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
        _JOB_STORE[0] = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), world,
                                      is_master=(rank == 0 and not agent), timeout=datetime.timedelta(seconds=1800),
                                      wait_for_workers=False, multi_tenant=True)
    return _JOB_STORE[0]


def init_group(backend, rank, world, dev=None, timeout_s=None):
    """The default process group, initialised ONCE per process, on a prefix of the job's store."""
    import datetime
    if dist.is_initialized():
        raise RuntimeError("bench.py: the default process group is initialised once per process")
    k = int(os.environ.get("CGCN_BENCH_TEST_SKEW_RANK", "-1"))   # test hook: one rank arrives late at the rendezvous
    if k == rank:
        time.sleep(float(os.environ.get("CGCN_BENCH_TEST_SKEW_S", "3")))
    kw = {"timeout": datetime.timedelta(seconds=timeout_s)} if timeout_s else {}
    if backend == "nccl":
        kw["device_id"] = dev
    dist.init_process_group(backend, store=dist.PrefixStore("cgcn_pg", job_store(rank, world)), rank=rank, world_size=world, **kw)


def external_ladder(args, world, rank):
    """Under an external launcher (the driver's `python -m torch.distributed.run ... bench.py --gpus N`): the ranks
    cannot be restarted, so the rung is chosen BEFORE any of them touches a GPU -- all ranks meet on a gloo group (CPU
    only), rank 0 runs the probe jobs as children on the still untouched GPUs, the choice is broadcast, the gloo group
    is torn down and the ranks go on to build the real (RCCL) group with the chosen switches.  CGCN_BENCH_LADDER=0
    skips the probes: then the conservative form runs (eager all-reduce, all-gather) unless CGCN_GROUP_GRAPH=1."""
    if world <= 1 or args.rung is not None or args.probe:
        return None
    if os.environ.get("CGCN_BENCH_LADDER", "1") == "0":
        if os.environ.get("CGCN_GROUP_GRAPH") != "1":
            args.no_group_graph = True
            if "--gather" not in sys.argv:
                args.gather = "all"
        return {"rung": None, "ran": "no probes (CGCN_BENCH_LADDER=0): " + ("as asked" if os.environ.get("CGCN_GROUP_GRAPH") == "1" else LADDER[1][0]),
                "launcher": "external"}
    # The rung travels over the job's c10d STORE, not over a process group: the default group is initialised exactly once
    # per process (init_group below).  [Round 4 met on a gloo group, destroyed it and initialised the default group again
    # on the same store: destroy_process_group resets the group counter, the second group reused the first one's store
    # prefix and a rank that arrived early read its peer's dead gloo address -- VERDICT r4.]
    store = job_store(rank, world)
    key = "cgcn/ladder/%s" % os.environ.get("TORCHELASTIC_RUN_ID", "job")
    if rank == 0:
        rung, rec = choose_rung(args, 0)
        store.set(key, json.dumps(rec))
    else:
        import datetime
        store.wait([key], datetime.timedelta(seconds=float(os.environ.get("CGCN_BENCH_LADDER_WAIT_S", "3600"))))
    rec = json.loads(store.get(key).decode())
    rung = rec.get("rung")
    if rung is None:
        raise SystemExit("bench.py: no rung of the launch ladder passed its probe: %s" % json.dumps(rec["tried"]))
    for f in LADDER[rung][1]:
        if f == "--no-group-graph":
            args.no_group_graph = True
        elif f == "--no-hip-graph":
            args.no_hip_graph = True
        elif f == "--gather":
            args.gather = "all"
    return {"rung": rung, "ran": LADDER[rung][0], "tried": rec["tried"],
            "launcher": "external (rank 0 probed with child jobs before any rank touched a GPU)"}


def _one_rank_allreduce(dev):
    """--force-collectives: the one-rank group's all-reduce of ones (= 1 when the backend is up)"""
    t = torch.ones(1, device=dev, dtype=torch.float32)
    dist.all_reduce(t)
    return t.item()


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


MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector rate (dense)
# Split products (chromegcn_amd/csrc/cgcn_common.hpp): one fp32 product = six bf16 MFMA partial products, so the matrix roof of
# a kernel that runs them is the dense bf16 peak / 6, in fp32-EQUIVALENT flops (the algorithmic 2 m n k; MI355X_MICROARCH.md:
# ~2.5 PF dense bf16).  The kernels that have the form (d = 128): the row-local forward, the one-launch forward, the ring backward, the training head.
MFMA_BF16_PEAK_TFLOPS = 2500.0
MATRIX_SPLIT_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0
SPLIT_KERNELS = ("k_layer_dense", "k_layer_fwd", "k_bwd_rowlocal", "k_head_fused")


def layer_fwd_bytes(n, nnz, S, d):
    """Algorithmic HBM bytes of ONE fused-layer forward launch in training, SURVEY.md 8(d): CSR (rowptr + col) +
    1/deg + X read + parameters + X' write + gate write + the saved Z.  (The kernel also writes H = A X for the
    weight gradient -- a by-product of the (A X) W re-association that 8(d) does not list, so it is NOT counted.)"""
    return 4 * (n + 1) + 4 * nnz + 4 * n + S * 4 * n * d + (4 * d * d + 8 * d + 4) + S * 4 * n * d + S * 4 * n + S * 4 * n * d


# Algorithmic bytes / flops of ONE launch of each kernel of the train step (column "algorithmic
# bytes per launch"); n nodes, nnz = nnz(A + I), S strands, d features, C labels, P partial records.
def kernel_costs(n, nnz, S, d, C, P_rl, P_head):
    csr = 4 * (n + 1) + 4 * nnz + 4 * n
    t = S * 4 * n * d                      # one [S, n, d] tensor
    par = 4 * d * d + 8 * d + 4
    CP = 128 if C <= 128 else 256
    gat = 4.0 * nnz * S * d   # bytes of neighbour rows a gather kernel pulls through L2 / L1 (the L2-level traffic figure)
    costs = {
        "k_aggregate_sliced": (csr + 2 * t, 2.0 * nnz * S * d),                               # X in, H out
        "k_layer_dense": (4 * t + S * 4 * n + par, 2.0 * S * n * d * d),                      # H, X in; X', Z out; gate
        "k_layer_fwd": (csr + 4 * t + S * 4 * n + par, 2.0 * nnz * S * d + 2.0 * S * n * d * d),   # X in; X', Z, H out
        # (the per-workgroup partial records -- P_rl x (d^2 + 2d + 4) floats of dW / db / dwg / dcg, P_head x (CP d + CP + 8 d)
        # of the head -- are what THIS implementation writes for its deterministic two-stage sums: implementation bytes,
        # not algorithmic ones; the algorithm's own output there is one d x d (+ 2d + 1) / C x d (+ C + 4d) gradient)
        "k_bwd_rowlocal": (5 * t + S * 4 * n + par + (4 * d * d + 8 * d + 4), 4.0 * S * n * d * d),   # Z, X, H, dXn, W in; dHs, dW/db/dwg/dcg out
        "k_bwd_rowlocal(head)": (4 * t + 4 * n * d + t + S * 4 * n + par + (4 * d * d + 8 * d + 4), 4.0 * S * n * d * d),   # dym [n,d] in, dL/dXn out
        "k_bwd_sliced": (4 * (n + 1) + 4 * nnz + 3 * t + S * 4 * n, 2.0 * nnz * S * d),      # dHs, dXn in; dX out
        "k_head_fused": (t + 2 * 4 * n * C + 4 * n * d + 2 * 4 * (C * d + C) + 4 * 4 * d, 6.0 * n * d * C),   # X, targets, W_out in; probs, dym, dW_out/db_out/BN sums out
        # band graphs ('constant'): the window sums stream the table once -- no index list among the algorithmic bytes
        "k_band_aggregate": (4 * n + 2 * t, 2.0 * nnz * S * d),                               # 1/deg, X in; H out
        "k_bwd_band": (3 * t + S * 4 * n, 2.0 * nnz * S * d),                                 # dHs, dXn, gate in; dX out
    }
    return {k: (v[0], v[1], gat if k in ("k_aggregate_sliced", "k_layer_fwd", "k_bwd_sliced") else 0.0) for k, v in costs.items()}


def time_kernels(stage, name, reps, dropout_p):
    """Average duration (seconds) of every kernel of one chromosome's train step, each launched in isolation `reps`
    times back to back and bracketed by HIP events on the stream the library launches on (torch's current stream):
    the C ABI runs the forward's two launches as cgcn_spmm (k_aggregate_sliced on tables >= 6 MiB) and cgcn_layer_fwd
    with H_in (k_layer_dense), the backward's and the head's through the cgcn_debug_*_phases hooks.
    Returns {kernel: (seconds per launch, launches per train step)}."""
    import ctypes
    from chromegcn_amd import _lib
    c = stage.chroms[name]
    m = stage.model
    g = c.graph
    S, n, d = c.x.shape
    C = c.target.shape[1]
    dev = c.x.device
    lib = _lib.load()
    P, st = _lib.ptr, _lib.stream_ptr
    xn, z, h, dx, dhs = (torch.empty_like(c.x) for _ in range(5))
    gate = torch.empty(S, n, device=dev)
    rng = m._rng_state
    rows = ctypes.c_int(0)
    # the statistics mode the ENGINE uses for this chromosome (accumulate where its features are in range: GCNStage._stat_acc_for)
    acc = bool(getattr(c, "stat_acc", False))
    tiles = lib.cgcn_layer_fwd_colstats_plan(n, S, d, _lib.COLSTATS_ACCUMULATE if acc else _lib.COLSTATS_RECORDS, ctypes.byref(rows))
    acc = rows.value == -1
    colstats = torch.zeros((tiles, S, d, 2), device=dev)
    # the engine's last layer (ChromeGCN.forward_loss): with a layer in front of it, THAT layer's first launch zeroes the totals
    # and the last layer accumulates on the route its table size gives it (colstats_rows = -3); a one-layer model zeroes in its own
    # aggregation launch (-1: always two launches)
    prezero = acc and m.n_layers > 1
    rows_last = _lib.COLSTATS_ROWS_ACCUMULATE_ZEROED if prezero else rows.value
    L = m.n_layers
    gc1, w1, gcL, wL, bn, out = m.GC1, m.W1, getattr(m, "GC%d" % L), getattr(m, "W%d" % L), m.batch_norm, m.out
    drop = dropout_p > 0
    from chromegcn_amd.graph import aux_ptr
    c16, c16t = aux_ptr(g.col), aux_ptr(g.col_t)   # the engine's own cgcn_graph_aux (16-bit indices, longest row)
    # the route the LIBRARY takes for this graph (table size against the current threshold, hub-heavy graphs): asked,
    # not re-derived here -- a hub graph on a small table runs k_aggregate_sliced + k_layer_dense, not k_layer_fwd
    route = lib.cgcn_debug_layer_fwd_route(n, S, d, c16, 0)           # the layers without column statistics
    route_last = lib.cgcn_debug_layer_fwd_route(n, S, d, c16, rows_last)    # the last layer, with its column statistics
    band = route == 2 and g.val is None      # the sliding-window kernels (k_band_aggregate / k_bwd_band)
    split, split_last = route in (1, 2), route_last in (1, 2)

    def ev_time(fn):
        for _ in range(3):
            _lib.check(fn(), "roofline launch")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    def fwd(layer, h_in, hbuf, cs):   # layer 1: inter-layer dropout; last layer: BatchNorm column statistics
        gc, wk = (gc1, w1) if layer == 1 else (gcL, wL)
        last = layer == L
        return lib.cgcn_layer_fwd(st(), n, S, d, P(g.rowptr), P(g.col), P(g.val), P(g.row_scale), c.x.data_ptr(),
                                  gc.weight.data_ptr(), gc.bias.data_ptr(), wk.weight.data_ptr(), wk.bias.data_ptr(),
                                  xn.data_ptr(), z.data_ptr(), P(hbuf), gate.data_ptr(), 0.0 if (last or not drop) else float(dropout_p),
                                  None if (last or not drop) else P(rng), layer, P(h_in), cs.data_ptr() if (last and cs is not None) else None,
                                  rows_last if (last and cs is not None) else 0, c16)

    out_t = {}
    agg = lambda: lib.cgcn_spmm(st(), n, n, S, d, P(g.rowptr), P(g.col), P(g.val), P(g.row_scale), c.x.data_ptr(), h.data_ptr(), c16)
    # L forward launches per step: L - 1 in the inter-layer-dropout form, the last with the column statistics (whose
    # accumulate mode takes the two-launch route at every table size: the aggregation launch zeroes the totals)
    t_d2 = None
    n_agg, n_dense, t_dense = 0, 0, 0.0
    if split and L > 1:
        t_dense += (L - 1) * ev_time(lambda: fwd(1, h, None, None))
        n_agg += L - 1
        n_dense += L - 1
    elif L > 1:
        out_t["k_layer_fwd"] = (ev_time(lambda: fwd(1, None, h, None)), L - 1)
    if split_last:
        _lib.check(agg(), "aggregate")
        t_d2 = ev_time(lambda: fwd(L, h, None, colstats))   # (accumulate mode, H_in form: + the memset node that zeroes the totals)
        t_dense += t_d2
        n_agg += 1
        n_dense += 1
    else:
        t_f2 = ev_time(lambda: fwd(L, None, h, colstats))
        prev = out_t.get("k_layer_fwd", (0.0, 0))
        out_t["k_layer_fwd"] = ((prev[0] * prev[1] + t_f2) / (prev[1] + 1), prev[1] + 1)
    if n_agg:
        out_t["k_band_aggregate" if band else "k_aggregate_sliced"] = (ev_time(agg), n_agg)
        out_t["k_layer_dense"] = (t_dense / n_dense, n_dense)
    # ---- head: cgcn_head_train once in full (valid state for the phases and for the backward's head mode), then k_head_fused alone
    hws_b = lib.cgcn_head_workspace_bytes(n, S, d, C)
    hws = torch.empty(hws_b, dtype=torch.uint8, device=dev)
    probs, loss = torch.empty(n, C, device=dev), torch.empty(1, device=dev)
    sm, si = torch.empty(S, d, device=dev), torch.empty(S, d, device=dev)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()

    def head(ph):
        return lib.cgcn_debug_head_train_phases(st(), n, S, d, C, xn.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), rm.data_ptr(),
                                                rv.data_ptr(), None, 0.1, 1e-5, out.weight.data_ptr(), out.bias.data_ptr(),
                                                c.target.data_ptr(), float(dropout_p) if drop else 0.0, P(rng) if drop else None,
                                                probs.data_ptr(), loss.data_ptr(), sm.data_ptr(), si.data_ptr(), colstats.data_ptr(), tiles,
                                                rows.value, hws.data_ptr(), hws_b, ph)
    def refill():   # the last layer's forward on the aggregation already in h: fresh column statistics (accumulate mode: zeroed totals)
        if prezero:
            colstats.zero_()   # (in the step: done by the previous layer's first launch)
        return fwd(L, h if split_last else None, None if split_last else h, colstats)
    _lib.check(refill(), "fwd")
    _lib.check(head(7), "head")
    if acc:
        # accumulate mode: the main kernel ADDS its backward sums and draws a ticket from the buffer -- every timed launch
        # needs freshly zeroed and refilled totals in front of it: the pair is timed and the refill (timed above) subtracted.
        # k_head_bn_finalize / k_head_train_finish are not launched in this mode and not listed.
        def pair():
            rc = refill()
            return rc if rc else head(2)
        t_refill = ev_time(refill)
        out_t["k_head_fused"] = (max(ev_time(pair) - t_refill, 0.0), 1)
        _lib.check(refill(), "fwd")
        _lib.check(head(7), "head")   # valid state for the backward's head mode
    else:
        out_t["k_head_fused"] = (ev_time(lambda: head(2)), 1)
        out_t["k_head_bn_finalize"] = (ev_time(lambda: head(1)), 1)
        out_t["k_head_train_finish"] = (ev_time(lambda: head(4)), 1)
    # ---- backward: row-local launch (head mode = last layer, plain = first layer), then the sliced gather launch
    o_dym, o_bnc, o_part = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(lib.cgcn_head_workspace_layout(n, S, d, C, ctypes.byref(o_dym), ctypes.byref(o_bnc), ctypes.byref(o_part)), "layout")
    one = torch.ones(1, device=dev)
    dW_out, db_out = torch.empty_like(out.weight), torch.empty(C, device=dev)
    dbn_w, dbn_b = torch.empty(d, device=dev), torch.empty(d, device=dev)
    hg = _lib.HeadGrad(hws.data_ptr() + o_dym.value, hws.data_ptr() + o_bnc.value, sm.data_ptr(), si.data_ptr(), bn.weight.data_ptr(),
                       float(dropout_p) if drop else 0.0, P(rng) if drop else None, hws.data_ptr() + o_part.value,
                       lib.cgcn_head_bwd_partials(n), C, dW_out.data_ptr(), db_out.data_ptr(), 0, one.data_ptr(), dbn_w.data_ptr(), dbn_b.data_ptr(),
                       colstats.data_ptr() if acc else None)   # accumulate mode: the backward sums live in the statistics buffer
    ws_b = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
    ws = torch.empty(ws_b, dtype=torch.uint8, device=dev)
    dW, db, dwg, dcg = torch.empty(d, d, device=dev), torch.empty(d, device=dev), torch.empty(d, device=dev), torch.empty(1, device=dev)
    dxn = torch.randn_like(c.x) * 1e-6

    def bwd(ph, head_mode):
        gc, wk = (gcL, wL) if head_mode else (gc1, w1)
        return lib.cgcn_debug_layer_bwd_phases(st(), n, S, d, P(g.rowptr_t), P(g.col_t), P(g.val_t), P(g.row_scale), c.x.data_ptr(), z.data_ptr(),
                                               h.data_ptr(), gate.data_ptr(), gc.weight.data_ptr(), wk.weight.data_ptr(),
                                               None if head_mode else dxn.data_ptr(), None, dx.data_ptr(), dhs.data_ptr(), dW.data_ptr(),
                                               db.data_ptr(), dwg.data_ptr(), dcg.data_ptr(), 0, float(dropout_p) if (head_mode and drop and L > 1) else 0.0,
                                               P(rng) if drop else None, max(L - 1, 0) if head_mode else 0,
                                               ctypes.byref(hg) if head_mode else None, ws.data_ptr(), ws_b, ph, c16t)
    _lib.check(bwd(3, True), "bwd")
    rl = "k_bwd_rowlocal_ring" if lib.cgcn_debug_layer_bwd_route(n, S, d) == 2 else "k_bwd_rowlocal"
    out_t[rl + "(head)"] = (ev_time(lambda: bwd(1, True)), 1)
    out_t[rl] = (ev_time(lambda: bwd(1, False)), max(L - 1, 0))
    _lib.check(bwd(3, False), "bwd")
    out_t["k_bwd_band" if band else "k_bwd_sliced"] = (ev_time(lambda: bwd(2, False)), L)
    torch.cuda.synchronize()
    return out_t, (lib.cgcn_layer_bwd_workspace_bytes(n, S, d) // ((d * d + 2 * d + 4) * 4), lib.cgcn_head_bwd_partials(n))


def stored_traffic(key, kernel=None):
    """Bytes beyond L2 per launch from the PMC passes of an earlier profiling run (profiles/traffic.json;
    FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE), for one kernel of one workload.  A stored, offline value:
    the bench line says so and names the tag."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(tpath))
    except Exception:
        return None, None
    # the counters were collected on ONE build of the library: a stored value says nothing about kernels that have
    # changed since, so it is only reported for the very sources it was measured on (content hash, _build.source_hash)
    from chromegcn_amd import _build
    if t.get("_src_hash") != _build.source_hash([]):
        return None, "stale"
    ent = t.get(key)
    if not isinstance(ent, dict):
        return None, None
    if kernel is not None:
        for k, v in (ent.get("per_kernel") or {}).items():
            if k.startswith("void " + kernel.split("(")[0] + "<") or k.startswith(kernel.split("(")[0]):
                return v.get("bytes_per_launch"), ent.get("tag")
        return None, ent.get("tag")
    return ent.get("bytes_per_launch"), ent.get("tag")


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


def pcie_link_info(dev):
    """current PCIe link speed / width of the GPU as sysfs reports it (None where it is not readable)"""
    try:
        import glob
        bus = torch.cuda.get_device_properties(dev).pci_bus_id if hasattr(torch.cuda.get_device_properties(dev), "pci_bus_id") else None
        out = []
        for p in sorted(glob.glob("/sys/class/drm/card*/device")):
            try:
                with open(os.path.join(p, "current_link_speed")) as f:
                    sp = f.read().strip()
                with open(os.path.join(p, "current_link_width")) as f:
                    wd = f.read().strip()
                out.append({"device": os.path.basename(os.path.realpath(p)), "speed": sp, "width": wd})
            except OSError:
                continue
        return {"pci_bus_id": bus, "links": out} if out else None
    except Exception:  # pragma: no cover
        return None


def band_roofline(stage, names, d):
    """The band route's aggregation kernel against the HBM roof: k_band_aggregate (through cgcn_spmm on the stage's own
    'constant' graphs) timed alone with HIP events, 50 launches per chromosome; algorithmic bytes = 1/deg + X in, H out
    (SURVEY 8d's per-SpMM figure without an index list: a band has none)."""
    from chromegcn_amd import _lib
    from chromegcn_amd.graph import aux_ptr, is_band
    lib = _lib.load()
    P, st = _lib.ptr, _lib.stream_ptr
    tot_s = tot_b = 0.0
    launches = 0
    for nm in names:
        c = stage.chroms[nm]
        g = c.graph
        if not is_band(g.col):
            return None
        S, n, _ = c.x.shape
        h = torch.empty_like(c.x)
        fn = lambda: lib.cgcn_spmm(st(), n, n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), c.x.data_ptr(), h.data_ptr(), aux_ptr(g.col))
        for _ in range(3):
            _lib.check(fn(), "band launch")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        e1.synchronize()
        tot_s += e0.elapsed_time(e1) / 50 * 1e-3
        tot_b += 4.0 * n + 2.0 * S * 4 * n * d
        launches += 1
    gbps = tot_b / tot_s / 1e9
    return {"kernel": "k_band_aggregate (adj_type constant: +-7 band + I as a sliding-window stream)", "bound": "hbm", "achieved": gbps,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "avg_kernel_us": tot_s / launches * 1e6,
            "algorithmic_bytes_per_launch": tot_b / launches, "traffic": None,
            "note": "mean over the %d train chromosomes, each launched alone 50 times (HIP events); input rows hot in the Infinity Cache "
                    "between launches (the table is 6-31 MB), as they are inside a train step" % launches}


def saliency_numbers(stage, names, d):
    """the adjacency saliency of scripts/visualize.py:29-55 on the pattern.  (i) `saliency_ms`: the whole
    analysis of one chromosome (both strands forward, backward of sigmoid(pred) . targets, one cgcn_sddmm per layer, the row
    normalisation) with chromegcn_amd.saliency.adjacency_saliency, wall time incl. its host work, mean over the chromosomes;
    (ii) k_sddmm alone (HIP events, 30 isolated launches per chromosome) against the HBM roof by algorithmic bytes -- rowptr +
    col + the two [S, n, d] tables once + one float per entry out -- and its gather rate (nnz S d 4 bytes of neighbour rows
    through the vector L1s; the row of A stays in registers)."""
    from chromegcn_amd import _lib
    from chromegcn_amd.saliency import adjacency_saliency
    lib = _lib.load()
    P, st = _lib.ptr, _lib.stream_ptr
    m = stage.model
    was_training = m.training
    m.eval()
    t_sal, tot_s, tot_b, tot_g, launches = 0.0, 0.0, 0.0, 0.0, 0
    try:
        for nm in names:
            c = stage.chroms[nm]
            g = c.graph
            S, n, _ = c.x.shape
            for rep in range(3):                 # the last two timed
                if rep == 1:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                adjacency_saliency(m, c.x[0], c.x[1], g, c.target)
            torch.cuda.synchronize()
            t_sal += (time.perf_counter() - t0) / 2
            a = torch.randn_like(c.x)
            out = torch.empty(g.col.shape[0], device=c.x.device)
            fn = lambda: lib.cgcn_sddmm(st(), n, S, d, P(g.rowptr), P(g.col), a.data_ptr(), c.x.data_ptr(), out.data_ptr(), 0)
            for _ in range(3):
                _lib.check(fn(), "sddmm launch")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                fn()
            e1.record()
            e1.synchronize()
            tot_s += e0.elapsed_time(e1) / 30 * 1e-3
            tot_b += 4.0 * (n + 1) + 4.0 * g.nnz + 2.0 * S * 4 * n * d + 4.0 * g.nnz
            tot_g += 4.0 * g.nnz * S * d
            launches += 1
    finally:
        m.train(was_training)
    gbps = tot_b / tot_s / 1e9
    return t_sal / launches * 1e3, {
        "kernel": "k_sddmm (dL/dA on the sparsity pattern: one wave per row, whole-row gather of the neighbours' features)", "bound": "hbm",
        "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "avg_kernel_us": tot_s / launches * 1e6,
        "algorithmic_bytes_per_launch": tot_b / launches, "gathered_bytes_per_launch": tot_g / launches,
        "gather_GBps": tot_g / tot_s / 1e9, "traffic": None,
        "note": "mean over %d chromosome(s), each launched alone 30 times (HIP events); one launch per gated layer in the analysis" % launches}


def physical_cores():
    """distinct (socket, core) pairs of /proc/cpuinfo (None where it does not say)"""
    try:
        seen, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip() and phys is not None and core is not None:
                    seen.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
        return len(seen) or None
    except OSError:
        return None


def cpu_baseline(args, chroms, budget_s, full=None):
    """The oracle (torch-CPU restatement of the reference ops, finetune.py:29-53) timed on this box's host cores: train steps,
    adjacency cached (the reference re-normalises it every chromosome every epoch, finetune.py:36 -- reported separately).
    chroms: a BOUNDED sample of the workload, on which the thread count is chosen (torch's CPU spmm does not scale to every
    core of a big host).  full: the WHOLE workload `value` is quoted on (the genome's 16 train chromosomes): at least three
    passes over it at the chosen thread count are what the reported figure comes from; the sample's own
    figure stays as `sample_value`.  full=None (single-chromosome workloads): the sample is the workload."""
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
        O.finetune_epoch(model, data, graphs, opt, "train", args.adj_type, adj_cache=cache if cached else None)
        return time.perf_counter() - t0

    # torch's CPU spmm does not scale to every core of a big host (256 threads measured 20x slower than 8-32): pick
    # the thread count that is fastest on this box, so the baseline is the CPU's best case.
    best = None
    for th in sorted({t for t in (4, 8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        one()  # warm-up at this thread count (the first call also builds the cached adjacency)
        dt = one()
        if best is not None and dt > 2.0 * best[1]:
            break    # far off the best already: no point in averaging this setting (or trying more threads)
        dt = float(np.median([dt, one(), one()]))   # three passes per setting: one pass picked 8 threads on one box, 32 on another
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
    out = {"value": n_sample / t_cached, "unit": "windows/s", "cores": cores, "threads": cores, "kind": "port",
           "sample": "%d passes over %s (%d windows; f+r fwd, BCE, bwd incl. d/dx, SGD step per chromosome), "
                     "adjacency cached; oracle = torch-CPU restatement of the reference ops; median" %
                     (reps, "+".join(c[0] for c in chroms), n_sample),
           "s_per_pass": t_cached, "p10_s": float(np.percentile(ts, 10)), "p90_s": float(np.percentile(ts, 90)),
           "with_process_graph_windows_per_s": n_sample / t_full, "host_cpus": ncpu, "physical_cores": physical_cores(),
           "cpu_model": host_info(), "torch": torch.__version__}
    if full:
        # the whole workload at the thread count chosen above: one untimed pass (builds the cached adjacencies), then >= 3 timed
        for nm, n, pairs, seed in full:
            if nm not in data:
                data[nm] = synth.chrom_features(n, args.d, synth.N_LABELS, 1000 + seed)
                graphs[nm] = synth.contact_graph(n, pairs, seed, args.hic_like)
        order = [c[0] for c in full]
        data_f, graphs_f = {k: data[k] for k in order}, {k: graphs[k] for k in order}
        n_full = sum(c[1] for c in full)

        def epoch():
            t0 = time.perf_counter()
            O.finetune_epoch(model, data_f, graphs_f, opt, "train", args.adj_type, adj_cache=cache)
            return time.perf_counter() - t0
        epoch()
        tf = [epoch() for _ in range(3)]
        while sum(tf) < budget_s / 2 and len(tf) < 10:
            tf.append(epoch())
        t_ep = float(np.median(tf))
        out.update({"sample_value": out["value"], "sample_s_per_pass": t_cached, "value": n_full / t_ep, "s_per_pass": t_ep,
                    "p10_s": float(np.percentile(tf, 10)), "p90_s": float(np.percentile(tf, 90)),
                    "sample": "%d passes over ALL %d train chromosomes of the workload (%d windows; f+r fwd, BCE, bwd incl. d/dx, SGD step per "
                              "chromosome), adjacency cached; oracle = torch-CPU restatement of the reference ops; median; %d threads "
                              "(the fastest of 4 ... %d on %s)" % (len(tf), len(full), n_full, cores, ncpu, "+".join(c[0] for c in chroms))})
    return out


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
    # test hooks of the launch ladder (tests/test_bench_launcher.py): a rung that fails / hangs on purpose
    if args.rung is not None:
        def _in(var):
            return str(args.rung) in [v for v in os.environ.get(var, "").split(",") if v]
        if _in("CGCN_BENCH_FAIL_RUNGS") or (not args.probe and _in("CGCN_BENCH_FAIL_MEASURED_RUNGS")):
            raise SystemExit("bench.py: rung %d told to fail (test hook)" % args.rung)
        if args.probe and _in("CGCN_BENCH_HANG_RUNGS"):
            time.sleep(3600)
    ladder_rec = external_ladder(args, world, rank) if "WORLD_SIZE" in os.environ else None
    if args.dry_run:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            init_group("gloo", rank, world)
        seen = ranks_seen(world, torch.device("cpu"))
        out = {"dry_run": True, "n_gpus": world, "ranks_seen_by_backend": seen, "backend": "gloo", "probe": bool(args.probe),
               "rung": args.rung, "no_group_graph": bool(args.no_group_graph), "gather": args.gather, "hip_graph": not args.no_hip_graph,
               "launcher": "self (torch.distributed.run child)" if os.environ.get("TORCHELASTIC_RUN_ID") else "none"}
        if ladder_rec is not None:
            out["launch_ladder"] = ladder_rec
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
    multi = world > 1 or args.force_collectives   # the N > 1 code path (one rank: --force-collectives, testing only)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # one forced rank outside a launcher
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        # a collective that does not complete within 4 minutes aborts the job with an error instead of hanging it (every
        # collective of this benchmark moves at most a few tens of MB)
        init_group(args.backend, rank, world, dev, int(os.environ.get("CGCN_DIST_TIMEOUT_S", "240")))
    seen = int(_one_rank_allreduce(dev)) if (multi and world == 1) else ranks_seen(world, dev)
    if args.workload == "e2e":
        from chromegcn_amd import e2e
        out = e2e.bench(args, dev, world, rank)
        if rank == 0:
            print(json.dumps(out))
            sys.stdout.flush()
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return out

    import chromegcn_amd as C
    from chromegcn_amd import synth, _lib
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
    stage = GCNStage(model, opt, args.adj_type, dev, hip_graphs=not args.no_hip_graph, input_grad=True,
                     group=dist.group.WORLD if multi else None, cache_input_aggregation=False,
                     group_graph=False if args.no_group_graph else None, p2p_allreduce=True if args.p2p_allreduce else None,
                     prediction_gather=args.gather, force_collectives=bool(args.force_collectives),
                     epoch_graph=False if args.no_epoch_graph else None)

    if genome:
        names = genome_train_names()
        if args.probe:   # launch ladder: the first two step groups of the real epoch (its largest chromosomes), nothing more
            names = sorted(names, key=lambda c: -synth.chrom_nodes(c))[:2 * world]
        shapes = []
        for nm in names:  # N > 1: registered only -- a rank normalises and uploads the chromosomes the shard plan gives it
            feats, hic = synth.synthetic_chromosome(nm, d=args.d, hic_like=args.hic_like)
            stage.add_chromosome(nm, feats, hic, defer=multi)
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
            if multi:
                return stage.train_group(name, world)
            return stage.train_step(name)

    def fence():
        if multi:
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
        if multi:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, per, r

    for _ in range(max(warmup, 1)):
        step()
    elapsed, per, last = timed(step, steps)
    final_loss = float(last[2]) if genome else float(last[0].item())   # (a deferred device loss is read here, after the timed region)

    extras = {}
    if not args.no_extras:
        if genome and not multi:
            # the same epochs with the loss left on the device (run_split(sync_loss=False)): no host wait per epoch, so
            # the next epoch's launches are queued while this one still runs.  Reported beside the headline, which
            # keeps one host sync per epoch (the reference waits for the device once per chromosome, finetune.py:51).
            def step_nosync():
                return stage.run_split("train", names, to_cpu=False, sync_loss=False)
            step_nosync()
            ns_el, _, ns_last = timed(step_nosync, steps)
            extras["deferred_loss_sync_windows_per_s"] = windows * steps / ns_el
            extras["deferred_loss_sync_ms_per_step"] = ns_el / steps * 1e3
            extras["deferred_loss_sync_note"] = ("total loss returned as a device tensor and read after the timed region; "
                                                 "not the headline")
        if genome and not multi:
            # the advertised drop-in: finetune()'s return value has the predictions on the HOST (finetune.py:52-53,67).
            # run_split(to_cpu=True) sends every chromosome's rows to a pinned host arena on a copy stream behind its
            # own step, under the next chromosome's kernels (finetune._HostArena); one sync at the end of the epoch.
            def step_cpu():
                return stage.run_split("train", names)
            for _ in range(2):
                step_cpu()
            c_el, c_per, c_last = timed(step_cpu, steps)
            extras["dropin_finetune_windows_per_s"] = windows * steps / c_el
            extras["dropin_finetune_ms_per_step"] = c_el / steps * 1e3
            # what the difference to the headline is made of: the copies are ordered behind their steps and overlap the next
            # chromosomes' kernels, so the exposed part is the last group's copy plus whatever the link cannot hide.  The
            # achieved device-to-host rate of this box, measured on the same pinned arena (one copy of the whole prediction
            # matrix, 5 repeats), and the PCIe link as the kernel reports it, say whether the link is the limit.
            try:
                pin = torch.empty(c_last[0].shape, dtype=torch.float32, pin_memory=True)
                src = stage._arena["probs"][:pin.shape[0]]
                pin.copy_(src, non_blocking=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    pin.copy_(src, non_blocking=True)
                torch.cuda.synchronize()
                d2h_s = (time.perf_counter() - t0) / 5
                extras["dropin_d2h_GBps"] = pin.numel() * 4 / d2h_s / 1e9
                extras["dropin_d2h_ms_whole_matrix"] = d2h_s * 1e3
            except Exception as e:  # pragma: no cover
                extras["dropin_d2h_GBps"] = None
                extras["dropin_d2h_error"] = str(e)
            extras["dropin_exposed_copy_ms"] = (c_el - elapsed) / steps * 1e3
            # the design hides every group's copy but the last under the next group's kernels: an exposed time of more than half
            # the whole matrix's own copy time means this box ran the copy stream's transfers one after the other with the
            # compute stream's graphs (seen on some leases: same link rate, 2.8 ms exposed instead of 0.3) -- a property of the
            # box's copy-engine scheduling, not of the link
            if extras.get("dropin_d2h_ms_whole_matrix"):
                extras["dropin_copies_overlap_compute"] = bool(extras["dropin_exposed_copy_ms"] < 0.5 * extras["dropin_d2h_ms_whole_matrix"])
            extras["dropin_over_headline"] = c_el / elapsed
            extras["pcie_link"] = pcie_link_info(dev)
            extras["dropin_finetune_note"] = ("chromegcn_amd.finetune.finetune()'s epoch: the same train epoch returning CPU predictions "
                                              "[%d x %d] fp32 (%.0f MB over PCIe per epoch, overlapped chromosome by chromosome), CPU targets "
                                              "(cached) and the summed loss, like the reference; not the headline (inputs AND outputs resident)"
                                              % (c_last[0].shape[0], c_last[0].shape[1], c_last[0].numel() * 4 / 1e6))
            # the same epoch on the other two contact generators (real top-K Hi-C is between them: distance decay + hubs)
            for gen in ("hic_like", "hub"):
                if gen == args.generator:
                    continue
                torch.manual_seed(0)
                m2 = C.ChromeGCN(args.d, args.d, synth.N_LABELS, args.dropout, True, args.layers).to(dev)
                o2 = torch.optim.SGD(m2.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
                s2 = GCNStage(m2, o2, "hic", dev, hip_graphs=not args.no_hip_graph, input_grad=True, cache_input_aggregation=False)
                for nm in names:
                    f2, h2 = synth.synthetic_chromosome(nm, d=args.d, hic_like=gen)
                    s2.add_chromosome(nm, f2, h2)
                for _ in range(3):
                    s2.run_split("train", names, to_cpu=False)
                g_el, _, _ = timed(lambda: s2.run_split("train", names, to_cpu=False), steps)
                extras["%s_generator_ms_per_step" % gen] = g_el / steps * 1e3
                extras["%s_generator_windows_per_s" % gen] = windows * steps / g_el
                del s2, m2, o2
            # the reference's other adjacency modes (-adj_type, config_args.py:45; utils/util_methods.py:146-174) on the same genome:
            # 'both' = Hi-C + the +-7 band + I with summed values (explicit-value kernels), 'constant' = the band + I alone
            # (recognised as a band: sliding-window kernels, no index list), 'none' = I
            for adj in ("both", "constant", "none"):
                if adj == args.adj_type:
                    continue
                torch.manual_seed(0)
                m2 = C.ChromeGCN(args.d, args.d, synth.N_LABELS, args.dropout, True, args.layers).to(dev)
                o2 = torch.optim.SGD(m2.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
                s2 = GCNStage(m2, o2, adj, dev, hip_graphs=not args.no_hip_graph, input_grad=True, cache_input_aggregation=False)
                for nm in names:
                    f2, h2 = synth.synthetic_chromosome(nm, d=args.d, hic_like=args.hic_like)
                    s2.add_chromosome(nm, f2, h2)
                for _ in range(3):
                    s2.run_split("train", names, to_cpu=False)
                g_el, _, _ = timed(lambda: s2.run_split("train", names, to_cpu=False), steps)
                extras["%s_ms_per_step" % adj] = g_el / steps * 1e3
                extras["%s_windows_per_s" % adj] = windows * steps / g_el
                if adj == "constant" and not args.no_roofline:
                    extras["constant_band_roofline"] = band_roofline(s2, names, args.d)
                del s2, m2, o2
            extras["adjacency_note"] = ("same genome and train epoch under the reference's other -adj_type settings (`value` is %s): both = "
                                        "Hi-C + band + I, values 1 / 2 (nnz about 1.45x hic's); constant = the +-7 band + I (15 entries per "
                                        "row), aggregated by the sliding-window kernels k_band_aggregate / k_bwd_band; none = I" % args.adj_type)
            extras["generators_note"] = ("same genome shape and train epoch on synth.contact_graph's other generators: hic_like = "
                                         "contact probability ~ 1 / distance, hub = top-K-style heavy-tailed degrees with 8 hubs of "
                                         "2 000 - 10 000 neighbours per chromosome; `value` is the %s generator" % args.generator)
        # the same workload with the dense products as the fp32 MFMA chain of rounds 1-5: a child process of this script with
        # CGCN_PRODUCTS=fp32 in its environment -- the same measurement protocol from a fresh process, like `value` itself
        # (timed inside this process behind the other extras, the chain form read 4.69 ms where a fresh process reads 4.23-4.30)
        if not multi and _lib.load().cgcn_debug_get_products() == 1 and args.d == 128:
            import subprocess
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-extras", "--no-roofline", "--no-cpu-baseline",
                                     "--steps", str(steps), "--warmup", str(warmup)] + _carried_args(),
                                    env=dict(os.environ, CGCN_PRODUCTS="fp32"), capture_output=True, text=True, timeout=600)
                cl = [l for l in cp.stdout.splitlines() if l.startswith("{")]
                cj = json.loads(cl[-1])
                assert cj["config"]["products"].startswith("fp32 MFMA chain")
                extras["fp32_chain_ms_per_step"] = cj["ms_per_step"]
                extras["fp32_chain_windows_per_s"] = cj["value"]
                extras["fp32_chain_note"] = ("the same workload, steps and warm-up in a child process with CGCN_PRODUCTS=fp32: the dense products on "
                                             "v_mfma_f32_16x16x4_f32 (the form of rounds 1-5; everything else as shipped); `value` is the split form")
            except Exception as e:  # pragma: no cover
                extras["fp32_chain_ms_per_step"] = None
                extras["fp32_chain_error"] = str(e)[:300]
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
        if genome and not multi:
            # the reference's compute_metrics on this split's predictions (runner.py:41 -> utils/evals.py:26),
            # on the device; ms per call incl. the one device-to-host copy of the per-label results
            from chromegcn_amd import metrics as _M
            pm, tm, _ = stage.run_split("train", names, to_cpu=False)
            for _ in range(2):
                _M.compute_metrics(pm, tm, 0.0, None, 0.0)
            torch.cuda.synchronize()
            t0m = time.perf_counter()
            for _ in range(5):
                _M.compute_metrics(pm, tm, 0.0, None, 0.0)
            torch.cuda.synchronize()
            extras["metrics_train_split_ms"] = (time.perf_counter() - t0m) / 5 * 1e3
            extras["metrics_note"] = ("chromegcn_amd.metrics.compute_metrics (AUROC / AUPR / recall@FDR50 / mAP per label) on the "
                                      "train split's %d x %d predictions, on the device: 32-bit keys, the library's segmented radix "
                                      "sort; not part of `value`" % (pm.shape[0], pm.shape[1]))
        if not multi:
            # the adjacency saliency on the pattern (scripts/visualize.py:29-55), per chromosome
            sal_names = [names[0], names[-1]] if genome else names[:1]
            extras["saliency_ms"], extras["sddmm_roofline"] = saliency_numbers(stage, sal_names, args.d)
            extras["saliency_note"] = ("chromegcn_amd.saliency.adjacency_saliency (|adj * d sum(sigmoid(pred) * targets) / d adj| on the CSR "
                                       "pattern, row-normalised) of one chromosome, mean over %s; the reference builds a dense n x n "
                                       "gradient on the CPU; not part of `value`" % "+".join(sal_names))
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
        # ---- roofline: every kernel of the train step timed live (HIP events, each kernel launched alone through the C
        # ABI / its profiling hooks on the chromosomes this rank holds); the DOMINANT kernel = the largest share of the
        # epoch's summed kernel time; roofline_top3 = the three largest, each against both roofs
        reps = 20 if genome else 100
        agg, forms = {}, {}
        products_split = _lib.load().cgcn_debug_get_products() == 1
        for nm, n, nnz in shapes:
            if nm not in stage.chroms or args.no_roofline:
                continue
            kt, (p_rl, p_head) = time_kernels(stage, nm, reps, args.dropout)
            costs = kernel_costs(n, stage.chroms[nm].graph.nnz, 2, args.d, synth.N_LABELS, p_rl, p_head)
            for k, (sec, per_step) in kt.items():
                # (one kernel, two forms: k_bwd_rowlocal with the head prologue -- last layer -- and without)
                e = agg.setdefault(k.split("(")[0], {"s": 0.0, "launches": 0, "bytes": 0.0, "flops": 0.0, "gather": 0.0})
                if per_step:
                    f = forms.setdefault(k, [0.0, 0])
                    f[0] += sec * per_step
                    f[1] += per_step
                e["s"] += sec * per_step
                e["launches"] += per_step
                ck = k.replace("k_bwd_rowlocal_ring", "k_bwd_rowlocal")   # same work, same algorithmic bytes / flops
                if ck in costs:
                    e["bytes"] += costs[ck][0] * per_step
                    e["flops"] += costs[ck][1] * per_step
                    e["gather"] += costs[ck][2] * per_step
        wl_key = (("genome" if genome else args.workload) + {"uniform": "", "hic_like": "_hic", "hub": "_hub"}[args.generator]
                  + ("" if args.adj_type == "hic" else "_" + args.adj_type) + "_d%d" % args.d)

        ga = agg.get("k_aggregate_sliced")
        gather_ref = (ga["gather"] / ga["s"] / 1e9) if (ga and ga["gather"] and ga["s"]) else None

        def roof_entry(k, e):
            if not e["launches"] or not e["bytes"]:
                return None
            us = e["s"] / e["launches"] * 1e6
            gbps, tfl = e["bytes"] / e["s"] / 1e9, e["flops"] / e["s"] / 1e12
            traffic, ttag = stored_traffic(wl_key, k)
            # the matrix roof of THIS kernel in the form it runs: the fp32 MFMA chain's, or -- split products, d = 128 -- the bf16
            # matrix cores' at six partial products per fp32 product (fp32-equivalent flops either way)
            split_k = products_split and args.d == 128 and k.split("(")[0].startswith(SPLIT_KERNELS)
            mpeak = MATRIX_SPLIT_PEAK_TFLOPS if split_k else MFMA_F32_PEAK_TFLOPS
            fh, fm = gbps / HBM_PEAK_GBPS, tfl / mpeak
            return {"kernel": k, "bound": "hbm" if fh >= fm else "mfma",
                    "achieved": gbps if fh >= fm else tfl, "peak": HBM_PEAK_GBPS if fh >= fm else mpeak,
                    "matrix_roof": ("bf16 MFMA dense peak / 6 (split products: six bf16 partial products per fp32 product)" if split_k
                                    else "fp32 MFMA (v_mfma_f32_16x16x4_f32)"), "matrix_peak_TFLOPs": mpeak,
                    "unit": "GB/s" if fh >= fm else "TFLOP/s", "frac": max(fh, fm), "traffic": traffic,
                    "traffic_source": ("profiles/traffic.json was collected on other library sources than this build: no value"
                                       if ttag == "stale" else None) if traffic is None else
                                      "stored profile value (profiles/traffic.json, tag %s, same library sources; 2*FETCH_SIZE + WRITE_SIZE per launch, "
                                      "beyond-L2 bytes incl. Infinity-Cache hits), not measured in this run" % ttag,
                    "algorithmic_bytes_per_launch": e["bytes"] / e["launches"], "flops_per_launch": e["flops"] / e["launches"],
                    "avg_kernel_us": us, "launches_per_step": e["launches"], "share_of_kernel_time": None,
                    "hbm_GBps": gbps, "frac_hbm": fh, "mfma_f32_TFLOPs": tfl, "frac_mfma_f32": tfl / MFMA_F32_PEAK_TFLOPS, "frac_matrix": fm,
                    # gathered neighbour rows (4 nnz S d bytes per launch): served by the L2s / vector L1s, not by HBM -- the
                    # table is cache resident -- so this rate may legitimately exceed the HBM peak
                    "gathered_bytes_per_launch": (e["gather"] / e["launches"]) if e["gather"] else None,
                    "gather_GBps": (e["gather"] / e["s"] / 1e9) if e["gather"] else None,
                    # what a gather kernel can be compared with: the BARE feature-sliced gather (k_aggregate_sliced: the same
                    # neighbour lists, the same 128-byte lines through the vector L1s, nothing else) timed in THIS run on THIS
                    # box over the same chromosomes; `frac` above prices the compulsory HBM bytes only.  (Rounds 2-5 quoted the
                    # guide's 16.8-18.8 TB/s as a ceiling; the bare gather beats it by 11 % on some boxes -- VERDICT r5 #7.)
                    "gather_reference_GBps": gather_ref if e["gather"] else None,
                    "gather_reference": "k_aggregate_sliced on the same graphs, isolated launches, this run" if (e["gather"] and gather_ref) else None,
                    "frac_of_gather_reference": (e["gather"] / e["s"] / 1e9 / gather_ref) if (e["gather"] and gather_ref) else None}
        tot_s = sum(e["s"] for e in agg.values())
        ranked = sorted(((k, e) for k, e in agg.items() if e["bytes"]), key=lambda kv: -kv[1]["s"])
        top3 = []
        for k, e in ranked[:3]:
            r = roof_entry(k, e)
            r["share_of_kernel_time"] = e["s"] / tot_s
            top3.append(r)
        roof = None
        if top3:
            roof = dict(top3[0])
            roof["kernel"] = ("%s: the largest share (%.0f %%) of the summed kernel time of one %s; HIP events around %d isolated launches "
                              "per chromosome on the library's stream, average over its %d launches per step; bytes / flops: DESIGN.md "
                              "section 4 (algorithmic, per launch)" %
                              (top3[0]["kernel"], 100 * top3[0]["share_of_kernel_time"], "train epoch" if genome else "train step", reps,
                               top3[0]["launches_per_step"]))
            roof["sum_kernel_ms_per_step"] = tot_s * 1e3
            roof["all_kernels_us"] = {k: round(e["s"] / max(e["launches"], 1) * 1e6, 2) for k, e in agg.items()}
            # (the row-local backward in its two forms: "(head)" = last layer, the head's backward recomputed in its prologue)
            roof["all_kernels_us"].update({k: round(f[0] / f[1] * 1e6, 2) for k, f in forms.items() if "(" in k})
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the host baseline is reported at N=1 only
            if genome:  # bounded sample of the same genome: its smallest, a middle and its largest train chromosome
                by_n = sorted(shapes, key=lambda s: s[1])
                pick = [by_n[0], by_n[len(by_n) // 2], by_n[-1]]
                sample = [(nm, n, synth.PAIRS_PER_CHROM, synth.chrom_seed(nm)) for nm, n, _ in pick]
                full = [(nm, n, synth.PAIRS_PER_CHROM, synth.chrom_seed(nm)) for nm, n, _ in shapes]   # what `value` is quoted on
            else:
                sample = [(names[0], shapes[0][1], single_shape(args.workload)[2],
                           (synth.chrom_seed(single_shape(args.workload)[0]) if args.workload != "config1" else 0))]
                full = None
            cpu = cpu_baseline(args, sample, args.cpu_seconds, full)
        per_ms = np.array(per) * 1e3
        if genome:
            wl = ("synthetic GM12878-shaped genome (SURVEY 8d config 3): %d train chromosomes, %d windows, 250000 contact "
                  "pairs each (nnz(A+I) %d..%d), d=%d, L=%d, C=%d, dropout=%.2f, SGD lr .25 m .9 wd 1e-6; step = one train "
                  "epoch in reference semantics: per chromosome f+r fwd, BCE, bwd incl. d/dx, optimizer step; all four "
                  "aggregations every step%s%s" %
                  (len(names), windows, min(s[2] for s in shapes), max(s[2] for s in shapes), args.d, args.layers,
                   synth.N_LABELS, args.dropout, "" if args.adj_type == "hic" else "; adj_type=%s" % args.adj_type,
                   "; chromosomes sharded over %d ranks (LPT), one flat-gradient all-reduce per step group (%s)" % (world, "RCCL" if args.backend == "nccl" else args.backend) if multi else ""))
        else:
            wl = ("%s-like synthetic Hi-C chromosome per rank: n=%d windows, %d contact pairs (nnz(A+I)=%d), d=%d, L=%d, "
                  "C=%d, dropout=%.2f, SGD lr .25 m .9 wd 1e-6; train step = f+r fwd, BCE, bwd incl. d/dx, optimizer step%s"
                  % (args.workload, shapes[0][1], single_shape(args.workload)[2], shapes[0][2], args.d, args.layers,
                     synth.N_LABELS, args.dropout, "; grad all-reduce over RCCL" if multi else ""))
        out = {
            "metric": "GCN windows/sec (2-layer, d_model=128) on GM12878 Hi-C graph" if genome else
                      "GCN windows/sec (2-layer, d_model=128) train step, one chromosome",
            "value": windows * steps / elapsed, "unit": "windows/s",
            "n_gpus": world, "ranks_seen_by_backend": seen, "backend": args.backend if multi else None,
            "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if genome else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl, "generator": args.generator,
                       "products": ("split: every dense fp32 product of the d = 128 row-local kernels and the head (U = H W, dW = H^T dU, dHs = dU W^T; pred, dym, dW_out) as six "
                                    "bf16 MFMA partial products of an EXACT three-way split of both fp32 operands (x = h + m + l, 8 + 8 + 8 "
                                    "significant bits), fp32 accumulators -- fp32 arithmetic on the bf16 matrix cores; against float64 its error is "
                                    "below the fp32 MFMA chain's (U, dHs: 2-3x) or on a par with it (dW) (tests/test_gpu_products.py, "
                                    "profiles/r06_bf16x6_probe.txt); "
                                    "the chain's figure: fp32_chain_ms_per_step" if _lib.load().cgcn_debug_get_products() == 1 else
                                    "fp32 MFMA chain (v_mfma_f32_16x16x4_f32), CGCN_PRODUCTS=fp32"),
                       "hip_graph": not args.no_hip_graph, "epoch_graph": bool(stage.epoch_graph and not args.no_hip_graph and not multi),
                       "parallelism": ("chromosomes sharded over %d rank(s)" % world) if genome else "chromosome-per-rank x%d" % world,
                       "allreduce": stage.allreduce_kind if multi else None,
                       "step_group_graph": bool(stage._group_graph_enabled()) if multi else None,
                       "prediction_gather": ((stage.prediction_gather_effective + (" (rows of every chromosome sent to rank 0 by its owner, asynchronously)" if stage.prediction_gather_effective == "rank0" else ""))
                                             + ("" if stage.prediction_gather_effective == args.gather else " [asked for %s: the backend has no device-tensor send / recv]" % args.gather)) if multi and genome else None,
                       "eager_collectives_on_own_communicator": (stage.aux_group is not stage.group) if multi else None},
            "step_ms": {"median": float(np.median(per_ms)), "p10": float(np.percentile(per_ms, 10)),
                        "p90": float(np.percentile(per_ms, 90)), "n": len(per),
                        "note": "per-step host time on rank 0" + (" (each epoch ends with its own loss sync)" if genome else " (launch only: steps are asynchronous)")},
            "value_at_median": windows / (float(np.median(per_ms)) * 1e-3) if genome else None,
            "roofline": roof, "roofline_top3": top3 if roof else None, "cpu_baseline": cpu, "final_loss": final_loss,
        }
        if args.probe:
            out["probe"] = True
        if ladder_rec is not None:
            out["launch_ladder"] = ladder_rec
        out.update(extras)
        print(json.dumps(out))
        sys.stdout.flush()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
