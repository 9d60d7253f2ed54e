"""GCN-stage loop: the build's counterpart of the reference's finetune.py / runner.py.

`finetune(...)` keeps the reference signature and return value (finetune.py:9, :67) so runner.py-style
callers work unchanged.  Underneath, `GCNStage` does what the MI355X wants instead of what the
reference does per chromosome per epoch:

  reference (finetune.py:20-53)                      here
  -------------------------------------------------  ------------------------------------------------
  pickle.load of all graphs every call (:21-23)      graphs normalised + uploaded once, cached on device
  process_graph on the CPU every chromosome (:36)    (ChromGraph), features/targets cached on device
  H2D of features / COO adjacency (:30-36)
  two forward calls, forward + reverse strand        one strand-batched pass ([2,n,d]) over the graph
  (:41-42)
  ~60 small launches + loss.item() sync (:51)        whole step captured once per chromosome into a HIP
                                                     graph and replayed; losses stay on device until the
                                                     split ends
  single GPU                                         one process per GPU, chromosomes sharded across ranks,
                                                     one all-reduce of the flat gradient buffer per step
                                                     group (RCCL over xGMI)
Semantics at world_size 1 are exactly the reference's: one SGD step per chromosome, in dict order."""
from __future__ import annotations

import gc
import os
import pickle
import time
import weakref
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import _lib as _lib_mod
from . import graph as G
from .dist import ShardPlan, plan_shards


class _Chrom:
    __slots__ = ("name", "n", "graph", "x", "target", "cost", "h1", "src_key", "stat_acc")

    def __init__(self, name, n, graph, x, target, cost, src_key=None, stat_acc=False):
        self.name, self.n, self.graph, self.x, self.target, self.cost = name, n, graph, x, target, cost
        self.stat_acc = bool(stat_acc)   # this chromosome's head takes its BatchNorm sums as fixed-point totals (GCNStage._stat_acc_for)
        self.h1 = {"h": None}  # cached A X of the first layer (features and graph are fixed per chromosome)
        self.src_key = src_key  # identity + version of the caller's tensors / graph object this was built from


class _SourceKey:
    """What a cached chromosome was built from: the caller's three feature tensors and Hi-C matrix THEMSELVES (weak
    references) plus the tensors' in-place versions.  An address alone is not an identity: a caller that frees its
    feature dict and regenerates it (handoff.FeatureCollector.finish() every epoch, `del feats; feats = ...`) routinely
    gets the same storage address, version 0 and the same shape back from the caching allocator.  A hit needs the very
    same live objects at the version they were uploaded at; anything else rebuilds (as graph_from_torch_sparse does)."""
    __slots__ = ("refs", "versions", "hic_ref", "hic_sig")

    def __init__(self, feats, hic):
        self.refs, self.versions = [], []
        for k in ("forward", "backward", "target"):
            t = feats[k]
            try:
                self.refs.append(weakref.ref(t))
            except TypeError:          # not weak-referenceable (a list, a numpy scalar ...): never a hit
                self.refs.append(None)
            self.versions.append(t._version if torch.is_tensor(t) else None)
        self.hic_ref = None
        if hic is not None:
            try:
                self.hic_ref = weakref.ref(hic)
            except TypeError:
                self.hic_ref = lambda: None
        self.hic_sig = None if hic is None else (getattr(hic, "nnz", None), getattr(hic, "shape", None))

    def matches(self, feats, hic) -> bool:
        for r, v, k in zip(self.refs, self.versions, ("forward", "backward", "target")):
            t = feats[k]
            if r is None or r() is not t or (torch.is_tensor(t) and t._version != v):
                return False
        if hic is None:
            return self.hic_ref is None
        if self.hic_ref is None or self.hic_ref() is not hic:
            return False
        return self.hic_sig == (getattr(hic, "nnz", None), getattr(hic, "shape", None))


def _graph_uses(key, name) -> bool:
    """does the captured graph stored under `key` touch chromosome `name`?  (key[0]: a name, None, or -- the epoch graph -- a tuple of names)"""
    return key[0] == name or (isinstance(key[0], tuple) and name in key[0])


class _HostArena:
    """Pinned host mirror of the stage's [sum n, C] prediction arena, for callers that want the reference's return value
    (CPU predictions, finetune.py:52-53,67).  The reference copies every chromosome's predictions to the host
    synchronously inside its loop; here the rows of a group of chromosomes (GCNStage._copy_groups; one chromosome without
    epoch graphs) travel on a COPY stream as soon as the group's steps are enqueued -- an event after them orders the copy
    behind them -- so the PCIe transfer (100 MB per train epoch of the GM12878-shaped genome, ~2 ms at 50 GB/s) runs under
    the following kernels and only the last chromosome's rows are exposed.  Two pinned buffers alternate: the tensor a split returns stays valid until the call AFTER the next one
    on the same stage (the reference returns fresh tensors; a caller that keeps predictions longer must clone them)."""

    def __init__(self, rows: int, C: int, device):
        self.device = device
        self.bufs = []
        for _ in range(2):
            try:
                self.bufs.append(torch.empty((rows, C), dtype=torch.float32, pin_memory=True))
            except RuntimeError:   # no pinned memory to be had: pageable (the copies then serialise with the host)
                self.bufs.append(torch.empty((rows, C), dtype=torch.float32))
        self.turn = 0
        self.stream = torch.cuda.Stream(device=device)
        self.events: List[torch.cuda.Event] = []
        self.used = 0

    def begin(self):
        self.turn ^= 1
        self.used = 0

    def fetch(self, rows_list, probs_dev):
        """enqueue the copies of the arena row ranges [r0, r1) behind everything the current stream holds so far
        (adjacent ranges travel as one copy)"""
        if self.used == len(self.events):
            self.events.append(torch.cuda.Event())
        ev = self.events[self.used]
        self.used += 1
        ev.record(torch.cuda.current_stream(self.device))
        self.stream.wait_event(ev)
        runs = []
        for r0, r1 in rows_list:
            if runs and runs[-1][1] == r0:
                runs[-1][1] = r1
            else:
                runs.append([r0, r1])
        with torch.cuda.stream(self.stream):
            for r0, r1 in runs:
                self.bufs[self.turn][r0:r1].copy_(probs_dev[r0:r1], non_blocking=True)

    def finish(self, span, rows_list):
        """wait for the copies; the split's rows as ONE host tensor (a view when the split is a contiguous run of the arena)"""
        self.stream.synchronize()
        buf = self.bufs[self.turn]
        if span is not None:
            return buf[span[0]:span[1]]
        return torch.cat([buf[r0:r1] for r0, r1 in rows_list], 0)


class GCNStage:
    """Device-resident state + step engine of the GCN stage for one split-independent model.

    model      : ChromeGCN-like module with forward_strands(x_fr [2,n,d], graph) -> (logits [2,n,C], gates)
    optimizer  : torch optimizer over model.parameters() (utils/util_methods.py:14-19 builds SGD/Adam)
    hip_graphs : capture each chromosome's step into a HIP graph (needs a GPU)
    input_grad : also produce d loss / d features.  finetune.py:33-34 sets requires_grad on the features, but the
                 resulting x.grad is unobservable there (the tensors are loop-local and finetune returns only
                 predictions, targets and the loss), so the default skips that last gather; train_step returns
                 dx = None then
    stat_acc   : the head's BatchNorm sums as fixed-point integer totals (accumulate mode, include/chromegcn.h:
                 cgcn_layer_fwd_colstats_plan; two launches fewer per step).  None = decided PER CHROMOSOME from a bound on
                 its own features (_stat_acc_for) -- a function of the chromosome's data alone, so every rank of a job and
                 every stage of a process takes the same decision for it; CGCN_STAT_ACC=0 in the environment = never;
                 False = never (per-workgroup records); True = always (NaN loss when a chromosome is out of range)
    group      : torch.distributed process group (None = single process)
    force_collectives : take the multi-rank path (shard plan, all-reduce, prediction gathers) even when the group has
                 one rank -- the only way to drive the engine's RCCL calls on a single-GPU box"""

    def __init__(self, model, optimizer=None, adj_type: str = "hic", device="cuda", hip_graphs: bool = True,
                 input_grad: bool = False, group=None, fused_head: bool = True, cache_input_aggregation: bool = True,
                 force_collectives: bool = False, group_graph: Optional[bool] = None, p2p_allreduce: Optional[bool] = None,
                 prediction_gather: str = "all", aux_group=None, epoch_graph: Optional[bool] = None,
                 stat_acc: Optional[bool] = None):
        self.model = model
        self.stat_acc = stat_acc if stat_acc is not None else (None if os.environ.get("CGCN_STAT_ACC", "1") != "0" else False)
        # one HIP graph for a whole single-rank split (every chromosome's step, back to back) instead of one graph launch
        # per chromosome, when the predictions stay on the device; CGCN_EPOCH_GRAPH=0 / epoch_graph=False = one graph per
        # chromosome (profiles/r04_epoch_graph_experiment.txt)
        self.epoch_graph = epoch_graph if epoch_graph is not None else os.environ.get("CGCN_EPOCH_GRAPH", "1") != "0"
        # where a split's predictions are assembled in a multi-rank run: "all" = on every rank (every rank's run_split
        # returns the whole split, like a single process), "rank0" = on rank 0 only, over direct point-to-point sends
        # (what nn.DataParallel does with the replicas' outputs, main.py:92-94: gathered on device 0; the other ranks'
        # run_split returns preds = None), "none" = nowhere (each rank keeps its chromosomes' rows in its arena)
        if prediction_gather not in ("all", "rank0", "none"):
            raise ValueError("prediction_gather must be 'all', 'rank0' or 'none'")
        self.prediction_gather = prediction_gather
        self.prediction_gather_effective = prediction_gather   # what the last multi-rank split actually did
        # multi-rank step group as ONE HIP graph (fwd + bwd + gradient all-reduce + fused 1/k SGD step): needs a backend
        # whose collectives are stream-ordered device work (nccl = RCCL); None = on when possible, CGCN_GROUP_GRAPH=0 disables
        self._group_graph_opt = group_graph if group_graph is not None else os.environ.get("CGCN_GROUP_GRAPH", "1") != "0"
        self._group_graph_ok = True
        self._comm_warm = False
        # one-shot peer-to-peer all-reduce over xGMI (every rank reads its peers' gradient arenas through symmetric
        # memory and sums them in rank order) instead of RCCL's all-reduce: opt-in (CGCN_P2P_ALLREDUCE=1), SURVEY section 5
        self._p2p_opt = p2p_allreduce if p2p_allreduce is not None else os.environ.get("CGCN_P2P_ALLREDUCE", "0") == "1"
        self._p2p = None
        self.allreduce_kind = "none"
        self.fused_head = fused_head
        # A X of the first layer is loop invariant across steps and epochs (like the normalised CSR): compute it once
        # per chromosome and stream it afterwards.  False = redo that gather every step, as the reference does.
        self.cache_input_aggregation = cache_input_aggregation
        self.optimizer = optimizer
        self.adj_type = adj_type
        self.device = torch.device(device)
        self.hip_graphs = bool(hip_graphs) and self.device.type == "cuda"
        self.input_grad = input_grad
        self.group = group
        if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(group)
            self.rank = torch.distributed.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        if force_collectives and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            raise RuntimeError("force_collectives needs an initialised torch.distributed process group")
        self.multi = self.world > 1 or bool(force_collectives)
        # Everything the split loop issues EAGERLY (asynchronous prediction gathers, the statistics / loss all-reduce)
        # goes over a communicator of its own: the gradient all-reduce is replayed from inside captured HIP graphs, and
        # one communicator must not carry a captured and an eager operation that may execute at the same time.
        # aux_group: a second process group over the same ranks (bench.py creates it); None = created here when the
        # stage spans the default group (new_group is collective over the default group: every rank builds its stage).
        self.aux_group = aux_group if aux_group is not None else group
        if (self.multi and aux_group is None and group in (None, getattr(torch.distributed.group, "WORLD", None))
                and torch.distributed.is_available() and torch.distributed.is_initialized()
                and os.environ.get("CGCN_AUX_GROUP", "1") != "0"):
            self.aux_group = torch.distributed.new_group(backend=torch.distributed.get_backend(group))
        self.chroms: Dict[str, _Chrom] = {}
        self._pending: Dict[str, tuple] = {}   # chromosomes registered with defer=True: (feats, hic) on the host
        self._meta: Dict[str, tuple] = {}      # name -> (n, C, cost) of every registered chromosome, resident or not
        self._graphs: Dict[tuple, dict] = {}
        self._pool = None
        self._flat_grad: Optional[torch.Tensor] = None
        self._flat_param: Optional[torch.Tensor] = None
        self._flat_mom: Optional[torch.Tensor] = None
        self._fused_sgd = False
        self._captured_lr = None
        self._one: Optional[torch.Tensor] = None
        self._targets_cpu: Dict[tuple, torch.Tensor] = {}
        self._gather_plans: Dict[tuple, dict] = {}
        self._targets_dev: Dict[tuple, torch.Tensor] = {}
        self._arena: Optional[dict] = None

    def _drop_graphs(self):
        """forget every captured HIP graph (and the memory pool they shared, which dies with the last of them)"""
        self._graphs.clear()
        self._pool = None

    # ------------------------------------------------------------------ data
    def add_chromosome(self, name: str, feats: Dict[str, torch.Tensor], hic=None, defer: bool = False):
        """feats: {'forward': [n,d], 'backward': [n,d], 'target': [n,C]} (utils/util_methods.py:183-199);
        hic: scipy matrix for 'hic'/'both' graphs (data/7create_graph_new.py:118).
        defer=True (multi-rank runs): only register the chromosome -- size, label count and a cost estimate that is the
        same on every rank -- and keep the host tensors; the graph is normalised and everything uploaded when THIS rank
        first runs the chromosome (run_split's shard plan), so a rank holds the chromosomes it owns, not the genome."""
        n = feats["forward"].shape[0]
        # shard-plan cost: ONE formula from the caller's inputs alone (never from the normalised graph, whose nnz a
        # rank only knows once it has built it): plan_shards must see identical costs on every rank whether a rank
        # registered the chromosome deferred, uploaded it at once, or held it from an earlier split
        cost = self._cost_estimate(hic, n, feats["forward"].shape[1])
        if defer:
            self._pending[name] = (feats, hic)
            self._meta[name] = (n, feats["target"].shape[1], cost)
            self.chroms.pop(name, None)
            self._invalidate_layout(name)
            return
        pend = self._pending.pop(name, None)
        materialising = pend is not None and pend[0] is feats and pend[1] is hic   # same data as registered
        h = G.normalize_graph(self.adj_type, hic, n)
        g = G.upload(h, self.device)
        x = torch.stack([feats["forward"], feats["backward"]]).to(self.device, torch.float32).contiguous()
        t = feats["target"].to(self.device, torch.float32).contiguous()
        known = self._meta.get(name)
        self.chroms[name] = _Chrom(name, n, g, x, t, cost, _SourceKey(feats, hic), self._stat_acc_for(name, n, x))
        if known is None or known[:2] != (n, t.shape[1]):
            self._meta[name] = (n, t.shape[1], cost)
            self._invalidate_layout(name)
        else:
            # same (n, C): the arena / gather layout already accounts for the chromosome; its captured graphs read the
            # old device tensors and go.  The cached target concatenations are COPIES of target data: they survive only
            # when a deferred chromosome materialises from the very tensors it was registered with, and are rebuilt
            # when the caller handed in new (or edited) targets of the same shape.
            if known[2] != cost:          # another graph: the shard plan (rounds, send / recv layout) may change with it
                self._gather_plans.clear()
            self._meta[name] = (n, t.shape[1], cost)
            self._graphs = {k: v for k, v in self._graphs.items() if not _graph_uses(k, name)}
            if not materialising:
                self._targets_cpu.clear()
                self._targets_dev.clear()

    def _stat_acc_for(self, name: str, n: int, x: torch.Tensor) -> bool:
        """Whether this chromosome's steps take the head's BatchNorm batch sums as 64-bit fixed point with 32 fraction bits
        (accumulate mode): sum relu(Xn)^2 must stay below 2^31 per column.  A gated layer keeps |Xn| <= max(1, max |X|) (tanh
        and a convex mix; inter-layer dropout scales by 1 / (1 - p)), so the chromosome's own features bound the sums; outside
        the range (|x| in the hundreds) THIS chromosome uses per-workgroup records -- nothing process-wide changes, the
        other chromosomes of the stage, other stages and other ranks are not affected (round 5 flipped a library switch)."""
        if self.stat_acc is not None:
            return bool(self.stat_acc)
        if n < 2 or x.numel() == 0 or x.device.type != "cuda":
            return False
        keep = 1.0 / max(1e-6, 1.0 - float(getattr(self.model, "dropout", 0.0) or 0.0))
        bound = max(1.0, float(x.abs().max())) * keep ** max(int(getattr(self.model, "n_layers", 1)) - 1, 0)
        ok = n * bound * bound < 2.0 ** 29    # a factor 4 below the totals' range (the per-workgroup partials: 2^22 each)
        if not ok:
            import warnings
            warnings.warn("chromegcn_amd: feature magnitudes up to %.3g on the %d windows of %s are outside the range of the "
                          "fixed-point BatchNorm sums; this chromosome uses per-workgroup records" % (bound, n, name))
        return ok

    def _cost_estimate(self, hic, n: int, d: int) -> float:
        """LPT cost of a chromosome (dist.plan_shards): gather work ~ nnz(A + I) d, dense work ~ 3 n d^2 / 16, with
        nnz estimated from the raw inputs (duplicates / band overlaps not removed)."""
        nnz = n
        if self.adj_type in ("hic", "both") and hic is not None and hasattr(hic, "nnz"):
            nnz += int(hic.nnz)
        if self.adj_type in ("constant", "both"):
            nnz += 14 * n
        return float(nnz) * d + 3.0 * n * d * d / 16.0

    def _invalidate_layout(self, name):
        self._targets_cpu.clear()
        self._targets_dev.clear()
        self._gather_plans.clear()
        self._arena = None   # the output arena is laid out over the chromosome set: rebuilt (and graphs dropped) lazily
        self._graphs = {k: v for k, v in self._graphs.items() if not _graph_uses(k, name)}
        if not self._graphs:
            self._pool = None

    def _resident(self, name: str) -> _Chrom:
        """the device-resident chromosome, uploading a deferred one on first use"""
        c = self.chroms.get(name)
        if c is None:
            feats, hic = self._pending[name]
            self.add_chromosome(name, feats, hic)
            c = self.chroms[name]
        return c

    def load(self, chrom_feature_dict, split_adj_dict=None, only: Optional[Iterable[str]] = None, defer: bool = False):
        """Upload the chromosomes that are not cached yet.  The reference re-reads everything on every call
        (finetune.py:20-36); here a chromosome is reused only while the caller's feature tensors (the same live objects,
        at the same in-place version) and graph object are the ones it was built from -- anything else rebuilds it, dropping its cached
        first-layer aggregation and captured HIP graphs (add_chromosome).  `invalidate()` forces a rebuild."""
        for name in chrom_feature_dict:
            if only is not None and name not in only:
                continue
            hic = None if split_adj_dict is None else split_adj_dict.get(name)
            cur = self.chroms.get(name)
            if cur is not None and cur.src_key is not None and cur.src_key.matches(chrom_feature_dict[name], hic):
                continue
            pend = self._pending.get(name)
            if defer and pend is not None and pend[0] is chrom_feature_dict[name] and pend[1] is hic:
                continue
            self.add_chromosome(name, chrom_feature_dict[name], hic, defer=defer)

    def invalidate(self, name: Optional[str] = None):
        """forget the device copy of one chromosome (or of all of them) and every HIP graph captured on it"""
        for nm in ([name] if name is not None else list(self._meta)):
            self.chroms.pop(nm, None)
            self._pending.pop(nm, None)
            self._meta.pop(nm, None)
            self._graphs = {k: v for k, v in self._graphs.items() if not _graph_uses(k, nm)}
        if not self._graphs:
            self._pool = None
        self._targets_cpu.clear()
        self._targets_dev.clear()
        self._gather_plans.clear()
        self._arena = None

    # ------------------------------------------------------------------ output arena
    def _ensure_arena(self):
        """One [sum n, C] buffer for the predictions and one [#chromosomes] buffer for the losses of every cached
        chromosome, in insertion order (= the reference's chromosome iteration order, finetune.py:29).  The fused head
        writes each chromosome's sigmoid(pred) and mean BCE straight into its slice, so the concatenation of
        finetune.py:52 and the loss sum of :51 need no per-chromosome copy / add kernels."""
        if self._arena is not None or not self._meta:
            return
        C = next(iter(self._meta.values()))[1]
        total = sum(m[0] for m in self._meta.values())
        probs = torch.empty((total, C), device=self.device, dtype=torch.float32)
        loss = torch.zeros(len(self._meta), device=self.device, dtype=torch.float32)
        slots, rows, order = {}, {}, {}
        off = 0
        for i, (nm, (n_, _c, _cost)) in enumerate(self._meta.items()):   # registration order, resident or deferred
            slots[nm] = {"probs": probs[off:off + n_], "loss": loss[i:i + 1]}
            rows[nm] = (off, off + n_)
            order[nm] = i
            off += n_
        self._arena = {"probs": probs, "loss": loss, "slots": slots, "rows": rows, "order": order}
        self._drop_graphs()  # captured graphs wrote into the previous arena

    def _arena_span(self, names):
        """(row0, row1, i0, i1) if `names` is a contiguous run of the arena's chromosome order, else None"""
        if not names:
            return None
        order = self._arena["order"]
        idx = [order[nm] for nm in names]
        if idx != list(range(idx[0], idx[0] + len(idx))):
            return None
        return self._arena["rows"][names[0]][0], self._arena["rows"][names[-1]][1], idx[0], idx[-1] + 1

    # ------------------------------------------------------------------ flat parameter / gradient / momentum arenas
    def _params(self):
        return [p for p in self.model.parameters() if p.requires_grad]

    def _fused_sgd_eligible(self, ps):
        o = self.optimizer
        if o is None or type(o) is not torch.optim.SGD or len(o.param_groups) != 1 or self.device.type != "cuda":
            return False
        g = o.param_groups[0]
        same = len(g["params"]) == len(ps) and all(a is b for a, b in zip(g["params"], ps))
        return bool(same and g.get("dampening", 0) == 0 and not g.get("maximize", False))

    @staticmethod
    def _offsets(ps):
        """element offset of each parameter in the flat arenas; every parameter starts 16-byte aligned
        (the kernels read weights with 16-byte loads), padding elements stay zero forever"""
        offs, off = [], 0
        for p in ps:
            offs.append(off)
            off += (p.numel() + 3) & ~3
        return offs, off

    def _views_ok(self, ps, flat, attr):
        offs, total = self._offsets(ps)
        if flat is None or flat.numel() != total or flat.device != ps[0].device:
            return False
        for p, off in zip(ps, offs):
            t = p.data if attr == "data" else p.grad
            if t is None or t.data_ptr() != flat.data_ptr() + 4 * off or not t.is_contiguous():
                return False
        return True

    def _ensure_flat_grad(self):
        """Make every parameter, its .grad and (plain SGD) its momentum buffer a view into one flat fp32
        buffer each: one all-reduce for the gradients, one launch for the optimizer step, and gradient
        'sinks' the backward kernels write into directly (no per-parameter autograd accumulate kernels)."""
        ps = self._params()
        offs, total = self._offsets(ps)
        dev = ps[0].device
        ok = self._views_ok(ps, self._flat_param, "data") and self._views_ok(ps, self._flat_grad, "grad")
        if ok and self._fused_sgd == self._fused_sgd_eligible(ps):
            return
        with torch.no_grad():
            flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
            for p, off in zip(ps, offs):
                flat_p[off:off + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + p.numel()].view(p.shape)
            self._flat_param = flat_p
            self._flat_grad = self._alloc_flat_grad(total, dev)
            for p, off in zip(ps, offs):
                p.grad = self._flat_grad[off:off + p.numel()].view(p.shape)
            self._fused_sgd = self._fused_sgd_eligible(ps)
            self._flat_mom = None
            if self._fused_sgd and self.optimizer.param_groups[0].get("momentum", 0) != 0:
                self._flat_mom = torch.zeros(total, device=dev, dtype=torch.float32)
                for p, off in zip(ps, offs):
                    st = self.optimizer.state[p]
                    view = self._flat_mom[off:off + p.numel()].view(p.shape)
                    if torch.is_tensor(st.get("momentum_buffer")):
                        view.copy_(st["momentum_buffer"])
                    st["momentum_buffer"] = view  # torch's own step() would keep using (and updating) this view
        managed = dev.type == "cuda" and hasattr(self.model, "_rng_state")
        self.model._rng_managed = managed
        self.model._grad_sink = dev.type == "cuda" and hasattr(self.model, "forward_loss") and self.fused_head
        self._drop_graphs()

    def _optimizer_step(self, grad_scale: float = 1.0):
        """finetune.py:49.  Plain torch SGD runs as one fused launch over the flat buffers (which also
        advances the dropout counter and applies the 1/k of a k-rank step group); anything else goes
        through optimizer.step()."""
        rng = self.model._rng_state if getattr(self.model, "_rng_managed", False) else None
        if self._fused_sgd:
            from . import ops
            g = self.optimizer.param_groups[0]
            ops.sgd_step(self._flat_param, self._flat_grad, self._flat_mom, g["lr"], g.get("momentum", 0),
                         g.get("weight_decay", 0), g.get("nesterov", False), rng, grad_scale)
        else:
            if grad_scale != 1.0:
                self._flat_grad.mul_(grad_scale)
            self.optimizer.step()
            if rng is not None:
                rng[1] += 1

    # ------------------------------------------------------------------ one chromosome, eager
    def _forward_loss(self, c: _Chrom, x):
        slot = self._arena["slots"][c.name] if self._arena is not None else None
        if self.fused_head and hasattr(self.model, "forward_loss"):
            loss, probs, _ = self.model.forward_loss(x, c.graph, c.target,   # fused head + loss kernels
                                                     h1_cache=c.h1 if self.cache_input_aggregation else None,
                                                     out_slots=slot, stat_acc=c.stat_acc)
            return loss, probs
        logits, _ = self.model.forward_strands(x, c.graph)
        pred = (logits[0] + logits[1]) / 2                                # finetune.py:43
        loss = F.binary_cross_entropy_with_logits(pred, c.target)          # finetune.py:45
        if slot is None:
            return loss, torch.sigmoid(pred).detach()                      # finetune.py:52
        with torch.no_grad():
            torch.sigmoid(pred, out=slot["probs"])
            slot["loss"].copy_(loss.detach().view(1))
        return loss, slot["probs"]

    def _fwd_bwd_step(self, c: _Chrom):
        """forward + backward + optimizer step of one chromosome (finetune.py:38-49).  With the fused SGD the step
        rides in the last backward launch (cgcn_sgd_fuse: the first layer's gather kernel -- or, when nobody wants
        d loss / d features, its partial-sum launch -- carries it in extra workgroups) instead of being a launch of
        its own."""
        from . import ops
        fuse = self._fused_sgd and getattr(self.model, "_grad_sink", False) and getattr(self.model, "n_layers", 0) >= 1
        if fuse:
            g = self.optimizer.param_groups[0]
            ops._sgd_fuse = {"param": self._flat_param, "grad": self._flat_grad, "mom": self._flat_mom, "lr": g["lr"],
                             "momentum": g.get("momentum", 0), "weight_decay": g.get("weight_decay", 0),
                             "nesterov": g.get("nesterov", False), "grad_scale": 1.0, "done": False,
                             "rng_state": self.model._rng_state if getattr(self.model, "_rng_managed", False) else None}
        try:
            out = self._fwd_bwd(c)
            done = bool(fuse and ops._sgd_fuse["done"])
        finally:
            ops._sgd_fuse = None
        if not done:
            self._optimizer_step()                                         # finetune.py:49
        return out

    def _fwd_bwd(self, c: _Chrom):
        x = c.x.detach().requires_grad_(True) if self.input_grad else c.x  # finetune.py:33-34
        if not getattr(self.model, "_grad_sink", False):
            self._flat_grad.zero_()                                        # finetune.py:39 (sinks overwrite instead)
        loss, probs = self._forward_loss(c, x)
        if self._one is None or self._one.device != loss.device:
            self._one = torch.ones((), device=loss.device)
        loss.backward(self._one)                                           # finetune.py:48 (root gradient reused: no fill kernel)
        return loss.detach(), probs, (x.grad if self.input_grad else None)

    def _eval(self, c: _Chrom):
        with torch.no_grad():
            loss, probs = self._forward_loss(c, c.x)
        return loss, probs

    # ------------------------------------------------------------------ HIP-graph capture
    def _snapshot(self):
        st = {"model": {k: v.clone() for k, v in self.model.state_dict().items()}}
        rng = getattr(self.model, "_rng_state", None)   # non-persistent buffer (not in the state_dict): the dropout
        if torch.is_tensor(rng):                        # step counter must not remember the warm-up steps either
            st["rng"] = rng.clone()
        if self.optimizer is not None:
            st["opt"] = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in s.items()}
                         for s in (self.optimizer.state.get(p, {}) for p in self._params())]
        return st

    def _restore(self, st):
        with torch.no_grad():
            for k, v in self.model.state_dict().items():
                v.copy_(st["model"][k])
            if "rng" in st:
                self.model._rng_state.copy_(st["rng"])
            if self.optimizer is not None:
                for p, saved in zip(self._params(), st["opt"]):
                    cur = self.optimizer.state.get(p, {})
                    for k, v in cur.items():
                        if torch.is_tensor(v):
                            if k in saved and torch.is_tensor(saved[k]):
                                v.copy_(saved[k])
                            else:
                                v.zero_()  # state created during warm-up (e.g. momentum_buffer): 0 == "not yet stepped"

    def _lr_signature(self):
        if self.optimizer is None:
            return None
        return tuple((g.get("lr"), g.get("momentum"), g.get("weight_decay"), g.get("nesterov")) for g in self.optimizer.param_groups)

    def _capture(self, c: Optional[_Chrom], kind: str, group_size: int = 1):
        """kind: 'train' (zero_grad+fwd+bwd+step), 'fwdbwd' (no optimizer step: multi-rank), 'eval', 'group' (multi-rank:
        fwd+bwd of `c` -- or a zeroed gradient when this rank sits the round out --, all-reduce, fused 1/k step)."""
        was_training = self.model.training
        self.model.train(kind not in ("eval", "epoch_eval"))
        snap = self._snapshot()

        def body(collective=True):
            if kind in ("epoch", "epoch_eval"):   # c: the split's chromosomes; their steps one after the other in ONE graph
                out = None
                for cc in c:
                    out = self._fwd_bwd_step(cc) if kind == "epoch" else self._eval(cc) + (None,)
                return out
            if kind == "eval":
                loss, probs = self._eval(c)
                return loss, probs, None
            if kind == "train":
                return self._fwd_bwd_step(c)
            if kind == "group":
                if c is not None:
                    out = self._fwd_bwd(c)
                else:
                    self._flat_grad.zero_()
                    out = (None, None, None)
                if collective:   # NOT in the warm-up passes: ranks capture at different times (a rank replays a graph it
                    self._allreduce_grads()   # already holds while a peer captures), the collective count must match
                self._optimizer_step(1.0 / group_size if group_size > 1 else 1.0)
                return out
            return self._fwd_bwd(c)

        if kind in ("train", "epoch") and not self._fused_sgd:
            # torch optimizers are not capturable by default (Adam.step raises under capture): callers replay
            # 'fwdbwd' and step eagerly instead (train_step)
            raise RuntimeError("only the fused SGD step can be captured; use kind='fwdbwd' + an eager optimizer.step()")
        try:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(2):  # warm-up: allocator pools, rocBLAS/MIOpen handles, lazy optimizer state
                    body(collective=False)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            self._restore(snap)
            graph = torch.cuda.CUDAGraph()
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            # thread_local: an RCCL watchdog thread polling events while we capture must not invalidate the capture.
            # No cyclic garbage collection while the stream captures: a collection that happens to free an earlier
            # stage's graphs / events / device buffers calls HIP APIs that are illegal during capture, and a failing
            # destructor aborts the process (seen once in ~10 full test runs; torch.cuda.graph collects on entry only).
            gc_was_enabled = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(graph, pool=self._pool, capture_error_mode="thread_local"):
                    loss, probs, dx = body()
            finally:
                if gc_was_enabled:
                    gc.enable()
        finally:
            # the warm-up steps mutated parameters / running statistics / optimizer state: whatever happened above,
            # hand the model back exactly as it came in (capture itself launches nothing)
            self._restore(snap)
            self.model.train(was_training)
        return {"graph": graph, "loss": loss, "probs": probs, "dx": dx}

    def _replay(self, c: Optional[_Chrom], kind: str, group_size: int = 1):
        if self._captured_lr != self._lr_signature():
            self._drop_graphs()  # the learning rate is baked into the captured optimizer kernels
            self._captured_lr = self._lr_signature()
        key = (c.name if c is not None else None, kind) if kind != "group" else (c.name if c is not None else None, kind, group_size)
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = self._capture(c, kind, group_size)
        ent["graph"].replay()
        return ent["loss"], ent["probs"], ent["dx"]

    def _replay_epoch(self, names, train: bool) -> bool:
        """The chromosomes `names` of a split as ONE captured graph (single rank; training needs the fused SGD): one graph
        launch instead of one per chromosome -- the launches' gaps were 3 % of the genome epoch
        (profiles/r04_epoch_graph_experiment.txt).  False = not applicable, the caller steps chromosome by chromosome."""
        if not (self.epoch_graph and self.hip_graphs and len(names) > 1):
            return False
        if train:
            self.model.train()
            self._ensure_flat_grad()
            if not self._fused_sgd:
                return False
        else:
            self.model.eval()
        self._ensure_arena()
        cs = tuple(self._resident(nm) for nm in names)
        if self._captured_lr != self._lr_signature():
            self._drop_graphs()  # the learning rate is baked into the captured optimizer kernels
            self._captured_lr = self._lr_signature()
        kind = "epoch" if train else "epoch_eval"
        key = (tuple(names), kind)
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = self._capture(cs, kind)
        ent["graph"].replay()
        return True

    def _copy_groups(self, names):
        """A to_cpu split as a few graphs instead of one: the host copies of a group's rows are ordered behind the group's
        graph by an event and run under the NEXT group's kernels (inside one graph neither memcpy nodes nor a copy kernel
        on a forked branch overlapped: 6.7 / 6.55 ms per epoch instead of 4.9).  Every group holds at least half of the
        rows still to come, so the copies keep up (PCIe moves a row 2-3x faster than the GPU computes one) and what is
        exposed at the end is the last chromosome's copy alone: 16 chromosomes -> groups of 6, 4, 3, 2, 1."""
        if not (self.epoch_graph and self.hip_graphs):
            return [[nm] for nm in names]
        left = sum(self._meta[nm][0] for nm in names)
        groups, cur, rows = [], [], 0
        for nm in names:
            cur.append(nm)
            rows += self._meta[nm][0]
            if 2 * rows >= left:
                groups.append(cur)
                left -= rows
                cur, rows = [], 0
        if cur:
            groups.append(cur)
        return groups

    # ------------------------------------------------------------------ public steps
    def train_step(self, name: str):
        """One reference train step on one chromosome (finetune.py:38-49).  Returns device tensors
        (loss [], probs [n,C], dx [2,n,d] or None); valid until the next step."""
        c = self._resident(name)
        self.model.train()
        self._ensure_flat_grad()
        self._ensure_arena()
        if self.multi:
            raise RuntimeError("use train_group() when running on more than one rank")
        if self._fused_sgd:                          # zero_grad + fwd + bwd + fused SGD: one HIP graph (or the same, eagerly)
            return self._replay(c, "train") if self.hip_graphs else self._fwd_bwd_step(c)
        # any other optimizer (Adam is the reference's -optim adam, utils/util_methods.py:20-21): its step() is
        # not capturable, so the graph ends after the backward and the step runs eagerly on the flat buffers
        loss, probs, dx = self._replay(c, "fwdbwd") if self.hip_graphs else self._fwd_bwd(c)
        self._optimizer_step()
        return loss, probs, dx

    def eval_step(self, name: str):
        c = self._resident(name)
        self.model.eval()
        self._ensure_arena()
        if self.hip_graphs:
            loss, probs, _ = self._replay(c, "eval")
            return loss, probs
        return self._eval(c)

    def _warm_comms(self):
        """One tiny eager all-reduce per communicator before anything else uses it: the lazy initialisation of a
        communicator must not happen under stream capture, and the first operation on a group must be one EVERY rank takes
        part in (a round's point-to-point sends involve only the ranks that own a chromosome in it).  Every rank reaches
        its first train_group / multi-rank run_split call at the same point of the program."""
        if not self.multi or self._comm_warm:
            return
        torch.distributed.all_reduce(torch.zeros(1, device=self.device), group=self.group)
        if self.aux_group is not self.group:
            torch.distributed.all_reduce(torch.zeros(1, device=self.device), group=self.aux_group)
        self._comm_warm = True

    def train_group(self, name: Optional[str], group_size: int):
        """Multi-rank step group: every rank runs fwd+bwd on its own chromosome (or none), gradients are
        summed across ranks in ONE all-reduce of the flat buffer and divided by the number of chromosomes
        in the group, then every rank takes the same optimizer step.
        With a stream-ordered collective backend (nccl = RCCL) and the fused SGD the WHOLE group step -- forward,
        backward, all-reduce, 1/k scaling + optimizer step -- is one HIP graph per (chromosome, group size): nothing
        is launched from the host between the last backward kernel and the collective, or between the collective and
        the step.  Otherwise: captured fwd+bwd, then the collective and ONE fused scale+step launch, stream-ordered."""
        self.model.train()
        self._ensure_flat_grad()
        self._ensure_arena()
        c = self._resident(name) if name is not None else None
        scale = 1.0 / group_size if group_size > 1 else 1.0
        self._warm_comms()
        if self._group_graph_enabled():
            try:
                return self._replay(c, "group", group_size)
            except Exception as e:   # capture refused by the backend: every rank fails alike, before any collective ran
                import warnings
                warnings.warn("chromegcn_amd: capturing the step group (collective included) failed (%r); "
                              "using the captured fwd+bwd + stream-ordered collective + fused step instead" % (e,))
                self._group_graph_ok = False
                self._drop_graphs()
        out = (None, None, None)
        if c is not None:
            out = self._replay(c, "fwdbwd") if self.hip_graphs else self._fwd_bwd(c)
        else:
            self._flat_grad.zero_()
        if self.multi:
            self._allreduce_grads()
        self._optimizer_step(scale)
        return out

    def _group_graph_enabled(self) -> bool:
        if not (self.multi and self.hip_graphs and self._fused_sgd and self._group_graph_opt and self._group_graph_ok):
            return False
        return torch.distributed.get_backend(self.group) == "nccl"

    def _allreduce_grads(self):
        """SUM of the flat gradient arena over the ranks (the 1/k is folded into the optimizer step)."""
        h = self._p2p
        if h is None:
            torch.distributed.all_reduce(self._flat_grad, op=torch.distributed.ReduceOp.SUM, group=self.group)
            return
        # one-shot peer-to-peer all-reduce: the arena is symmetric memory; after a device-side barrier every rank reads
        # its peers' arenas over its direct xGMI links and adds them IN RANK ORDER (the same order everywhere: all ranks
        # end up with bit-identical sums), a second barrier keeps any rank from overwriting an arena a peer still reads
        h.barrier(channel=0)
        acc = self._p2p_acc
        torch.add(self._p2p_views[0], self._p2p_views[1], out=acc) if len(self._p2p_views) > 1 else acc.copy_(self._p2p_views[0])
        for v in self._p2p_views[2:]:
            acc.add_(v)
        h.barrier(channel=1)
        self._flat_grad.copy_(acc)

    def _alloc_flat_grad(self, total, dev):
        """the flat gradient arena: plain device memory, or (opt-in P2P all-reduce) symmetric memory every peer can read"""
        self._p2p = None
        # what carries the gradient all-reduce: "rccl" (torch backend nccl on ROCm), "gloo", ... or "p2p_one_shot" below
        self.allreduce_kind = ({"nccl": "rccl"}.get(torch.distributed.get_backend(self.group), torch.distributed.get_backend(self.group))
                               if self.multi else "none")
        if self.multi and self._p2p_opt and dev.type == "cuda" and torch.distributed.get_backend(self.group) == "nccl":
            try:   # collective: every rank takes this branch in the same call
                import torch.distributed._symmetric_memory as symm
                gname = (self.group or torch.distributed.group.WORLD).group_name
                buf = symm.empty(total, dtype=torch.float32, device=dev)
                buf.zero_()
                h = symm.rendezvous(buf, gname)
                self._p2p_views = [h.get_buffer(r, (total,), torch.float32) for r in range(h.world_size)]
                self._p2p_acc = torch.zeros(total, device=dev, dtype=torch.float32)
                self._p2p = h
                self.allreduce_kind = "p2p_one_shot"
                return buf
            except Exception as e:
                import warnings
                warnings.warn("chromegcn_amd: symmetric-memory P2P all-reduce unavailable (%r); using RCCL" % (e,))
        return torch.zeros(total, device=dev, dtype=torch.float32)

    def sync_running_stats(self, extra: Optional[torch.Tensor] = None, calls_total: int = 0, calls_mine: int = 0):
        """BatchNorm running statistics see different chromosomes on different ranks; average them so every rank
        evaluates with the same model (deliberate deviation).  ONE all-reduce: the floating-point buffers
        and `extra` (values to be SUMMED over ranks, e.g. the split's loss) travel in the same flat tensor.
        Integer counters (num_batches_tracked) need no collective: every rank knows how many BatchNorm calls the
        whole step plan makes (`calls_total`) and how many of them it made itself (`calls_mine`).
        Returns the summed `extra` (or None)."""
        if not self.multi:
            return extra
        bufs = [b for k, b in self.model.named_buffers() if b.dtype.is_floating_point]
        parts = [b.reshape(-1).float() for b in bufs]
        n_stat = sum(p.numel() for p in parts)
        if extra is not None:
            parts.append(extra.reshape(-1).float() * self.world)   # undo the mean below: extras are sums
        if not parts:
            return extra
        flat = torch.cat(parts)
        torch.distributed.all_reduce(flat, group=self.aux_group)
        flat.div_(self.world)
        off = 0
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
        if calls_total:
            for k, b in self.model.named_buffers():
                if k.endswith("num_batches_tracked"):    # (not the dropout RNG state, which is an integer buffer too)
                    b += int(calls_total - calls_mine)   # the calls the other ranks made
        return flat[n_stat:].view_as(extra) if extra is not None else None

    # ------------------------------------------------------------------ a whole split
    def run_split(self, split: str, names: Optional[Sequence[str]] = None, to_cpu: bool = True, sync_loss: bool = True):
        """(all_preds, all_targets, total_loss) with finetune.py:67's meaning: sigmoid probabilities and
        targets concatenated in chromosome order, total_loss = sum of per-chromosome mean BCE.
        to_cpu=True returns CPU tensors like the reference (finetune.py:52-53 moves every chromosome's
        predictions to the host); to_cpu=False leaves them on the device for chromegcn_amd.metrics -- the
        predictions are then a view of the stage's output arena, valid until the next step on this stage.
        Multi-rank: every rank returns the full concatenation (prediction_gather="all").
        sync_loss=False (with to_cpu=False): total_loss comes back as a 0-dim DEVICE tensor and the split does not wait for
        the GPU at all -- a training loop that only logs the loss can read it an epoch late and keep the queue full
        (finetune.py:51 waits for the device once per chromosome, sync_loss=True once per split)."""
        names = list(self._meta) if names is None else list(names)
        train = split == "train"
        C = next(iter(self._meta.values()))[1] if self._meta else 0
        if not self.multi:
            self._ensure_arena()
            # to_cpu (the reference's return value, finetune.py:52-53): the rows of every group of chromosomes leave for a
            # pinned host arena on a copy stream right behind the group's graph, while the next group computes (_HostArena)
            host = self._host_arena(C) if (to_cpu and names and self.device.type == "cuda") else None
            for grp in ([names] if host is None else self._copy_groups(names)):
                if not self._replay_epoch(grp, train):
                    for nm in grp:
                        self.train_step(nm) if train else self.eval_step(nm)   # results land in the arena: nothing to copy or add
                if host is not None:
                    host.fetch([self._arena["rows"][nm] for nm in grp], self._arena["probs"])
            span = self._arena_span(names)
            if span is not None:   # the usual case: the split is the stage's chromosomes in order -> views, no copy
                preds_dev = self._arena["probs"][span[0]:span[1]]
                loss_dev = self._arena["loss"][span[2]:span[3]].sum()
            elif names:
                preds_dev = torch.cat([self._arena["slots"][nm]["probs"] for nm in names], 0)
                loss_dev = torch.cat([self._arena["slots"][nm]["loss"] for nm in names]).sum()
            else:
                preds_dev = torch.empty((0, C), device=self.device)
                loss_dev = torch.zeros((), device=self.device)
            total = float(loss_dev.item()) if (sync_loss or to_cpu) else loss_dev   # the one host sync of the split (finetune.py:51 syncs per chromosome)
            if not to_cpu:
                return preds_dev, self._split_targets_dev(names, C), total
            if host is not None:
                preds = host.finish(span, [self._arena["rows"][nm] for nm in names])
            else:
                preds = preds_dev.cpu()
        else:
            self._warm_comms()
            plan = plan_shards({nm: self._meta[nm][2] for nm in names}, self.world)
            ag = self.aux_group
            nccl = torch.distributed.get_backend(ag) == "nccl"
            mode = self.prediction_gather
            if mode == "rank0" and not nccl and self.device.type == "cuda":
                mode = "all"   # gloo has no device-tensor send / recv (functional runs on a shared GPU): all-gather there
            self.prediction_gather_effective = mode
            gp = self._gather_plan(names, plan, C, mode)
            peer = (lambda r: r) if ag is None else (lambda r: torch.distributed.get_global_rank(ag, r))
            loss_sum = torch.zeros((), device=self.device)
            pending = []
            for r, group in enumerate(plan.rounds):
                nm = group[self.rank] if self.rank < len(group) else None
                k = sum(1 for g in group if g is not None)
                if train:
                    loss, p, _ = self.train_group(nm, k)
                elif nm is not None:
                    loss, p = self.eval_step(nm)
                else:
                    loss = p = None
                if nm is not None:
                    loss_sum += loss
                if mode == "all":
                    send, recv = gp["send"][r], gp["recv"][r]
                    if nm is not None:
                        send[:p.shape[0]].copy_(p)            # this rank's slab of the round's gather
                    # predictions of this round: ONE all-gather, issued asynchronously so that it travels (xGMI) while
                    # the next round computes; the split waits for all of them once, at the end
                    if nccl:
                        pending.append(torch.distributed.all_gather_into_tensor(recv, send, group=ag, async_op=True))
                    else:  # gloo (CPU tests, single-GPU functional runs): the list form is the one every backend implements
                        pending.append(torch.distributed.all_gather(list(recv.view(self.world, -1, C).unbind(0)), send,
                                                                    group=ag, async_op=True))
                elif mode == "rank0":
                    # every owner sends its chromosome's rows straight to rank 0 (one xGMI link each, unpadded); rank 0
                    # posts the matching receives into its slice of the split buffer; asynchronous like the all-gather
                    ops = []
                    if self.rank == 0:
                        for src, g in enumerate(group):
                            if g is None:
                                continue
                            dst_rows = gp["rows"][g]
                            if src == 0:
                                dst_rows.copy_(p)
                            else:
                                ops.append(torch.distributed.P2POp(torch.distributed.irecv, dst_rows, peer(src), group=ag))
                    elif nm is not None:
                        ops.append(torch.distributed.P2POp(torch.distributed.isend, p, peer(0), group=ag))
                    if ops:
                        pending.extend(torch.distributed.batch_isend_irecv(ops))
            mine = sum(1 for nm in names if plan.owner[nm] == self.rank)
            S = 2   # strands per chromosome: forward + reverse complement (add_chromosome stacks both)
            if train:   # running statistics + the loss in one all-reduce; counters without communication
                loss_sum = self.sync_running_stats(loss_sum, calls_total=S * len(names), calls_mine=S * mine)
            else:
                torch.distributed.all_reduce(loss_sum, group=ag)
            for w in pending:
                w.wait()
            total = float(loss_sum.item())
            if mode == "all":
                # one index_select puts the gathered rows into the reference's chromosome order (finetune.py:52)
                preds_dev = gp["recv_all"].index_select(0, gp["index"]) if gp["total"] else gp["recv_all"][:0]
            elif mode == "rank0":
                preds_dev = gp["recv_all"] if self.rank == 0 else None   # received in the reference's order already
            else:
                preds_dev = None
            if preds_dev is None:
                return None, (self._split_targets_dev(names, C) if not to_cpu else None), total
            if not to_cpu:
                return preds_dev, self._split_targets_dev(names, C), total
            preds = preds_dev.cpu()
        key = tuple(names)
        if key not in self._targets_cpu:  # targets never change: one D2H per split, not one per epoch
            self._targets_cpu[key] = torch.cat([self._target_of(nm).cpu() for nm in names], 0) if names else torch.empty(0, C)
        return preds, self._targets_cpu[key], total

    def _host_arena(self, C):
        """the pinned mirror of the output arena (built with it, dropped with it)"""
        ha = self._arena.get("host")
        if ha is None:
            ha = self._arena["host"] = _HostArena(self._arena["probs"].shape[0], C, self.device)
        ha.begin()
        return ha

    def _split_targets_dev(self, names, C):
        key = tuple(names)
        if key not in self._targets_dev:  # targets never change: concatenated once per split, not once per epoch
            self._targets_dev[key] = (torch.cat([self._target_of(nm).to(self.device) for nm in names], 0) if names
                                      else torch.empty(0, C, device=self.device))
        return self._targets_dev[key]

    def _target_of(self, nm):
        """targets [n, C] of a registered chromosome: the device copy when resident, the caller's host tensor otherwise"""
        c = self.chroms.get(nm)
        return c.target if c is not None else self._pending[nm][0]["target"].to(torch.float32)

    def _gather_plan(self, names, plan: ShardPlan, C: int, mode: str = "all"):
        """Buffers and the row permutation of the prediction gathers, built once per (split, plan).  Round r of the
        plan gathers [world, max_r, C] rows (max_r = the round's largest chromosome; every rank sends one padded slab);
        the rounds' receive buffers are consecutive slices of ONE allocation, and `index` maps the reference's
        concatenation order (finetune.py:52, chromosome iteration order) onto its rows."""
        key = (tuple(names), self.world, C, mode)
        gp = self._gather_plans.get(key)
        if gp is not None:
            return gp
        sizes = {nm: self._meta[nm][0] for nm in names}
        if mode != "all":   # "rank0": one [sum n, C] buffer on rank 0 in the reference's order, a view per chromosome
            gp = {"recv_all": None, "rows": {}, "total": sum(sizes.values())}
            if mode == "rank0" and self.rank == 0:
                gp["recv_all"] = torch.empty((gp["total"], C), device=self.device, dtype=torch.float32)
                off = 0
                for nm in names:
                    gp["rows"][nm] = gp["recv_all"][off:off + sizes[nm]]
                    off += sizes[nm]
            self._gather_plans[key] = gp
            return gp
        max_r = [max([sizes[g] for g in group if g is not None] + [0]) for group in plan.rounds]
        base_r, tot = [], 0
        for m in max_r:
            base_r.append(tot)
            tot += self.world * m
        recv_all = torch.empty((tot, C), device=self.device, dtype=torch.float32)
        where = {}
        for r, group in enumerate(plan.rounds):
            for rank, g in enumerate(group):
                if g is not None:
                    where[g] = base_r[r] + rank * max_r[r]
        idx = [torch.arange(where[nm], where[nm] + sizes[nm], dtype=torch.int64) for nm in names]
        index = torch.cat(idx) if idx else torch.empty(0, dtype=torch.int64)
        gp = {"send": [torch.zeros((m, C), device=self.device, dtype=torch.float32) for m in max_r],
              "recv": [recv_all[base_r[r]:base_r[r] + self.world * max_r[r]] for r in range(len(max_r))],
              "recv_all": recv_all, "index": index.to(self.device), "total": int(index.numel())}
        self._gather_plans[key] = gp
        return gp


# ---------------------------------------------------------------------------------------------
# reference-signature entry points
# ---------------------------------------------------------------------------------------------
_GRAPH_FILES: Dict[str, dict] = {}


def _load_graph_file(opt, split):
    """finetune.py:20-23 -- but unpickled once per file, not once per call."""
    path = os.path.join(opt.graph_root, split + "_graphs_" + opt.hicsize + "_" + opt.hicnorm + "norm.pkl")
    if path not in _GRAPH_FILES:
        with open(path, "rb") as f:
            _GRAPH_FILES[path] = pickle.load(f)
    return _GRAPH_FILES[path]


def finetune(WindowModel, ChromeModel, chrom_feature_dict, crit, optimizer, epoch, data_dict, opt, split,
             split_adj_dict=None):
    """Same arguments and return value as the reference's finetune (finetune.py:9,67).  WindowModel, crit,
    epoch and data_dict are accepted and unused, as in the reference.  Extra keyword split_adj_dict lets a
    caller hand the {chrom: scipy matrix} dict in directly instead of via opt.graph_root."""
    adj_type = getattr(opt, "adj_type", "hic")
    if split_adj_dict is None and adj_type in ("hic", "both"):
        split_adj_dict = _load_graph_file(opt, split)
    stages = ChromeModel.__dict__.setdefault("_cgcn_stages", {})
    key = (split, id(optimizer), adj_type)
    stage = stages.get(key)
    if stage is None:
        dev = next(ChromeModel.parameters()).device
        stage = stages[key] = GCNStage(ChromeModel, optimizer, adj_type=adj_type, device=dev,
                                       hip_graphs=getattr(opt, "hip_graphs", True))
    stage.load(chrom_feature_dict, split_adj_dict)
    return stage.run_split(split, list(chrom_feature_dict))


def run_epoch(WindowModel, ChromeModel, split_data, crit, optimizer, epoch, data_dict, opt, split, **kw):
    """runner.py:10-23: times the stage; elapsed is in minutes like the reference."""
    start = time.time()
    pred, targ, loss = finetune(WindowModel, ChromeModel, split_data, crit, optimizer, epoch, data_dict, opt, split, **kw)
    elapsed = (time.time() - start) / 60
    return pred, targ, loss, elapsed
