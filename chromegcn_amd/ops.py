"""torch.autograd bindings of the C ABI (include/chromegcn.h).  Plumbing only: every tensor
allocation is torch's caching allocator, every launch goes to torch's current HIP stream, so
all of it can be captured into a HIP graph.  No fallbacks: CPU tensors raise."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from . import graph as G
from .graph import ChromGraph

SUPPORTED_D = (128, 256)


def _require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(
            "chromegcn_amd: %s is on %s; the gated-GCN path only exists as HIP kernels "
            "(no CPU fallback) -- move the model and inputs to the GPU" % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("chromegcn_amd: %s must be float32, got %s" % (name, t.dtype))


def _dense(t: torch.Tensor) -> torch.Tensor:
    """contiguous and 16-byte aligned (the kernels use 8/16-byte row accesses): a view that starts in the middle of
    another tensor's storage is copied once instead of being rejected by the C ABI"""
    t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t


def _check_feat(x: torch.Tensor, g: ChromGraph, name="x", any_width=False):
    _require_cuda(x, name)
    ok_d = (x.dim() == 3 and x.shape[2] % 4 == 0 and 4 <= x.shape[2] <= 4096) if any_width else (x.dim() == 3 and x.shape[2] in SUPPORTED_D)
    if x.dim() != 3 or x.shape[0] not in (1, 2) or not ok_d:
        raise RuntimeError("chromegcn_amd: %s must be [S in {1,2}, n, d %s], got %s" %
                           (name, "a multiple of 4, <= 4096" if any_width else "in {128,256}", tuple(x.shape)))
    if x.shape[1] != g.n:
        raise RuntimeError("chromegcn_amd: %s has %d nodes but the graph has %d" % (name, x.shape[1], g.n))


def spmm(x, graph):
    """registered operator torch.ops.chromegcn.spmm (chromegcn_amd/torch_ops.py) on the graph's CSR tensors"""
    _check_feat(x, graph, any_width=True)
    from . import torch_ops  # noqa: F401  (registers the ops)
    return torch.ops.chromegcn.spmm(x, graph.rowptr, graph.col, graph.val, graph.row_scale, graph.rowptr_t, graph.col_t,
                                    graph.val_t)


def sddmm(a, b, graph: ChromGraph, transposed=False, out=None):
    """out[k] = sum_s <a[s,i,:], b[s,col[k],:]> on the graph's pattern (cgcn_sddmm).  No autograd.
    out given: the product is ADDED to it (the saliency sums one product per layer)."""
    _check_feat(a, graph, "a")
    _check_feat(b, graph, "b")
    a, b = _dense(a), _dense(b)
    S, n, d = a.shape
    rowptr, col = (graph.rowptr_t, graph.col_t) if transposed else (graph.rowptr, graph.col)
    acc = out is not None
    if acc:
        if out.shape != (col.shape[0],) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != a.device:
            raise RuntimeError("chromegcn_amd: sddmm accumulates into a contiguous fp32 [nnz] tensor on the features' device")
    else:
        out = torch.empty(col.shape[0], device=a.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.cgcn_sddmm(_lib.stream_ptr(), n, S, d, rowptr.data_ptr(), col.data_ptr(), a.data_ptr(), b.data_ptr(),
                              out.data_ptr(), 1 if acc else 0), "cgcn_sddmm")
    return out


def saliency_normalize(raw, graph: ChromGraph):
    """|val * raw| divided by its row sum, then by the row maximum of the quotients (scripts/visualize.py:49-55) on the
    graph's pattern, one launch (cgcn_saliency_normalize)."""
    _require_cuda(raw, "raw")
    raw = raw.contiguous()
    out = torch.empty_like(raw)
    lib = _lib.load()
    _lib.check(lib.cgcn_saliency_normalize(_lib.stream_ptr(), graph.n, graph.rowptr.data_ptr(), _lib.ptr(graph.val), raw.data_ptr(),
                                           out.data_ptr()), "cgcn_saliency_normalize")
    return out


_EVAL_BWD_MSG = ("chromegcn_amd: backward through the fused classifier head needs train mode (the eval-mode kernel "
                 "saves nothing for it); call model.train(), or use ChromeGCN.forward / forward_strands, whose "
                 "torch head differentiates in eval mode like the reference's")

_saliency_tap = None  # set by chromegcn_amd.saliency while it collects per-layer (X, dHs)

# Set by the stage engine around one train step (finetune.GCNStage): the flat parameter / gradient / momentum arenas and
# the SGD hyper-parameters.  The FIRST layer's backward -- the last launch of the step -- then carries the optimizer step
# in extra workgroups of its gather (or, without an input gradient, its partial-sum) launch (cgcn_sgd_fuse) and marks
# the request done; anything it cannot fuse (gradients not in the flat arena) is left for cgcn_sgd_step.
_sgd_fuse = None


def _sgd_fuse_arg(layer_id, dx, sink):
    """ctypes reference to a cgcn_sgd_fuse for this cgcn_layer_bwd call, or None (plus the struct to keep alive)"""
    rq = _sgd_fuse
    if rq is None or rq.get("done") or layer_id != 1 or sink is None or _lib.aux_stream_ptr() is not None:
        return None, None
    fg = rq["grad"]
    lo, hi = fg.data_ptr(), fg.data_ptr() + 4 * fg.numel()
    if not all(lo <= t.data_ptr() and t.data_ptr() + 4 * t.numel() <= hi for t in sink):
        return None, None
    sg = _lib.SgdFuse(rq["param"].data_ptr(), fg.data_ptr(), _lib.ptr(rq["mom"]), fg.numel(), float(rq["lr"]),
                      float(rq["momentum"]), float(rq["weight_decay"]), float(rq["grad_scale"]), 1 if rq["nesterov"] else 0,
                      _lib.ptr(rq["rng_state"]))
    return ctypes.byref(sg), sg


# feature tables from this size on do not fit the L2s: cgcn_layer_fwd's split route (FWD_SPLIT_TABLE_BYTES in csrc)
_SPLIT_TABLE_BYTES = 6 << 20


def _resolve_h_cache(h_cache, x, need_bwd):
    """h_cache: None, or a dict holder {'h': tensor-or-None} for H = A X of a layer whose input never changes
    (the engine keeps one per chromosome for the first layer).  Returns (H_in to stream, H buffer to write)."""
    if h_cache is not None and h_cache.get("h") is not None:
        hc = h_cache["h"]
        if hc.shape != x.shape or hc.device != x.device:
            raise RuntimeError("chromegcn_amd: cached aggregation does not match the input")
        return hc, None
    if need_bwd or h_cache is not None or (x.numel() * 4 >= _SPLIT_TABLE_BYTES and x.shape[0] * x.shape[2] <= 256):
        # inference on a large table too: with an H buffer cgcn_layer_fwd takes the feature-sliced two-launch route
        return None, torch.empty_like(x)
    return None, None


def _store_h_cache(h_cache, h_in, h):
    if h_in is not None:
        return h_in
    if h_cache is not None and h is not None:
        h_cache["h"] = h
    return h


def _out_slots(out_slots, n, C, device):
    """(probs [n,C], loss [1]) buffers for the fused head.  out_slots: None, or a holder {'probs': [n,C] view,
    'loss': [1] view} into caller-owned arenas (GCNStage lays every chromosome's predictions out in one buffer in
    chromosome order, so a split's concatenated predictions -- finetune.py:52 -- need no copy at all)."""
    if out_slots is not None:
        probs, loss = out_slots["probs"], out_slots["loss"]
        if tuple(probs.shape) != (n, C) or loss.numel() != 1 or not probs.is_contiguous() or probs.device != device:
            raise RuntimeError("chromegcn_amd: out_slots do not match [n, C] = [%d, %d]" % (n, C))
        return probs, loss.view(1)
    return (torch.empty((n, C), device=device, dtype=torch.float32), torch.empty(1, device=device, dtype=torch.float32))


def _sink_ok(sink, shapes):
    return sink is not None and all(t is not None and tuple(t.shape) == tuple(sh) and t.is_contiguous()
                                    for t, sh in zip(sink, shapes))


class GatedLayerFn(torch.autograd.Function):
    """One gated GCN layer (models/ChromeModels.py:37-40), fused; see cgcn_layer_fwd / cgcn_layer_bwd.

    dropout_out / dropout_in implement the F.dropout between layers (models/ChromeModels.py:42) inside the
    kernels: this layer's output is dropped with stream id `layer_id`; its input was dropped by the previous
    layer (stream id layer_id - 1), which the backward undoes on dX.  rng_state: uint64[2] device tensor
    {seed, step counter}, constant over one step.
    grad_sink (engine use): tensors (dW, db, dgate_w, dgate_b) the backward writes the parameter gradients
    INTO (overwrite); autograd then receives None for them, which skips its per-parameter accumulate kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, gate_w, gate_b, graph: ChromGraph, dropout_out, dropout_in, rng_state,
                layer_id, grad_sink, h_cache, zero_stat=None):
        _check_feat(x, graph)
        for t, nm in ((weight, "weight"), (bias, "bias"), (gate_w, "gate weight"), (gate_b, "gate bias")):
            _require_cuda(t, nm)
        x = _dense(x)
        S, n, d = x.shape
        if tuple(weight.shape) != (d, d):
            raise RuntimeError("chromegcn_amd: fused layer needs a square [d,d] weight, got %s" % (tuple(weight.shape),))
        weight = _dense(weight)
        bias = bias.contiguous()
        wg = gate_w.contiguous().view(-1)
        cg = gate_b.contiguous().view(-1)
        need_bwd = any(ctx.needs_input_grad[:5])
        xn = torch.empty_like(x)
        gate = torch.empty((S, n), device=x.device, dtype=torch.float32)
        z = torch.empty_like(x) if need_bwd else None
        h_in, h = _resolve_h_cache(h_cache, x, need_bwd)
        if (dropout_out > 0 or dropout_in > 0) and rng_state is None:
            raise RuntimeError("chromegcn_amd: fused dropout needs the model's rng_state tensor")
        lib = _lib.load()
        _lib.check(lib.cgcn_layer_fwd(_lib.stream_ptr(), n, S, d, _lib.ptr(graph.rowptr), _lib.ptr(graph.col),
                                      _lib.ptr(graph.val), _lib.ptr(graph.row_scale), x.data_ptr(), weight.data_ptr(),
                                      bias.data_ptr(), wg.data_ptr(), cg.data_ptr(), xn.data_ptr(), _lib.ptr(z),
                                      _lib.ptr(h), gate.data_ptr(), float(dropout_out),
                                      _lib.ptr(rng_state) if dropout_out > 0 else None, int(layer_id), _lib.ptr(h_in),
                                      # zero_stat: the NEXT (last) layer's statistics totals, zeroed by this call's first launch
                                      _lib.ptr(zero_stat), _lib.COLSTATS_ROWS_ZERO_ONLY if zero_stat is not None else 0,
                                      G.aux_ptr(graph.col)),
                   "cgcn_layer_fwd")
        h = _store_h_cache(h_cache, h_in, h)
        if need_bwd:
            ctx.save_for_backward(x, z, h, gate, weight, wg, rng_state if dropout_in > 0 else None)
        ctx.graph = graph
        ctx.gate_w_shape = gate_w.shape
        ctx.gate_b_shape = gate_b.shape
        ctx.dropout_in = float(dropout_in)
        ctx.layer_id = int(layer_id)
        ctx.sink = grad_sink if _sink_ok(grad_sink, ((d, d), (d,), gate_w.shape, gate_b.shape)) else None
        ctx.set_materialize_grads(False)  # an unused gate output must not cost a zero-fill + an extra read
        return xn, gate

    @staticmethod
    def backward(ctx, dxn, dgate):
        x, z, h, gate, weight, wg, rng_state = ctx.saved_tensors
        g = ctx.graph
        S, n, d = x.shape
        if dxn is None and dgate is None:
            return (None,) * 13
        dxn = torch.zeros_like(x) if dxn is None else _dense(dxn)
        dgate = None if dgate is None else dgate.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None  # None: skip the gather over Ahat^T
        # dHs = diag(row_scale) dL/dU W^T: the gather's operand (and the saliency SDDMM's); not computed when unused
        dhs = torch.empty_like(x) if (dx is not None or _saliency_tap is not None) else None
        if ctx.sink is not None:
            dw, db, dwg, dcg = ctx.sink
        else:
            dw = torch.empty_like(weight)
            db = torch.empty(d, device=x.device, dtype=torch.float32)
            dwg = torch.empty(d, device=x.device, dtype=torch.float32)
            dcg = torch.empty(1, device=x.device, dtype=torch.float32)
        lib = _lib.load()
        ws_bytes = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
        ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
        sg_ref, sg_keep = _sgd_fuse_arg(ctx.layer_id, dx, ctx.sink)
        _lib.check(lib.cgcn_layer_bwd(_lib.stream_ptr(), n, S, d, _lib.ptr(g.rowptr_t), _lib.ptr(g.col_t),
                                      _lib.ptr(g.val_t), _lib.ptr(g.row_scale), x.data_ptr(), z.data_ptr(),
                                      h.data_ptr(), gate.data_ptr(), weight.data_ptr(), wg.data_ptr(),
                                      dxn.data_ptr(), _lib.ptr(dgate), _lib.ptr(dx), _lib.ptr(dhs), dw.data_ptr(),
                                      db.data_ptr(), dwg.data_ptr(), dcg.data_ptr(), 0, ctx.dropout_in,
                                      _lib.ptr(rng_state), max(ctx.layer_id - 1, 0), None, ws.data_ptr(), ws_bytes,
                                      _lib.aux_stream_ptr(), sg_ref, G.aux_ptr(g.col_t)),
                   "cgcn_layer_bwd")
        if sg_ref is not None:
            _sgd_fuse["done"] = True
        if _saliency_tap is not None:
            _saliency_tap.append((x, dhs, g))
        if ctx.sink is not None:
            return (dx, None, None, None, None) + (None,) * 8
        return (dx, dw, db, dwg.view(ctx.gate_w_shape), dcg.view(ctx.gate_b_shape)) + (None,) * 8


def gated_layer(x, weight, bias, gate_w, gate_b, graph, dropout_out=0.0, dropout_in=0.0, rng_state=None,
                layer_id=0, grad_sink=None, h_cache=None, zero_stat=None):
    """One gated layer -> (X', gate).  Plain callers (ChromeGCN.forward, tests) go through the registered operator
    torch.ops.chromegcn.gated_layer; the engine's extras (gradient sinks, cached aggregation, the saliency tap) need
    the autograd node below.
    zero_stat: the statistics-totals buffer of the LAST layer's call (stat_buffer), zeroed by this call's first launch so
    that the last layer may accumulate into it on any route (cgcn_layer_fwd, colstats_rows = -2 / -3)."""
    if grad_sink is None and h_cache is None and _saliency_tap is None and zero_stat is None:
        _check_feat(x, graph)
        from . import torch_ops  # noqa: F401
        xn, gate, _z, _h = torch.ops.chromegcn.gated_layer(
            x, weight, bias, gate_w, gate_b, graph.rowptr, graph.col, graph.val, graph.row_scale, graph.rowptr_t,
            graph.col_t, graph.val_t, float(dropout_out), float(dropout_in), rng_state, int(layer_id))
        return xn, gate
    return GatedLayerFn.apply(x, weight, bias, gate_w, gate_b, graph, float(dropout_out), float(dropout_in),
                              rng_state, int(layer_id), grad_sink, h_cache, zero_stat)


def stat_buffer(x):
    """The accumulate-mode statistics buffer for features x [S, n, d] (include/chromegcn.h, cgcn_layer_fwd_colstats_plan with
    CGCN_COLSTATS_ACCUMULATE), or None when the shape has none (n < 2).  Uninitialised: a launch of the step zeroes it."""
    S, n, d = x.shape
    rows = ctypes.c_int(0)
    tiles = _lib.load().cgcn_layer_fwd_colstats_plan(n, S, d, _lib.COLSTATS_ACCUMULATE, ctypes.byref(rows))
    if tiles <= 0 or rows.value != -1:
        return None
    return torch.empty((tiles, S, d, 2), device=x.device, dtype=torch.float32)


class HeadLossFn(torch.autograd.Function):
    """relu -> BatchNorm1d -> dropout -> Linear, mean over strands, BCE-with-logits, sigmoid -- fused
    (models/ChromeModels.py:48-51 + finetune.py:43,45,52); see cgcn_head_fwd / cgcn_head_bwd.
    Returns (loss [], probs [n,C]).  Running statistics / num_batches_tracked are updated in place when
    training, exactly as two successive ChromeGCN.forward calls would.  grad_sink: (dbn_w, dbn_b, dW_out, db_out)."""

    @staticmethod
    def forward(ctx, x, bn_w, bn_b, w_out, b_out, target, run_mean, run_var, nbt, momentum, eps, training,
                dropout_p, rng_state, grad_sink, out_slots=None):
        _require_cuda(x, "x")
        for t, nm in ((bn_w, "bn weight"), (bn_b, "bn bias"), (w_out, "out.weight"), (b_out, "out.bias"), (target, "target")):
            _require_cuda(t, nm)
        if momentum is None:
            raise RuntimeError("chromegcn_amd: BatchNorm momentum=None (cumulative average) is not supported by the fused head")
        x = _dense(x)
        S, n, d = x.shape
        C = w_out.shape[0]
        target = target.contiguous()
        if tuple(target.shape) != (n, C):
            raise RuntimeError("chromegcn_amd: target must be [n, C] = [%d, %d], got %s" % (n, C, tuple(target.shape)))
        bn_w, bn_b, w_out, b_out = _dense(bn_w), _dense(bn_b), _dense(w_out), b_out.contiguous()
        lib = _lib.load()
        ws_bytes = lib.cgcn_head_workspace_bytes(n, S, d, C)
        if ws_bytes == 0:
            raise RuntimeError("chromegcn_amd: fused head does not support S=%d n=%d d=%d C=%d" % (S, n, d, C))
        ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
        need_bwd = training and any(ctx.needs_input_grad[:5])
        ctx.eval_grad = (not training) and any(ctx.needs_input_grad[:5])
        probs, loss = _out_slots(out_slots, n, C, x.device)
        dpred = torch.empty((n, C), device=x.device, dtype=torch.float32) if need_bwd else None
        save_mean = torch.empty((S, d), device=x.device, dtype=torch.float32) if training else None
        save_invstd = torch.empty((S, d), device=x.device, dtype=torch.float32) if training else None
        drop = bool(training) and dropout_p > 0
        if drop and rng_state is None:
            raise RuntimeError("chromegcn_amd: fused dropout needs the model's rng_state tensor")
        _lib.check(lib.cgcn_head_fwd(_lib.stream_ptr(), n, S, d, C, x.data_ptr(), bn_w.data_ptr(), bn_b.data_ptr(),
                                     run_mean.data_ptr(), run_var.data_ptr(), _lib.ptr(nbt), float(momentum), float(eps),
                                     1 if training else 0, w_out.data_ptr(), b_out.data_ptr(), target.data_ptr(),
                                     float(dropout_p), _lib.ptr(rng_state) if drop else None,
                                     probs.data_ptr(), loss.data_ptr(), _lib.ptr(dpred), _lib.ptr(save_mean),
                                     _lib.ptr(save_invstd), ws.data_ptr(), ws_bytes), "cgcn_head_fwd")
        if need_bwd:
            ctx.save_for_backward(x, bn_w, bn_b, w_out, dpred, save_mean, save_invstd, rng_state if drop else None)
            ctx.dropout_p = float(dropout_p) if drop else 0.0
            ctx.sink = grad_sink if _sink_ok(grad_sink, ((d,), (d,), (C, d), (C,))) else None
        ctx.mark_non_differentiable(probs)
        return loss.view(()), probs

    @staticmethod
    def backward(ctx, dloss, _dprobs):
        if ctx.eval_grad:
            raise RuntimeError(_EVAL_BWD_MSG)
        x, bn_w, bn_b, w_out, dpred, save_mean, save_invstd, rng_state = ctx.saved_tensors
        S, n, d = x.shape
        C = w_out.shape[0]
        lib = _lib.load()
        ws_bytes = lib.cgcn_head_workspace_bytes(n, S, d, C)
        ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
        dx = torch.empty_like(x)
        if ctx.sink is not None:
            dbn_w, dbn_b, dw_out, db_out = ctx.sink
        else:
            dw_out = torch.empty_like(w_out)
            db_out = torch.empty(C, device=x.device, dtype=torch.float32)
            dbn_w = torch.empty(d, device=x.device, dtype=torch.float32)
            dbn_b = torch.empty(d, device=x.device, dtype=torch.float32)
        dloss = dloss.contiguous().view(1)
        _lib.check(lib.cgcn_head_bwd(_lib.stream_ptr(), n, S, d, C, x.data_ptr(), bn_w.data_ptr(), bn_b.data_ptr(),
                                     save_mean.data_ptr(), save_invstd.data_ptr(), w_out.data_ptr(), dpred.data_ptr(),
                                     dloss.data_ptr(), ctx.dropout_p, _lib.ptr(rng_state), dx.data_ptr(),
                                     dw_out.data_ptr(), db_out.data_ptr(), dbn_w.data_ptr(), dbn_b.data_ptr(), 0,
                                     ws.data_ptr(), ws_bytes), "cgcn_head_bwd")
        if ctx.sink is not None:
            return (dx,) + (None,) * 15
        return (dx, dbn_w, dbn_b, dw_out, db_out) + (None,) * 11


class LastLayerHeadLossFn(torch.autograd.Function):
    """Last gated layer + classifier head + loss as ONE autograd node (engine path).  Forward = cgcn_layer_fwd
    then cgcn_head_fwd.  Backward = cgcn_head_bwd in deferred mode, then cgcn_layer_bwd in head mode: the
    gradient w.r.t. the layer's output never exists in memory as a tensor of its own (include/chromegcn.h,
    cgcn_head_grad).  Inputs/semantics as GatedLayerFn + HeadLossFn; returns (loss [], probs [n,C], gate [S,n])."""

    @staticmethod
    def forward(ctx, x, weight, bias, gate_w, gate_b, bn_w, bn_b, w_out, b_out, graph, target, run_mean, run_var, nbt,
                momentum, eps, training, dropout_p, dropout_in, rng_state, layer_id, layer_sink, head_sink, h_cache,
                out_slots=None, stat_acc=False, stat_buf=None):
        _check_feat(x, graph)
        for t, nm in ((weight, "weight"), (bias, "bias"), (gate_w, "gate weight"), (gate_b, "gate bias"),
                      (bn_w, "bn weight"), (bn_b, "bn bias"), (w_out, "out.weight"), (b_out, "out.bias"), (target, "target")):
            _require_cuda(t, nm)
        if momentum is None:
            raise RuntimeError("chromegcn_amd: BatchNorm momentum=None (cumulative average) is not supported by the fused head")
        x = _dense(x)
        S, n, d = x.shape
        C = w_out.shape[0]
        target = target.contiguous()
        if tuple(weight.shape) != (d, d) or tuple(target.shape) != (n, C):
            raise RuntimeError("chromegcn_amd: bad shapes for the fused last layer + head")
        weight, bias = _dense(weight), bias.contiguous()
        wg, cg = gate_w.contiguous().view(-1), gate_b.contiguous().view(-1)
        bn_w, bn_b, w_out, b_out = _dense(bn_w), _dense(bn_b), _dense(w_out), b_out.contiguous()
        need_bwd = training and any(ctx.needs_input_grad[:9])
        ctx.eval_grad = (not training) and any(ctx.needs_input_grad[:9])
        lib = _lib.load()
        xn = torch.empty_like(x)
        gate = torch.empty((S, n), device=x.device, dtype=torch.float32)
        z = torch.empty_like(x) if need_bwd else None
        h_in, h = _resolve_h_cache(h_cache, x, need_bwd)
        colstats, cs_tiles, cs_rows = None, 0, 0
        if need_bwd:
            # the layer kernel also emits the first stage of the head's BatchNorm statistics (tile still on chip)
            # stat_acc: the caller vouches for the range of the fixed-point totals (include/chromegcn.h,
            # cgcn_layer_fwd_colstats_plan; GCNStage does, per chromosome) -> accumulate mode: no finalize / finish launches;
            # otherwise per-workgroup records.  Accumulate mode needs the two-launch route, i.e. an H buffer of this call.
            # stat_buf: totals an EARLIER launch of this step has zeroed (the previous layer's forward: gated_layer(zero_stat=...));
            # this layer then accumulates on whatever route its table size gives it -- the fused kernel for small tables.
            rows = ctypes.c_int(0)
            fwd_rows = None
            if stat_acc and stat_buf is not None:
                colstats, cs_tiles, cs_rows = stat_buf, stat_buf.shape[0], -1
                fwd_rows = _lib.COLSTATS_ROWS_ACCUMULATE_ZEROED
            else:
                mode = _lib.COLSTATS_ACCUMULATE if (stat_acc and (h is not None or h_in is not None)) else _lib.COLSTATS_RECORDS
                cs_tiles = lib.cgcn_layer_fwd_colstats_plan(n, S, d, mode, ctypes.byref(rows))
                cs_rows = rows.value
                colstats = torch.empty((cs_tiles, S, d, 2), device=x.device, dtype=torch.float32)
        _lib.check(lib.cgcn_layer_fwd(_lib.stream_ptr(), n, S, d, _lib.ptr(graph.rowptr), _lib.ptr(graph.col),
                                      _lib.ptr(graph.val), _lib.ptr(graph.row_scale), x.data_ptr(), weight.data_ptr(),
                                      bias.data_ptr(), wg.data_ptr(), cg.data_ptr(), xn.data_ptr(), _lib.ptr(z),
                                      _lib.ptr(h), gate.data_ptr(), 0.0, None, int(layer_id), _lib.ptr(h_in), _lib.ptr(colstats),
                                      cs_rows if (not need_bwd or fwd_rows is None) else fwd_rows, G.aux_ptr(graph.col)), "cgcn_layer_fwd")
        h = _store_h_cache(h_cache, h_in, h)
        ws_bytes = lib.cgcn_head_workspace_bytes(n, S, d, C)
        if ws_bytes == 0:
            raise RuntimeError("chromegcn_amd: fused head does not support S=%d n=%d d=%d C=%d" % (S, n, d, C))
        ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
        probs, loss = _out_slots(out_slots, n, C, x.device)
        save_mean = torch.empty((S, d), device=x.device, dtype=torch.float32) if training else None
        save_invstd = torch.empty((S, d), device=x.device, dtype=torch.float32) if training else None
        drop = bool(training) and dropout_p > 0
        if (drop or dropout_in > 0) and rng_state is None:
            raise RuntimeError("chromegcn_amd: fused dropout needs the model's rng_state tensor")
        if need_bwd:
            # forward of the head + the tile-local half of its backward in one pass (cgcn_head_train); the workspace
            # carries dym and the partial sums to backward()
            _lib.check(lib.cgcn_head_train(_lib.stream_ptr(), n, S, d, C, xn.data_ptr(), bn_w.data_ptr(), bn_b.data_ptr(),
                                           run_mean.data_ptr(), run_var.data_ptr(), _lib.ptr(nbt), float(momentum), float(eps),
                                           w_out.data_ptr(), b_out.data_ptr(), target.data_ptr(), float(dropout_p) if drop else 0.0,
                                           _lib.ptr(rng_state) if drop else None, probs.data_ptr(), loss.data_ptr(),
                                           save_mean.data_ptr(), save_invstd.data_ptr(), _lib.ptr(colstats), cs_tiles, cs_rows,
                                           ws.data_ptr(), ws_bytes), "cgcn_head_train")
        else:
            _lib.check(lib.cgcn_head_fwd(_lib.stream_ptr(), n, S, d, C, xn.data_ptr(), bn_w.data_ptr(), bn_b.data_ptr(),
                                         run_mean.data_ptr(), run_var.data_ptr(), _lib.ptr(nbt), float(momentum), float(eps),
                                         1 if training else 0, w_out.data_ptr(), b_out.data_ptr(), target.data_ptr(),
                                         float(dropout_p), _lib.ptr(rng_state) if drop else None, probs.data_ptr(),
                                         loss.data_ptr(), None, _lib.ptr(save_mean), _lib.ptr(save_invstd),
                                         ws.data_ptr(), ws_bytes), "cgcn_head_fwd")
        if need_bwd:
            ctx.save_for_backward(x, z, h, gate, weight, wg, xn, bn_w, bn_b, w_out, ws, save_mean, save_invstd,
                                  rng_state if (drop or dropout_in > 0) else None,
                                  # accumulate mode (ABI v23, rows = -1): the backward reads the BatchNorm-backward column sums
                                  # from the statistics buffer (cgcn_head_grad.stat_acc), not from the workspace's bnc
                                  colstats if cs_rows == -1 else None)
            ctx.graph = graph
            ctx.dropout_p = float(dropout_p) if drop else 0.0
            ctx.dropout_in = float(dropout_in)
            ctx.layer_id = int(layer_id)
            ctx.shapes = (gate_w.shape, gate_b.shape)
            ctx.layer_sink = layer_sink if _sink_ok(layer_sink, ((d, d), (d,), gate_w.shape, gate_b.shape)) else None
            ctx.head_sink = head_sink if _sink_ok(head_sink, ((d,), (d,), (C, d), (C,))) else None
        ctx.mark_non_differentiable(probs, gate)
        ctx.set_materialize_grads(False)  # no zero-fill kernels for the two non-differentiable outputs
        return loss.view(()), probs, gate

    @staticmethod
    def backward(ctx, dloss, _dprobs, _dgate):
        if ctx.eval_grad:
            raise RuntimeError(_EVAL_BWD_MSG)
        (x, z, h, gate, weight, wg, xn, bn_w, bn_b, w_out, hws, save_mean, save_invstd, rng_state, stat_acc) = ctx.saved_tensors
        g = ctx.graph
        S, n, d = x.shape
        C = w_out.shape[0]
        dev = x.device
        lib = _lib.load()
        f32 = dict(device=dev, dtype=torch.float32)
        if ctx.head_sink is not None:
            dbn_w, dbn_b, dw_out, db_out = ctx.head_sink
        else:
            dw_out, db_out = torch.empty_like(w_out), torch.empty(C, **f32)
            dbn_w, dbn_b = torch.empty(d, **f32), torch.empty(d, **f32)
        if ctx.layer_sink is not None:
            dw, db, dwg, dcg = ctx.layer_sink
        else:
            dw, db, dwg, dcg = torch.empty_like(weight), torch.empty(d, **f32), torch.empty(d, **f32), torch.empty(1, **f32)
        hws_bytes = hws.numel()
        if dloss is None:
            return (None,) * 27
        dloss = dloss.contiguous().view(1)
        # dym, bnc and the partials are already in the workspace cgcn_head_train filled (for d loss = 1); every head
        # gradient is finished inside cgcn_layer_bwd (cgcn_head_grad.dloss / dbn_w / dbn_b), so no head launch here
        o_dym, o_bnc, o_part = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(lib.cgcn_head_workspace_layout(n, S, d, C, ctypes.byref(o_dym), ctypes.byref(o_bnc), ctypes.byref(o_part)),
                   "cgcn_head_workspace_layout")
        hg = _lib.HeadGrad(hws.data_ptr() + o_dym.value, hws.data_ptr() + o_bnc.value, save_mean.data_ptr(),
                           save_invstd.data_ptr(), bn_w.data_ptr(), ctx.dropout_p, _lib.ptr(rng_state),
                           hws.data_ptr() + o_part.value, lib.cgcn_head_bwd_partials(n), C, dw_out.data_ptr(),
                           db_out.data_ptr(), 0, dloss.data_ptr(), dbn_w.data_ptr(), dbn_b.data_ptr(), _lib.ptr(stat_acc))
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dhs = torch.empty_like(x) if dx is not None else None
        ws_bytes = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        # a one-layer model: this IS the first layer's backward, i.e. the last launch of the step
        sg_ref, sg_keep = _sgd_fuse_arg(ctx.layer_id, dx, ctx.layer_sink if ctx.head_sink is not None else None)
        _lib.check(lib.cgcn_layer_bwd(_lib.stream_ptr(), n, S, d, _lib.ptr(g.rowptr_t), _lib.ptr(g.col_t),
                                      _lib.ptr(g.val_t), _lib.ptr(g.row_scale), x.data_ptr(), z.data_ptr(),
                                      h.data_ptr(), gate.data_ptr(), weight.data_ptr(), wg.data_ptr(),
                                      None, None, _lib.ptr(dx), _lib.ptr(dhs), dw.data_ptr(), db.data_ptr(),
                                      dwg.data_ptr(), dcg.data_ptr(), 0, ctx.dropout_in, _lib.ptr(rng_state),
                                      max(ctx.layer_id - 1, 0), ctypes.byref(hg), ws.data_ptr(), ws_bytes,
                                      _lib.aux_stream_ptr(), sg_ref, G.aux_ptr(g.col_t)), "cgcn_layer_bwd")
        if sg_ref is not None:
            _sgd_fuse["done"] = True
        gl = (None,) * 4 if ctx.layer_sink is not None else (dw, db, dwg.view(ctx.shapes[0]), dcg.view(ctx.shapes[1]))
        gh = (None,) * 4 if ctx.head_sink is not None else (dbn_w, dbn_b, dw_out, db_out)
        return (dx,) + gl + gh + (None,) * 18


def last_layer_head_loss(x, gc, wk, bn, out, graph, target, training, dropout_p, dropout_in, rng_state, layer_id,
                         layer_sink=None, head_sink=None, h_cache=None, out_slots=None, stat_acc=False, stat_buf=None):
    """stat_acc: accumulate mode of the head's BatchNorm sums (fixed-point integer totals, two launches fewer per step).  Its
    range is finite (sum relu(x)^2 < 2.1e9 per column) and outside it the loss is NaN, so it is the caller's statement about
    its inputs: False (records, any magnitude) unless the caller has bounded them -- finetune.GCNStage does per chromosome.
    stat_buf (with stat_acc): stat_buffer(x) that an earlier launch of this step zeroes (gated_layer(zero_stat=stat_buf) of the
    layer before): the last layer then keeps the route its table size gives it (small tables: the one-launch forward)."""
    return LastLayerHeadLossFn.apply(x, gc.weight, gc.bias, wk.weight, wk.bias, bn.weight, bn.bias, out.weight, out.bias,
                                     graph, target, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum,
                                     bn.eps, bool(training), float(dropout_p), float(dropout_in), rng_state, int(layer_id),
                                     layer_sink, head_sink, h_cache, out_slots, bool(stat_acc), stat_buf)


def head_loss(x, bn: torch.nn.BatchNorm1d, out: torch.nn.Linear, target, training, dropout_p, rng_state, grad_sink=None,
              out_slots=None):
    if grad_sink is None and out_slots is None:
        from . import torch_ops
        _require_cuda(x, "x")
        return torch_ops.head_loss_module(x, bn, out, target, training, dropout_p, rng_state)  # torch.ops.chromegcn.head_loss
    return HeadLossFn.apply(x, bn.weight, bn.bias, out.weight, out.bias, target, bn.running_mean, bn.running_var,
                            bn.num_batches_tracked, bn.momentum, bn.eps, bool(training), float(dropout_p), rng_state,
                            grad_sink, out_slots)


def head_logits(x, bn: torch.nn.BatchNorm1d, out: torch.nn.Linear):
    """The eval-mode head of ChromeGCN.forward per strand (models/ChromeModels.py:48-51 with running statistics and
    dropout off): relu -> BatchNorm1d -> Linear in one kernel (torch.ops.chromegcn.head_logits -> cgcn_head_logits).
    x: [S, n, d] -> logits [S, n, C].  No autograd: callers use it when nothing needs a gradient (layers.ChromeGCN._head)."""
    _require_cuda(x, "x")
    from . import torch_ops  # noqa: F401  (registers the ops)
    return torch.ops.chromegcn.head_logits(x.detach(), bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                           float(bn.eps), out.weight.detach(), out.bias.detach())


def sgd_step(flat_param, flat_grad, flat_mom, lr, momentum, weight_decay, nesterov, rng_state=None, grad_scale=1.0):
    """torch.optim.SGD step on flat buffers in one launch (cgcn_sgd_step); also advances the dropout counter."""
    lib = _lib.load()
    _lib.check(lib.cgcn_sgd_step(_lib.stream_ptr(), flat_param.numel(), flat_param.data_ptr(), flat_grad.data_ptr(),
                                 _lib.ptr(flat_mom), float(lr), float(momentum), float(weight_decay),
                                 1 if nesterov else 0, float(grad_scale), _lib.ptr(rng_state)), "cgcn_sgd_step")
