"""PyTorch custom-operator registration of the C ABI (north star: "exposed to Python through PyTorch-ROCm custom ops").

    torch.ops.chromegcn.spmm                  torch.spmm(adj, support)                 models/SubLayers.py:46
    torch.ops.chromegcn.gated_layer           one gated graph-convolution layer        models/ChromeModels.py:37-40 (+:42)
    torch.ops.chromegcn.gated_layer_backward  its backward
    torch.ops.chromegcn.head_loss             relu/BatchNorm/dropout/Linear/BCE        models/ChromeModels.py:48-51, finetune.py:43-45,52
    torch.ops.chromegcn.head_loss_backward
    torch.ops.chromegcn.head_logits           eval-mode relu/BatchNorm/Linear per strand   models/ChromeModels.py:48-51 (model.eval())
    torch.ops.chromegcn.sgd_step              SGD(momentum, weight decay) in place     utils/util_methods.py:14-19

Pure-tensor signatures (the graph is passed as its CSR tensors), fake/meta implementations, and `register_autograd`
formulas whose backward is itself made of registered ops, so `torch.library.opcheck`, FakeTensor tracing and
`torch.compile` see the path.  Underneath every op is one ctypes call into libchromegcn_hip.so on torch's current HIP
stream -- the same entry points ops.py drives.  There is no CPU implementation: the ops are registered for
device_types="cuda" only, and dispatching them on CPU tensors raises.

`chromegcn_amd.layers` (ChromeGCN.forward / GraphConvolution) goes through these ops; the stage engine
(`finetune.GCNStage`) keeps its own autograd nodes in ops.py, which add what a pure-tensor op cannot express
(gradient sinks into the flat arena, the fused last-layer + head node, cached aggregations)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .graph import aux_ptr

_P = _lib.ptr


def _cuda_f32(t: Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError("chromegcn_amd: %s is on %s; the path only exists as HIP kernels (no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("chromegcn_amd: %s must be float32, got %s" % (name, t.dtype))


def _dense(t: Tensor) -> Tensor:
    t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t


# ------------------------------------------------------------------------------------------------------------------
# spmm
# ------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("chromegcn::spmm", mutates_args=(), device_types="cuda")
def spmm(x: Tensor, rowptr: Tensor, col: Tensor, val: Optional[Tensor], row_scale: Optional[Tensor],
         rowptr_t: Tensor, col_t: Tensor, val_t: Optional[Tensor]) -> Tensor:
    """Y[s] = diag(row_scale) Ahat X[s];  x: [S, n_cols, d] (S in {1,2}, d % 4 == 0).  rowptr_t / col_t / val_t: CSR of
    Ahat^T for the backward (the same tensors for the symmetric graphs the reference writes)."""
    _cuda_f32(x, "x")
    if x.dim() != 3:
        raise RuntimeError("chromegcn::spmm: x must be [S, n, d], got %s" % (tuple(x.shape),))
    x = _dense(x)
    S, n_cols, d = x.shape
    n_rows = rowptr.numel() - 1
    y = torch.empty((S, n_rows, d), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.cgcn_spmm(_lib.stream_ptr(), n_rows, n_cols, S, d, _P(rowptr), _P(col), _P(val), _P(row_scale),
                             x.data_ptr(), y.data_ptr(), aux_ptr(col)), "cgcn_spmm")
    return y


@spmm.register_fake
def _(x, rowptr, col, val, row_scale, rowptr_t, col_t, val_t):
    return x.new_empty((x.shape[0], rowptr.shape[0] - 1, x.shape[2]))


def _spmm_setup(ctx, inputs, output):
    x, rowptr, col, val, row_scale, rowptr_t, col_t, val_t = inputs
    ctx.save_for_backward(rowptr, col, val, row_scale, rowptr_t, col_t, val_t)


def _spmm_backward(ctx, dy):
    rowptr, col, val, row_scale, rowptr_t, col_t, val_t = ctx.saved_tensors
    if row_scale is not None:
        dy = dy * row_scale.view(1, -1, 1)       # A^T dY = Ahat^T (diag(row_scale) dY)
    dx = torch.ops.chromegcn.spmm(dy, rowptr_t, col_t, val_t, None, rowptr, col, val)
    return dx, None, None, None, None, None, None, None


spmm.register_autograd(_spmm_backward, setup_context=_spmm_setup)


# ------------------------------------------------------------------------------------------------------------------
# gated layer
# ------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("chromegcn::gated_layer", mutates_args=(), device_types="cuda")
def gated_layer(x: Tensor, weight: Tensor, bias: Tensor, gate_w: Tensor, gate_b: Tensor, rowptr: Tensor, col: Tensor,
                val: Optional[Tensor], row_scale: Optional[Tensor], rowptr_t: Tensor, col_t: Tensor,
                val_t: Optional[Tensor], dropout_out: float, dropout_in: float, rng_state: Optional[Tensor],
                layer_id: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """(X', gate, Z, H) of one gated layer: H = diag(row_scale) Ahat X, Z = tanh(H W + b), g = sigmoid(Z w + c),
    X' = dropout_out((1 - g) X + g Z).  Z and H are what the backward needs.  x: [S, n, d], d in {128, 256}."""
    for t, nm in ((x, "x"), (weight, "weight"), (bias, "bias"), (gate_w, "gate weight"), (gate_b, "gate bias")):
        _cuda_f32(t, nm)
    if x.dim() != 3 or x.shape[0] not in (1, 2) or x.shape[2] not in (128, 256):
        raise RuntimeError("chromegcn::gated_layer: x must be [S in {1,2}, n, d in {128,256}], got %s" % (tuple(x.shape),))
    x = _dense(x)
    S, n, d = x.shape
    if tuple(weight.shape) != (d, d) or rowptr.numel() != n + 1:
        raise RuntimeError("chromegcn::gated_layer: weight must be [d, d] and the graph must have n nodes")
    if (dropout_out > 0 or dropout_in > 0) and rng_state is None:
        raise RuntimeError("chromegcn::gated_layer: dropout needs the rng_state tensor")
    weight, bias = _dense(weight), bias.contiguous()
    wg, cg = gate_w.contiguous().view(-1), gate_b.contiguous().view(-1)
    xn, z, h = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    gate = torch.empty((S, n), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.cgcn_layer_fwd(_lib.stream_ptr(), n, S, d, _P(rowptr), _P(col), _P(val), _P(row_scale), x.data_ptr(),
                                  weight.data_ptr(), bias.data_ptr(), wg.data_ptr(), cg.data_ptr(), xn.data_ptr(),
                                  z.data_ptr(), h.data_ptr(), gate.data_ptr(), float(dropout_out),
                                  _P(rng_state) if dropout_out > 0 else None, int(layer_id), None, None, 0, aux_ptr(col)), "cgcn_layer_fwd")
    return xn, gate, z, h


@gated_layer.register_fake
def _(x, weight, bias, gate_w, gate_b, rowptr, col, val, row_scale, rowptr_t, col_t, val_t, dropout_out, dropout_in,
      rng_state, layer_id):
    return x.new_empty(x.shape), x.new_empty(x.shape[:2]), x.new_empty(x.shape), x.new_empty(x.shape)


@torch.library.custom_op("chromegcn::gated_layer_backward", mutates_args=(), device_types="cuda")
def gated_layer_backward(dxn: Tensor, dgate: Optional[Tensor], x: Tensor, z: Tensor, h: Tensor, gate: Tensor,
                         weight: Tensor, gate_w: Tensor, rowptr_t: Tensor, col_t: Tensor, val_t: Optional[Tensor],
                         row_scale: Optional[Tensor], dropout_in: float, rng_state: Optional[Tensor], layer_id: int,
                         need_dx: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """(dX, dW, db, dgate_w, dgate_b, dHs) given dL/dX' and (optionally) dL/dgate.  need_dx = False skips the gather
    over Ahat^T (dX comes back empty).  dHs = diag(row_scale) dL/dU W^T (the gather's and the saliency SDDMM's operand)."""
    dxn = _dense(dxn)
    x, z, h, weight = _dense(x), _dense(z), _dense(h), _dense(weight)
    S, n, d = x.shape
    dev = x.device
    f32 = dict(device=dev, dtype=torch.float32)
    dx = torch.empty_like(x) if need_dx else torch.empty(0, **f32)
    dhs = torch.empty_like(x)
    dw, db, dwg, dcg = torch.empty_like(weight), torch.empty(d, **f32), torch.empty(d, **f32), torch.empty(1, **f32)
    lib = _lib.load()
    ws_bytes = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    wg = gate_w.contiguous().view(-1)
    _lib.check(lib.cgcn_layer_bwd(_lib.stream_ptr(), n, S, d, _P(rowptr_t), _P(col_t), _P(val_t), _P(row_scale),
                                  x.data_ptr(), z.data_ptr(), h.data_ptr(), gate.contiguous().data_ptr(), weight.data_ptr(),
                                  wg.data_ptr(), dxn.data_ptr(), _P(None if dgate is None else dgate.contiguous()),
                                  dx.data_ptr() if need_dx else None, dhs.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                  dwg.data_ptr(), dcg.data_ptr(), 0, float(dropout_in), _P(rng_state) if dropout_in > 0 else None,
                                  max(int(layer_id) - 1, 0), None, ws.data_ptr(), ws_bytes, None, None, aux_ptr(col_t)), "cgcn_layer_bwd")
    return dx, dw, db, dwg, dcg, dhs


@gated_layer_backward.register_fake
def _(dxn, dgate, x, z, h, gate, weight, gate_w, rowptr_t, col_t, val_t, row_scale, dropout_in, rng_state, layer_id, need_dx):
    d = x.shape[2]
    return (x.new_empty(x.shape) if need_dx else x.new_empty(0), weight.new_empty(weight.shape), x.new_empty(d),
            x.new_empty(d), x.new_empty(1), x.new_empty(x.shape))


def _layer_setup(ctx, inputs, output):
    (x, weight, bias, gate_w, gate_b, rowptr, col, val, row_scale, rowptr_t, col_t, val_t, dropout_out, dropout_in,
     rng_state, layer_id) = inputs
    xn, gate, z, h = output
    ctx.save_for_backward(x, z, h, gate, weight, gate_w, rowptr_t, col_t, val_t, row_scale, rng_state)
    ctx.dropout_in, ctx.layer_id = float(dropout_in), int(layer_id)
    ctx.gate_w_shape, ctx.gate_b_shape = gate_w.shape, gate_b.shape
    ctx.mark_non_differentiable(z, h)     # saved activations handed to the backward, not differentiable results
    ctx.set_materialize_grads(False)


def _layer_backward(ctx, dxn, dgate, dz, dh):
    x, z, h, gate, weight, gate_w, rowptr_t, col_t, val_t, row_scale, rng_state = ctx.saved_tensors
    if dxn is None and dgate is None:
        return (None,) * 16
    if dxn is None:
        dxn = torch.zeros_like(x)
    dx, dw, db, dwg, dcg, _ = torch.ops.chromegcn.gated_layer_backward(
        dxn, dgate, x, z, h, gate, weight, gate_w, rowptr_t, col_t, val_t, row_scale, ctx.dropout_in, rng_state,
        ctx.layer_id, bool(ctx.needs_input_grad[0]))
    return (dx if ctx.needs_input_grad[0] else None, dw, db, dwg.view(ctx.gate_w_shape), dcg.view(ctx.gate_b_shape)) + (None,) * 11


gated_layer.register_autograd(_layer_backward, setup_context=_layer_setup)


# ------------------------------------------------------------------------------------------------------------------
# classifier head + loss
# ------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("chromegcn::head_loss", mutates_args=(), device_types="cuda")
def head_loss(x: Tensor, bn_w: Tensor, bn_b: Tensor, w_out: Tensor, b_out: Tensor, target: Tensor, run_mean: Tensor,
              run_var: Tensor, momentum: float, eps: float, training: bool, dropout_p: float,
              rng_state: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """(loss [], probs [n,C], save_mean [S,d], save_invstd [S,d], dpred [n,C], new_run_mean [d], new_run_var [d]) of
    relu -> BatchNorm1d -> dropout -> Linear on every strand, mean over strands, BCE-with-logits (mean), sigmoid.
    Functional (an operator with an autograd formula may not mutate its inputs): in training the updated running
    statistics -- forward strand first, then the reverse strand, exactly as two successive ChromeGCN.forward calls
    update them -- are RETURNED; `head_loss_module` copies them into the BatchNorm module and bumps
    num_batches_tracked.  dpred = d loss / d pred for the backward (empty in eval mode)."""
    for t, nm in ((x, "x"), (bn_w, "bn weight"), (bn_b, "bn bias"), (w_out, "out.weight"), (b_out, "out.bias"), (target, "target")):
        _cuda_f32(t, nm)
    x = _dense(x)
    S, n, d = x.shape
    C = w_out.shape[0]
    target = target.contiguous()
    if tuple(target.shape) != (n, C):
        raise RuntimeError("chromegcn::head_loss: target must be [n, C] = [%d, %d], got %s" % (n, C, tuple(target.shape)))
    bn_w, bn_b, w_out, b_out = _dense(bn_w), _dense(bn_b), _dense(w_out), b_out.contiguous()
    lib = _lib.load()
    ws_bytes = lib.cgcn_head_workspace_bytes(n, S, d, C)
    if ws_bytes == 0:
        raise RuntimeError("chromegcn::head_loss: unsupported shape S=%d n=%d d=%d C=%d" % (S, n, d, C))
    f32 = dict(device=x.device, dtype=torch.float32)
    ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
    probs, loss = torch.empty((n, C), **f32), torch.empty(1, **f32)
    dpred = torch.empty((n, C), **f32) if training else torch.empty(0, **f32)
    save_mean = torch.empty((S, d), **f32) if training else torch.empty(0, **f32)
    save_invstd = torch.empty((S, d), **f32) if training else torch.empty(0, **f32)
    drop = bool(training) and dropout_p > 0
    if drop and rng_state is None:
        raise RuntimeError("chromegcn::head_loss: dropout needs the rng_state tensor")
    new_rm, new_rv = run_mean.detach().clone().contiguous(), run_var.detach().clone().contiguous()  # the kernel updates these copies
    _lib.check(lib.cgcn_head_fwd(_lib.stream_ptr(), n, S, d, C, x.data_ptr(), bn_w.data_ptr(), bn_b.data_ptr(),
                                 new_rm.data_ptr(), new_rv.data_ptr(), None, float(momentum), float(eps),
                                 1 if training else 0, w_out.data_ptr(), b_out.data_ptr(), target.data_ptr(),
                                 float(dropout_p), _P(rng_state) if drop else None, probs.data_ptr(), loss.data_ptr(),
                                 dpred.data_ptr() if training else None, save_mean.data_ptr() if training else None,
                                 save_invstd.data_ptr() if training else None, ws.data_ptr(), ws_bytes), "cgcn_head_fwd")
    return loss.view(()), probs, save_mean, save_invstd, dpred, new_rm, new_rv


@head_loss.register_fake
def _(x, bn_w, bn_b, w_out, b_out, target, run_mean, run_var, momentum, eps, training, dropout_p, rng_state):
    S, n, d = x.shape
    C = w_out.shape[0]
    return (x.new_empty(()), x.new_empty((n, C)), x.new_empty((S, d)) if training else x.new_empty(0),
            x.new_empty((S, d)) if training else x.new_empty(0), x.new_empty((n, C)) if training else x.new_empty(0),
            run_mean.new_empty(run_mean.shape), run_var.new_empty(run_var.shape))


@torch.library.custom_op("chromegcn::head_loss_backward", mutates_args=(), device_types="cuda")
def head_loss_backward(dloss: Tensor, x: Tensor, bn_w: Tensor, bn_b: Tensor, w_out: Tensor, dpred: Tensor,
                       save_mean: Tensor, save_invstd: Tensor, dropout_p: float,
                       rng_state: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """(dX, dbn_w, dbn_b, dW_out, db_out) of head_loss for the upstream d loss (a scalar tensor)."""
    x, bn_w, bn_b, w_out = _dense(x), _dense(bn_w), _dense(bn_b), _dense(w_out)
    S, n, d = x.shape
    C = w_out.shape[0]
    f32 = dict(device=x.device, dtype=torch.float32)
    lib = _lib.load()
    ws_bytes = lib.cgcn_head_workspace_bytes(n, S, d, C)
    ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8)
    dx, dw_out = torch.empty_like(x), torch.empty_like(w_out)
    db_out, dbn_w, dbn_b = torch.empty(C, **f32), torch.empty(d, **f32), torch.empty(d, **f32)
    dl = dloss.contiguous().view(1)
    _lib.check(lib.cgcn_head_bwd(_lib.stream_ptr(), n, S, d, C, x.data_ptr(), bn_w.data_ptr(), bn_b.data_ptr(),
                                 save_mean.data_ptr(), save_invstd.data_ptr(), w_out.data_ptr(), dpred.contiguous().data_ptr(),
                                 dl.data_ptr(), float(dropout_p), _P(rng_state) if dropout_p > 0 else None, dx.data_ptr(),
                                 dw_out.data_ptr(), db_out.data_ptr(), dbn_w.data_ptr(), dbn_b.data_ptr(), 0,
                                 ws.data_ptr(), ws_bytes), "cgcn_head_bwd")
    return dx, dbn_w, dbn_b, dw_out, db_out


@head_loss_backward.register_fake
def _(dloss, x, bn_w, bn_b, w_out, dpred, save_mean, save_invstd, dropout_p, rng_state):
    return (x.new_empty(x.shape), bn_w.new_empty(bn_w.shape), bn_b.new_empty(bn_b.shape), w_out.new_empty(w_out.shape),
            w_out.new_empty(w_out.shape[0]))


def _head_setup(ctx, inputs, output):
    (x, bn_w, bn_b, w_out, b_out, target, run_mean, run_var, momentum, eps, training, dropout_p, rng_state) = inputs
    loss, probs, save_mean, save_invstd, dpred, _rm, _rv = output
    ctx.training = bool(training)
    ctx.dropout_p = float(dropout_p) if training else 0.0
    ctx.save_for_backward(x, bn_w, bn_b, w_out, dpred, save_mean, save_invstd, rng_state)
    ctx.mark_non_differentiable(probs, save_mean, save_invstd, dpred, _rm, _rv)   # only the loss is differentiable
    ctx.set_materialize_grads(False)


def _head_backward(ctx, dloss, dprobs, dsm, dsi, ddpred, drm, drv):
    from .ops import _EVAL_BWD_MSG
    if not ctx.training:
        raise RuntimeError(_EVAL_BWD_MSG)
    if dloss is None:
        return (None,) * 13
    x, bn_w, bn_b, w_out, dpred, save_mean, save_invstd, rng_state = ctx.saved_tensors
    dx, dbn_w, dbn_b, dw_out, db_out = torch.ops.chromegcn.head_loss_backward(
        dloss, x, bn_w, bn_b, w_out, dpred, save_mean, save_invstd, ctx.dropout_p, rng_state)
    return (dx, dbn_w, dbn_b, dw_out, db_out) + (None,) * 8


head_loss.register_autograd(_head_backward, setup_context=_head_setup)


@torch.library.custom_op("chromegcn::head_logits", mutates_args=(), device_types="cuda")
def head_logits(x: Tensor, bn_w: Tensor, bn_b: Tensor, run_mean: Tensor, run_var: Tensor, eps: float, w_out: Tensor,
                b_out: Tensor) -> Tensor:
    """logits[s] = BatchNorm1d(relu(x[s]); running statistics) W_out^T + b_out -- the head of ChromeGCN.forward with the
    module in eval mode, per strand, one kernel each (cgcn_head_logits).  x: [S, n, d] -> [S, n, C].  No autograd formula:
    layers.ChromeGCN._head takes it only when nothing needs a gradient."""
    for t, nm in ((x, "x"), (bn_w, "bn weight"), (bn_b, "bn bias"), (run_mean, "running_mean"), (run_var, "running_var"),
                  (w_out, "out.weight"), (b_out, "out.bias")):
        _cuda_f32(t, nm)
    if x.dim() != 3:
        raise RuntimeError("chromegcn::head_logits: x must be [S, n, d], got %s" % (tuple(x.shape),))
    x = _dense(x)
    S, n, d = x.shape
    C = w_out.shape[0]
    logits = torch.empty((S, n, C), dtype=torch.float32, device=x.device)
    # the dense copies (if any were needed) stay bound to names until the launch is enqueued: a temporary dropped
    # earlier hands its block back to the caching allocator, and the next copy of the same size would alias it
    bn_w, bn_b, run_mean, run_var, w_out, b_out = map(_dense, (bn_w, bn_b, run_mean, run_var, w_out, b_out))
    _lib.check(_lib.load().cgcn_head_logits(_lib.stream_ptr(), n, S, d, C, x.data_ptr(), bn_w.data_ptr(), bn_b.data_ptr(),
                                            run_mean.data_ptr(), run_var.data_ptr(), float(eps), w_out.data_ptr(),
                                            b_out.data_ptr(), logits.data_ptr()), "cgcn_head_logits")
    return logits


@head_logits.register_fake
def _(x, bn_w, bn_b, run_mean, run_var, eps, w_out, b_out):
    return x.new_empty((x.shape[0], x.shape[1], w_out.shape[0]))


def head_loss_module(x: Tensor, bn: torch.nn.BatchNorm1d, out: torch.nn.Linear, target: Tensor, training: bool,
                     dropout_p: float, rng_state: Optional[Tensor]):
    """nn.Module-level wrapper of chromegcn::head_loss: applies the returned running statistics to `bn` and counts the
    BatchNorm calls (one per strand) like the reference's two forward calls.  Returns (loss [], probs [n,C])."""
    if bn.momentum is None:
        raise RuntimeError("chromegcn_amd: BatchNorm momentum=None (cumulative average) is not supported by the fused head")
    loss, probs, _sm, _si, _dp, rm, rv = torch.ops.chromegcn.head_loss(
        x, bn.weight, bn.bias, out.weight, out.bias, target, bn.running_mean, bn.running_var, float(bn.momentum),
        float(bn.eps), bool(training), float(dropout_p), rng_state)
    if training:
        with torch.no_grad():
            bn.running_mean.copy_(rm)
            bn.running_var.copy_(rv)
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += x.shape[0]
    return loss, probs


# ------------------------------------------------------------------------------------------------------------------
# optimizer step
# ------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("chromegcn::sgd_step", mutates_args=("param", "momentum_buf", "rng_state"), device_types="cuda")
def sgd_step(param: Tensor, grad: Tensor, momentum_buf: Optional[Tensor], lr: float, momentum: float, weight_decay: float,
             nesterov: bool, grad_scale: float, rng_state: Optional[Tensor]) -> None:
    """torch.optim.SGD semantics on flat fp32 buffers in one launch: d = grad_scale g + wd p; m = mu m + d;
    p -= lr (nesterov ? d + mu m : m).  rng_state (optional): the dropout step counter, advanced by one."""
    if not (param.is_contiguous() and grad.is_contiguous() and param.numel() == grad.numel()):
        raise RuntimeError("chromegcn::sgd_step: param and grad must be contiguous and of equal size")
    lib = _lib.load()
    _lib.check(lib.cgcn_sgd_step(_lib.stream_ptr(), param.numel(), param.data_ptr(), grad.data_ptr(), _P(momentum_buf),
                                 float(lr), float(momentum), float(weight_decay), 1 if nesterov else 0, float(grad_scale),
                                 _P(rng_state)), "cgcn_sgd_step")


@sgd_step.register_fake
def _(param, grad, momentum_buf, lr, momentum, weight_decay, nesterov, grad_scale, rng_state):
    return None
