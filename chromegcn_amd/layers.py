"""Host-side mirror of the reference's module surface for the hot path:

    GraphConvolution(in_features, out_features, bias=True, init='xavier').forward(input, adj, deg)
        -- models/SubLayers.py:7-52
    ChromeGCN(nfeat, nhid, nclass, dropout, gate, layers).forward(x_in, adj, deg, src_dict=None,
        return_gate=False) -> (x_in, out, (g, g2), None)
        -- models/ChromeModels.py:21-52

Same constructor arguments, forward signatures, return tuples and state_dict keys, so reference
checkpoints load and reference callers (finetune.py:41-42) run unchanged.  The math runs in the
HIP kernels behind include/chromegcn.h; there is no CPU path."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import as_graph


class GraphConvolution(nn.Module):
    """output = adj @ (input @ weight) + bias   (models/SubLayers.py:42-52).  `deg` is accepted and
    ignored, as in the reference.  adj: ChromGraph, torch sparse COO, or None (-> input @ weight + bias)."""

    def __init__(self, in_features, out_features, bias=True, init="xavier"):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        if init == "uniform":  # SubLayers.py:26-30 (the reference forgets to import math here)
            stdv = 1.0 / math.sqrt(self.weight.size(1))
            nn.init.uniform_(self.weight, -stdv, stdv)
            if self.bias is not None:
                nn.init.uniform_(self.bias, -stdv, stdv)
        elif init == "xavier":  # SubLayers.py:32-35
            nn.init.xavier_normal_(self.weight, gain=0.02)
            if self.bias is not None:
                nn.init.zeros_(self.bias)
        elif init == "kaiming":  # SubLayers.py:37-40
            nn.init.kaiming_normal_(self.weight, a=0, mode="fan_in")
            if self.bias is not None:
                nn.init.zeros_(self.bias)
        else:
            raise NotImplementedError

    def forward(self, input, adj, deg=None):
        ops._require_cuda(input, "input")
        support = torch.mm(input, self.weight)                              # SubLayers.py:43
        if adj is not None:
            g = as_graph(adj, input.device)
            w = support.shape[1]
            if w % 4:   # the kernels move 16-byte column groups: pad to the next multiple of 4 (e.g. out_features = C
                support = F.pad(support, (0, 4 - w % 4))                    # = 103 labels), aggregate, cut back
            output = ops.spmm(support.unsqueeze(0), g).squeeze(0)           # SubLayers.py:46
            if w % 4:
                output = output[:, :w]
        else:
            output = support                                                # SubLayers.py:48
        return output + self.bias if self.bias is not None else output

    def __repr__(self):
        return "%s (%d -> %d)" % (self.__class__.__name__, self.in_features, self.out_features)


class ChromeGCN(nn.Module):
    """Gated GCN over one chromosome (models/ChromeModels.py:21-52).

    `layers`: the reference builds a second layer only when layers == 2 and silently a 1-layer model
    for every other value (:25); here layers = L builds L gated layers (GC1/W1 ... GCL/WL, extending
    the reference's naming), unless reference_layer_rule=True.  `gate` is accepted and ignored
    exactly as in the reference (:22-31)."""

    def __init__(self, nfeat, nhid, nclass, dropout, gate=True, layers=2, reference_layer_rule=False):
        super().__init__()
        if nfeat != nhid:
            raise ValueError("ChromeGCN needs nfeat == nhid (the gated residual adds x and z)")
        if reference_layer_rule:
            layers = 2 if layers == 2 else 1
        if layers < 1:
            raise ValueError("layers must be >= 1")
        self.n_layers = int(layers)
        for k in range(1, self.n_layers + 1):
            setattr(self, "GC%d" % k, GraphConvolution(nfeat, nhid, bias=True, init="xavier"))
            setattr(self, "W%d" % k, nn.Linear(nfeat, 1))
        self.dropout = dropout
        self.batch_norm = nn.BatchNorm1d(nfeat)
        self.out = nn.Linear(nfeat, nclass)
        # {seed, step counter} of the counter-based dropout RNG used by the fused kernels; lives on the
        # device so captured HIP graphs draw fresh masks on every replay.  Not part of the state_dict
        # (the reference has no such key).
        self.register_buffer("_rng_state", torch.tensor([0x5DEECE66D, 0], dtype=torch.int64), persistent=False)

    def seed_dropout(self, seed: int):
        """reseed the fused-kernel dropout stream (torch.manual_seed does not reach it)"""
        self._rng_state[0] = int(seed) & 0x7FFFFFFFFFFFFFFF
        self._rng_state[1] = 0

    # -- dropout RNG bookkeeping ---------------------------------------------------------------
    def _step_rng(self):
        """rng_state tensor for this forward.  Stand-alone use: a snapshot is taken and the live counter is
        advanced, so every forward draws fresh masks and its backward still sees the counter it used.  Under
        GCNStage (`_rng_managed`), the live tensor is used and the engine advances it once per step."""
        if not (self.training and self.dropout > 0):
            return None
        if getattr(self, "_rng_managed", False):
            return self._rng_state
        snap = self._rng_state.clone()
        self._rng_state[1] += 1
        return snap

    def _sink(self, *params):
        """gradient sinks (engine use): the .grad views the backward kernels write straight into"""
        if not getattr(self, "_grad_sink", False):
            return None
        grads = tuple(p.grad for p in params)
        return None if any(g is None for g in grads) else grads

    # -- the gated stack on a [S, n, d] block -------------------------------------------------
    def _gated_stack(self, x, graph, rng, upto=None, h1_cache=None, zero_stat=None):
        gates = []
        L = self.n_layers
        p = float(self.dropout) if (self.training and rng is not None) else 0.0
        last = L if upto is None else upto
        for k in range(1, last + 1):
            gc = getattr(self, "GC%d" % k)
            wk = getattr(self, "W%d" % k)
            # F.dropout between layers (ChromeModels.py:42) runs inside the kernels: layer k drops its own
            # output (k < L) and un-drops the gradient of its input (k > 1)
            x, g = ops.gated_layer(x, gc.weight, gc.bias, wk.weight, wk.bias, graph,
                                   dropout_out=p if k < L else 0.0, dropout_in=p if k > 1 else 0.0,
                                   rng_state=rng, layer_id=k, grad_sink=self._sink(gc.weight, gc.bias, wk.weight, wk.bias),
                                   h_cache=h1_cache if k == 1 else None,
                                   zero_stat=zero_stat if k == last else None)   # (the next layer's statistics totals)
            gates.append(g)
        return x, gates

    def _head(self, x):
        """relu -> BatchNorm1d over the node axis -> dropout -> Linear (ChromeModels.py:48-51).
        x: [S, n, d]; strands go through BatchNorm one after the other, as the reference's two
        forward calls do (running statistics are updated forward strand first)."""
        bn, out = self.batch_norm, self.out
        if (not self.training and not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in
                (bn.weight, bn.bias, out.weight, out.bias)))) and bn.track_running_stats and bn.affine
                and x.shape[-1] in (128, 256) and out.weight.shape[0] <= 256 and out.bias is not None):
            return ops.head_logits(x, bn, out)   # eval, nothing to differentiate: one fused kernel per strand
        x = F.relu(x)
        x = torch.stack([self.batch_norm(x[s]) for s in range(x.shape[0])], 0)
        x = F.dropout(x, self.dropout, training=self.training)
        return self.out(x)

    def forward(self, x_in, adj, deg=None, src_dict=None, return_gate=False):
        ops._require_cuda(x_in, "x_in")
        # adj=None: the reference's GraphConvolution then skips the aggregation (SubLayers.py:45-48), i.e. A = I
        graph = as_graph(adj, x_in.device, n=x_in.shape[0])
        x, gates = self._gated_stack(x_in.unsqueeze(0), graph, self._step_rng())
        out = self._head(x).squeeze(0)
        gs = [g.view(-1, 1) for g in gates]  # [n,1] like sigmoid(Linear(d,1)(z))
        if len(gs) > 2:
            return x_in, out, tuple(gs), None
        return x_in, out, (gs[0], gs[1] if len(gs) > 1 else None), None  # ChromeModels.py:52

    def forward_strands(self, x_fr, adj):
        """Both strands of finetune.py:41-42 in one pass over the graph.
        x_fr: [2, n, d] (forward strand, reverse-complement strand).  Returns (logits [2,n,C], gates)."""
        ops._require_cuda(x_fr, "x_fr")
        graph = as_graph(adj, x_fr.device, n=x_fr.shape[1])
        x, gates = self._gated_stack(x_fr, graph, self._step_rng())
        return self._head(x), gates

    def forward_loss(self, x_fr, adj, target, h1_cache=None, out_slots=None, stat_acc=False):
        """The whole per-chromosome forward of the GCN stage (finetune.py:41-45,52) in fused kernels:
        gated stack on both strands, then ReLU/BatchNorm/dropout/Linear/strand-mean/BCE.  The last gated
        layer and the head form one autograd node (ops.LastLayerHeadLossFn).
        h1_cache: optional dict holder for H1 = A X of the FIRST layer.  H1 depends on the graph and the input
        features only, not on any weight, so a caller whose features are fixed (GCNStage: they never change
        across epochs) computes it on the first call and streams it afterwards instead of repeating the gather.
        out_slots: optional {'probs': [n,C], 'loss': [1]} views the head writes its results into (ops._out_slots).
        stat_acc: the caller has bounded the features (ops.last_layer_head_loss): BatchNorm sums as fixed-point totals.
        Returns (loss [], probs [n,C] = sigmoid(pred), gates)."""
        ops._require_cuda(x_fr, "x_fr")
        graph = as_graph(adj, x_fr.device, n=x_fr.shape[1])
        rng = self._step_rng()
        L = self.n_layers
        # accumulate mode with a layer in front of the last one: that layer's first launch zeroes the totals, so the last
        # layer needs no aggregation launch of its own to do it and keeps the one-launch forward on small tables
        stat_buf = ops.stat_buffer(x_fr) if (stat_acc and self.training and L > 1 and torch.is_grad_enabled()) else None
        x, gates = self._gated_stack(x_fr, graph, rng, upto=L - 1, h1_cache=h1_cache, zero_stat=stat_buf)
        p = float(self.dropout) if (self.training and rng is not None) else 0.0
        gc, wk, bn, out = getattr(self, "GC%d" % L), getattr(self, "W%d" % L), self.batch_norm, self.out
        loss, probs, g = ops.last_layer_head_loss(
            x, gc, wk, bn, out, graph, target, self.training, p, p if L > 1 else 0.0, rng, L,
            layer_sink=self._sink(gc.weight, gc.bias, wk.weight, wk.bias),
            head_sink=self._sink(bn.weight, bn.bias, out.weight, out.bias), h_cache=h1_cache if L == 1 else None,
            out_slots=out_slots, stat_acc=stat_acc, stat_buf=stat_buf)
        return loss, probs, gates + [g]
