"""Adjacency saliency on the sparsity pattern.

The reference computes it by making the adjacency a DENSE n x n tensor with requires_grad on the CPU
(scripts/visualize.py:29-49): adj_grad = |adj * d sum(sigmoid(pred) * targets) / d adj|, then a row-sum and a
row-max normalisation.  Entries outside the pattern have adj = 0 and drop out of the product, so only
    dL/dA_ij = sum over layers and strands of < dL/dU_i , (X W)_j > = < dL/dU_i W^T , X_j >      for stored (i, j)
is needed: one SDDMM per layer (cgcn_sddmm) instead of an O(n^2) gradient (3.6 GB at n = 30k)."""
from __future__ import annotations

import torch

from . import ops
from .graph import ChromGraph, as_graph


def adjacency_saliency(model, x_f: torch.Tensor, x_r: torch.Tensor, adj, targets: torch.Tensor, normalize: bool = True):
    """Returns (graph, sal) with sal[k] the saliency of stored entry k of the graph's CSR (row-major).
    model: chromegcn_amd.ChromeGCN; x_f, x_r: [n,d]; targets: [n,C] -- the arguments of scripts/visualize.py:37-49."""
    graph: ChromGraph = as_graph(adj, x_f.device)
    if not graph.symmetric and graph.val_t is not graph.val:
        raise NotImplementedError("saliency is implemented for symmetric A-hat (every graph the reference writes)")
    x = torch.stack([x_f, x_r]).detach().requires_grad_(True)
    tap = []
    ops._saliency_tap = tap
    try:
        logits, _ = model.forward_strands(x, graph)
        pred = (logits[0] + logits[1]) / 2                       # visualize.py:39
        torch.sigmoid(pred).backward(gradient=targets)           # visualize.py:40,47
    finally:
        ops._saliency_tap = None
    # dL/dA_ij (A = diag(rs) Ahat) = <dU_i, (XW)_j>;  a_ij * dL/dA_ij = ahat_ij * <rs_i dU_i W^T, X_j> = ahat_ij * <dHs_i, X_j>
    # with dHs = diag(rs) dL/dU W^T, which the layer backward leaves behind (cgcn_layer_bwd)
    total = None
    for (xin, dhs, g) in tap:
        total = ops.sddmm(dhs, xin, graph, out=total)                # one product per layer, summed in place
    if total is None:
        total = torch.zeros(graph.col.shape[0], device=x.device)
    if normalize:                                                     # visualize.py:49-55, one launch on the pattern
        return graph, ops.saliency_normalize(total, graph)
    if graph.val is not None:
        total = total * graph.val
    return graph, total.abs()                                         # visualize.py:49  |adj * adj.grad|
