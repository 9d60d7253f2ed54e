"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.

CPU restatement of the ChromeGCN gated-GCN hot path (the path named by
BASELINE.json's north_star / SURVEY.md section 8).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file,
and only as the checker / the timed CPU baseline.  chromegcn_amd/ never does.

Parity status: PINNED.  The reference ships no tests and no golden vectors
(SURVEY.md section 4), so the pin is tests/golden/*.npz, produced in the authoring
container by tests/golden/make_golden.py, which imports the real reference
(/root/reference) and records its inputs/outputs.  tests/test_oracle_golden.py
checks every function below against those vectors.

Every function cites the reference file:line it restates (paths relative to the
reference repository root).

Two flavours are provided on purpose:
  * torch-CPU ops in the reference's op order (GatedGCNOracle) -- autograd gives
    the backward, and this is what the cpu_baseline times (same MKL kernels the
    reference itself would hit);
  * explicit numpy math for one gated layer, forward AND hand-derived backward
    (layer_forward_np / layer_backward_np) -- SURVEY.md Appendix A -- which is the
    form the HIP kernels implement, so kernel tests can compare intermediates.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn
import torch.nn.functional as F

BAND_RADIUS = 7  # utils/util_methods.py:147  (constant_range = 7)


# --------------------------------------------------------------------------- #
# Graph normalisation  (utils/util_methods.py:99-180)
# --------------------------------------------------------------------------- #
def band_graph(n: int, radius: int = BAND_RADIUS) -> sp.csr_matrix:
    """+-radius off-diagonal band of ones, zero diagonal.
    Restates create_constant_graph, utils/util_methods.py:137-144."""
    offs = [k for k in range(-radius, radius + 1) if k != 0 and abs(k) < n]
    if not offs:
        return sp.csr_matrix((n, n), dtype=np.float64)
    diags = [np.ones(n - abs(k)) for k in offs]
    return sp.diags(diags, offs, shape=(n, n), format="csr", dtype=np.float64)


def unnormalized_adjacency(adj_type: str, hic: Optional[sp.spmatrix], n: int) -> sp.csr_matrix:
    """A-hat before row normalisation, float64 CSR with sorted indices.
    Restates the four branches of process_graph, utils/util_methods.py:148-174:
      'constant': band + I                       (:148-150)
      'hic'     : binarise(hic + I)              (:152-165)
      'both'    : hic + band + I, NOT binarised  (:168-171)
      'none'    : I                              (:173-174)
    """
    eye = sp.identity(n, dtype=np.float64, format="csr")
    if adj_type == "constant":
        a = band_graph(n) + eye
    elif adj_type == "hic":
        a = sp.csr_matrix(hic, dtype=np.float64) + eye
        a = a.tocsr()
        # :164-165  split_adj[split_adj > 0] = 1 ; split_adj[split_adj < 0] = 0
        a.data = np.where(a.data > 0, 1.0, 0.0)
    elif adj_type == "both":
        a = sp.csr_matrix(hic, dtype=np.float64) + band_graph(n) + eye
    elif adj_type == "none":
        a = eye
    else:
        # the reference leaves split_adj unbound here (UnboundLocalError)
        raise ValueError("unsupported adj_type %r" % (adj_type,))
    a = sp.csr_matrix(a)
    a.sum_duplicates()
    a.sort_indices()
    return a


def row_normalize(a: sp.spmatrix) -> sp.csr_matrix:
    """D^-1 A in float64 with inf -> 0.  Restates normalize, utils/util_methods.py:99-106."""
    a = sp.csr_matrix(a, dtype=np.float64)
    rowsum = np.asarray(a.sum(1)).astype(float).ravel()
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1.0)
    r_inv[np.isinf(r_inv)] = 0.0
    return sp.csr_matrix(sp.diags(r_inv).dot(a))


def normalized_adjacency(adj_type: str, hic: Optional[sp.spmatrix], n: int) -> sp.csr_matrix:
    """process_graph up to (but excluding) the torch conversion: float32 CSR,
    explicit zeros kept exactly where the reference keeps them.
    utils/util_methods.py:146-178 (+ the float32 cast at :122)."""
    a = row_normalize(unnormalized_adjacency(adj_type, hic, n))
    a = a.astype(np.float32)
    a.sort_indices()
    return a


def to_torch_coo(a: sp.spmatrix) -> torch.Tensor:
    """scipy -> torch sparse COO, int64 indices / float32 values.
    Restates sparse_mx_to_torch_sparse_tensor, utils/util_methods.py:120-135."""
    coo = sp.coo_matrix(a).astype(np.float32)
    idx = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
    val = torch.from_numpy(coo.data)
    return torch.sparse_coo_tensor(idx, val, torch.Size(coo.shape))


def process_graph(adj_type: str, split_adj_dict: Optional[Dict[str, sp.spmatrix]], x_size: int, chrom: str) -> torch.Tensor:
    """Same signature/return as the reference's process_graph (utils/util_methods.py:146)."""
    hic = None if split_adj_dict is None else split_adj_dict.get(chrom)
    return to_torch_coo(normalized_adjacency(adj_type, hic, x_size))


# --------------------------------------------------------------------------- #
# Model, torch-CPU flavour (models/SubLayers.py:42-52, models/ChromeModels.py:21-52)
# --------------------------------------------------------------------------- #
class GraphConvOracle(nn.Module):
    """A (X W) + b.  models/SubLayers.py:8-24 (ctor), :32-35 (init), :42-52 (forward)."""

    def __init__(self, d_in: int, d_out: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d_in, d_out))
        self.bias = nn.Parameter(torch.zeros(d_out))
        nn.init.xavier_normal_(self.weight, gain=0.02)  # :33

    def forward(self, x, adj):
        s = x @ self.weight  # :43
        u = torch.sparse.mm(adj, s) if adj is not None else s  # :45-48
        return u + self.bias  # :50


class GatedGCNOracle(nn.Module):
    """ChromeGCN restated (models/ChromeModels.py:21-52), generalised to n_layers >= 1
    by repeating lines :42-46.  The reference itself builds a second layer only when
    layers == 2 (:25) -- pass reference_layer_rule=True to reproduce that quirk.
    state_dict keys match the reference (GC1, W1, GC2, W2, batch_norm, out)."""

    def __init__(self, d: int, n_class: int, dropout: float, n_layers: int, reference_layer_rule: bool = False):
        super().__init__()
        if reference_layer_rule:
            n_layers = 2 if n_layers == 2 else 1
        self.n_layers = n_layers
        self.dropout = dropout
        for k in range(1, n_layers + 1):
            setattr(self, "GC%d" % k, GraphConvOracle(d, d))
            setattr(self, "W%d" % k, nn.Linear(d, 1))
        self.batch_norm = nn.BatchNorm1d(d)
        self.out = nn.Linear(d, n_class)

    def forward(self, x_in, adj, deg=None, src_dict=None, return_gate=False):
        x = x_in
        gates: List[torch.Tensor] = []
        for k in range(1, self.n_layers + 1):
            if k > 1:
                x = F.dropout(x, self.dropout, training=self.training)  # :42
            z = torch.tanh(getattr(self, "GC%d" % k)(x, adj))  # :37-38 / :43-44
            g = torch.sigmoid(getattr(self, "W%d" % k)(z))  # :39 / :45
            x = (1 - g) * x + g * z  # :40 / :46
            gates.append(g)
        x = F.relu(x)  # :48
        x = self.batch_norm(x)  # :49
        x = F.dropout(x, self.dropout, training=self.training)  # :50
        out = self.out(x)  # :51
        g1 = gates[0]
        g2 = gates[1] if len(gates) > 1 else None
        if len(gates) > 2:
            return x_in, out, tuple(gates), None
        return x_in, out, (g1, g2), None  # :52


# --------------------------------------------------------------------------- #
# One gated layer, explicit numpy math
# --------------------------------------------------------------------------- #
def _csr_parts(a: sp.csr_matrix):
    a = sp.csr_matrix(a)
    return a.indptr.astype(np.int64), a.indices.astype(np.int64), a.data


def layer_forward_np(a_norm: sp.csr_matrix, x: np.ndarray, w: np.ndarray, b: np.ndarray,
                     wg: np.ndarray, cg: float, dtype=np.float64):
    """Forward of one gated layer: S = X W, U = A S + b, Z = tanh U, g = sigma(Z wg + cg),
    X' = (1-g) X + g Z.  models/SubLayers.py:43-50, models/ChromeModels.py:37-40."""
    a = sp.csr_matrix(a_norm).astype(dtype)
    x = x.astype(dtype); w = w.astype(dtype); b = b.astype(dtype); wg = wg.astype(dtype).reshape(-1)
    s = x @ w
    u = a @ s + b
    z = np.tanh(u)
    logit = z @ wg + dtype(cg)
    g = 1.0 / (1.0 + np.exp(-logit))
    xn = (1.0 - g)[:, None] * x + g[:, None] * z
    return {"S": s, "U": u, "Z": z, "g": g, "Xn": xn}


def layer_backward_np(a_norm: sp.csr_matrix, x: np.ndarray, w: np.ndarray, wg: np.ndarray,
                      z: np.ndarray, g: np.ndarray, grad_xn: np.ndarray, grad_g: Optional[np.ndarray] = None,
                      dtype=np.float64):
    """Hand-derived backward of one gated layer.  grad_g is an
    optional upstream gradient on the gate output itself (return_gate consumers)."""
    a = sp.csr_matrix(a_norm).astype(dtype)
    x = x.astype(dtype); w = w.astype(dtype); wg = wg.astype(dtype).reshape(-1)
    z = z.astype(dtype); g = g.astype(dtype); G = grad_xn.astype(dtype)
    dg = np.sum(G * (z - x), axis=1)
    if grad_g is not None:
        dg = dg + grad_g.astype(dtype).reshape(-1)
    gamma = g * (1.0 - g) * dg
    dz = g[:, None] * G + gamma[:, None] * wg[None, :]
    du = dz * (1.0 - z * z)
    db = du.sum(0)
    dwg = (gamma[:, None] * z).sum(0)
    dcg = gamma.sum()
    ds = a.T @ du
    dw = x.T @ ds
    dx = (1.0 - g)[:, None] * G + ds @ w.T
    return {"gamma": gamma, "dU": du, "dS": ds, "dW": dw, "db": db, "dwg": dwg, "dcg": dcg, "dX": dx}


# --------------------------------------------------------------------------- #
# Stage loop (finetune.py:9-67) and its timer (runner.py:10-23)
# --------------------------------------------------------------------------- #
def finetune_epoch(model: nn.Module, chrom_feature_dict, split_adj_dict, optimizer, split: str,
                   adj_type: str = "hic", adj_cache: Optional[dict] = None, input_grads: Optional[dict] = None):
    """CPU restatement of the reference GCN-stage loop body, finetune.py:29-53, minus the
    hard-coded .cuda() calls (:30-36) that make the original unrunnable without a GPU.
    Returns (all_preds, all_targets, total_loss) exactly like finetune.py:67.
    adj_cache (optional dict) lets a caller hoist process_graph out of the timed loop; the
    reference recomputes it every chromosome every epoch (finetune.py:36).
    input_grads (optional dict) receives {chrom: (x_f.grad, x_r.grad)}: finetune.py:33-34 makes the features
    require grad, but the loop-local tensors die with the iteration; tests need them to check d loss / d features."""
    model.train() if split == "train" else model.eval()  # :10-13
    all_preds = torch.Tensor()
    all_targets = torch.Tensor()
    total_loss = 0.0
    for chrom in chrom_feature_dict:  # :29
        x_f = chrom_feature_dict[chrom]["forward"].clone().requires_grad_(True)  # :30,33
        x_r = chrom_feature_dict[chrom]["backward"].clone().requires_grad_(True)  # :31,34
        targets = chrom_feature_dict[chrom]["target"]  # :32
        if adj_cache is not None and chrom in adj_cache:
            adj = adj_cache[chrom]
        else:
            adj = process_graph(adj_type, split_adj_dict, x_f.size(0), chrom)  # :36
            if adj_cache is not None:
                adj_cache[chrom] = adj
        if split == "train":
            optimizer.zero_grad()  # :39
        _, pred_f, _, _ = model(x_f, adj, None)  # :41
        _, pred_r, _, _ = model(x_r, adj, None)  # :42
        pred = (pred_f + pred_r) / 2  # :43
        loss = F.binary_cross_entropy_with_logits(pred, targets.float())  # :45
        if split == "train":
            loss.backward()  # :48
            optimizer.step()  # :49
            if input_grads is not None:
                input_grads[chrom] = (x_f.grad.detach().clone(), x_r.grad.detach().clone())
        total_loss += loss.sum().item()  # :51
        all_preds = torch.cat((all_preds, torch.sigmoid(pred).detach()), 0)  # :52
        all_targets = torch.cat((all_targets, targets.detach().float()), 0)  # :53
    return all_preds, all_targets, total_loss


def make_sgd(model: nn.Module, lr: float):
    """get_optimizer, 'sgd' branch: utils/util_methods.py:14-19."""
    return torch.optim.SGD(model.parameters(), lr=lr, weight_decay=1e-6, momentum=0.9)


# --------------------------------------------------------------------------- #
# Seeded synthetic inputs shared by tests and bench
# --------------------------------------------------------------------------- #
def random_symmetric_graph(n: int, pairs: int, seed: int, hic_like: bool = False) -> sp.csr_matrix:
    """Zero-diagonal symmetric {0,1} float64 CSR -- the on-disk graph contract
    (data/7create_graph_new.py:108-120).  hic_like draws |i-j| from a truncated 1/k law."""
    rng = np.random.RandomState(seed)
    if hic_like:
        kmax = max(2, n - 1)
        u = rng.random_sample(pairs)
        dist = np.clip(np.floor(np.exp(u * math.log(kmax))).astype(np.int64), 1, n - 1)
        i = (rng.random_sample(pairs) * (n - dist)).astype(np.int64)
        j = i + dist
    else:
        i = rng.randint(0, n, pairs)
        j = rng.randint(0, n, pairs)
    keep = i != j
    i, j = i[keep], j[keep]
    a = sp.coo_matrix((np.ones(i.size), (i, j)), shape=(n, n)).tocsr()
    a = a + a.T
    a.data[:] = 1.0
    a.sort_indices()
    return sp.csr_matrix(a, dtype=np.float64)


# --------------------------------------------------------------------------- #
# Multi-label metrics (utils/metrics.py:25-26,148-183,238-253; utils/evals.py:89-92)
# --------------------------------------------------------------------------- #
def multilabel_metrics_np(targets: np.ndarray, preds: np.ndarray, fdr_cutoff: float = 0.5):
    """Per-label arrays exactly as the reference's helpers compute them with scikit-learn (its dependency):
    roc_auc_score (metrics.py:243), auc(recall, precision) of precision_recall_curve (:171-173), recall at the
    first index with 1 - precision <= cutoff (:152-156), average_precision_score (:25-26).  NaN where sklearn
    raises or returns NaN (the reference skips those labels)."""
    import warnings
    from sklearn import metrics as skm
    C = targets.shape[1]
    out = {k: np.full(C, np.nan) for k in ("auroc", "aupr", "recall_at_fdr", "average_precision")}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for c in range(C):
            try:
                out["auroc"][c] = skm.roc_auc_score(targets[:, c], preds[:, c])
            except ValueError:
                pass
            try:
                precision, recall, _ = skm.precision_recall_curve(targets[:, c], preds[:, c], pos_label=1)
                out["aupr"][c] = skm.auc(recall, precision)
                idx = next(i for i, x in enumerate(1 - precision) if x <= fdr_cutoff)
                out["recall_at_fdr"][c] = recall[idx]
                out["average_precision"][c] = skm.average_precision_score(targets[:, c], preds[:, c], pos_label=1)
            except Exception:
                pass
    return out
