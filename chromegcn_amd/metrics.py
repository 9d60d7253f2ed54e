"""Multi-label metrics on the GPU: the build's counterpart of the reference's
compute_metrics (utils/evals.py:26-120) / utils/metrics.py, which calls scikit-learn once per label per metric on
the CPU after every split (runner.py:41,45,51) -- the dominant cost of an epoch once the GCN step takes
microseconds.  One device sort + one scan (cgcn_multilabel_metrics)."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import _lib


def _metrics_raw(probs: torch.Tensor, targets: torch.Tensor, fdr_cutoff: float, nonneg: bool) -> torch.Tensor:
    """one launch sequence; returns a flat device tensor: [4 * C] results, then (nonneg path) the `bad` word as element 4 C
    (bit pattern of an int32: nonzero = a score was negative or NaN and the results are to be discarded)"""
    if not probs.is_cuda or not targets.is_cuda:
        raise RuntimeError("chromegcn_amd.metrics: tensors must be on the GPU (there is no CPU fallback; "
                           "the reference's sklearn path is utils/metrics.py)")
    probs = probs.contiguous().float()
    targets = targets.contiguous().float()
    n, C = probs.shape
    if tuple(targets.shape) != (n, C):
        raise RuntimeError("probs and targets must both be [n, C]")
    lib = _lib.load()
    ws_bytes = lib.cgcn_metrics_workspace_bytes(n, C)
    if ws_bytes == 0:
        raise RuntimeError("chromegcn_amd.metrics: unsupported size n=%d C=%d" % (n, C))
    ws = torch.empty(ws_bytes, device=probs.device, dtype=torch.uint8)
    out = torch.empty(4 * C + 1, device=probs.device, dtype=torch.float32)
    if nonneg:
        _lib.check(lib.cgcn_multilabel_metrics_nonneg(_lib.stream_ptr(), n, C, probs.data_ptr(), targets.data_ptr(), float(fdr_cutoff),
                                                      out.data_ptr(), out.data_ptr() + 16 * C, ws.data_ptr(), ws_bytes),
                   "cgcn_multilabel_metrics_nonneg")
    else:
        _lib.check(lib.cgcn_multilabel_metrics(_lib.stream_ptr(), n, C, probs.data_ptr(), targets.data_ptr(), float(fdr_cutoff),
                                               out.data_ptr(), ws.data_ptr(), ws_bytes), "cgcn_multilabel_metrics")
    return out


def _split(flat: torch.Tensor, C: int) -> Dict[str, torch.Tensor]:
    out = flat[:4 * C].view(4, C)
    return {"auroc": out[0], "aupr": out[1], "recall_at_fdr": out[2], "average_precision": out[3]}


def multilabel_metrics(probs: torch.Tensor, targets: torch.Tensor, fdr_cutoff: float = 0.5) -> Dict[str, torch.Tensor]:
    """probs, targets: [n, C] float32 CUDA tensors.  Returns per-label tensors [C] (NaN where undefined):
    'auroc', 'aupr', 'recall_at_fdr', 'average_precision'.
    Scores are taken for probabilities first (non-negative: 32-bit keys and the library's own segmented radix sort,
    cgcn_multilabel_metrics_nonneg); if the device reports a negative score or a NaN the general path (any float scores,
    64-bit keys) runs instead -- same results either way where both apply."""
    C = probs.shape[1]
    flat = _metrics_raw(probs, targets, fdr_cutoff, nonneg=True)
    if int(flat[4 * C:].view(torch.int32).item()) != 0:
        flat = _metrics_raw(probs, targets, fdr_cutoff, nonneg=False)
    return _split(flat, C)


def compute_metrics(all_predictions, all_targets, loss, args=None, elapsed=0.0, data_dict=None, cell_type=None,
                    device="cuda", verbose=False):
    """Same positional arguments and result keys as the reference's compute_metrics (utils/evals.py:26,107-120;
    the per_label_type / plot branches are analysis-only and not reproduced).  Labels whose metric is undefined
    (a single class present) are skipped in the means, as the reference's try/except does.  Unlike the reference
    this does NOT threshold all_predictions in place (utils/evals.py:99-100)."""
    p = torch.as_tensor(all_predictions).to(device=device, dtype=torch.float32)
    t = torch.as_tensor(all_targets).to(device=device, dtype=torch.float32)
    C = p.shape[1]
    host = _metrics_raw(p, t, 0.5, nonneg=True).cpu()          # ONE device-to-host copy: results + the `bad` word
    if int(host[4 * C:].view(torch.int32).item()) != 0:        # scores that are not probabilities: the general path
        host = _metrics_raw(p, t, 0.5, nonneg=False).cpu()
    m = {k: v.double().numpy() for k, v in _split(host, C).items()}
    auc = m["auroc"][~np.isnan(m["auroc"])]
    aupr = m["aupr"][~np.isnan(m["aupr"])]
    fdr = m["recall_at_fdr"][~np.isnan(m["recall_at_fdr"])]
    out = {
        "mAP": float(np.mean(m["average_precision"])) if m["average_precision"].size else float("nan"),
        "meanAUC": float(np.mean(auc)) if auc.size else float("nan"),
        "medianAUC": float(np.median(auc)) if auc.size else float("nan"),
        "allAUC": auc, "allFDR": fdr,
        "meanAUPR": float(np.mean(aupr)) if aupr.size else float("nan"),
        "medianAUPR": float(np.median(aupr)) if aupr.size else float("nan"),
        "allAUPR": aupr,
        "meanFDR": float(np.mean(fdr)) if fdr.size else float("nan"),
        "medianFDR": float(np.median(fdr)) if fdr.size else float("nan"),
        "loss": loss, "time": elapsed,
    }
    if verbose:  # utils/evals.py:102-105
        print("mAP:      " + str(round(out["mAP"], 3)))
        print("meanAUC:  " + str(round(out["meanAUC"], 3)))
        print("meanAUPR: " + str(round(out["meanAUPR"], 3)))
        print("meanFDR:  " + str(round(out["meanFDR"], 3)))
    return out
