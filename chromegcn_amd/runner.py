"""Epoch driver of the GCN stage: the build's counterpart of the reference's run_epoch / run_model
(runner.py:10-62) for `-chrome_model gcn` runs.  Same call signature and the same artefacts a reference user
relies on:
  * per-split `(preds, targets, loss, elapsed_minutes)` from run_epoch (runner.py:10-23);
  * `<model_name>/{train,valid,test}.log` lines `epoch,loss,mAP,meanAUC,meanAUPR,meanFDR` (utils/evals.py:297-300);
  * `<model_name>/model.chkpt` = {'model': state_dict, 'settings': opt, 'epoch'} whenever the validation metric
    sum (meanAUPR + meanAUPR + meanFDR, runner.py:46) is a new maximum (utils/evals.py:250-263) -- loadable by the
    reference's `-load_gcn` path (main.py:66-69) because the state_dict keys are the reference's.
Metrics come from the device (chromegcn_amd.metrics) instead of per-label scikit-learn calls on the host."""
from __future__ import annotations

import os
import time
from typing import Optional

import torch

from . import metrics as M
from .finetune import GCNStage, _load_graph_file


def _stage_for(ChromeModel, optimizer, opt, split):
    stages = ChromeModel.__dict__.setdefault("_cgcn_stages", {})
    adj_type = getattr(opt, "adj_type", "hic")
    key = (split, id(optimizer), adj_type)
    if key not in stages:
        dev = next(ChromeModel.parameters()).device
        stages[key] = GCNStage(ChromeModel, optimizer, adj_type=adj_type, device=dev,
                               hip_graphs=getattr(opt, "hip_graphs", True))
    return stages[key]


def run_epoch(WindowModel, ChromeModel, split_data, crit, optimizer, epoch, data_dict, opt, split, split_adj_dict=None,
              to_cpu=False):
    """runner.py:10-23.  Returns (pred, targ, loss, elapsed_minutes); tensors stay on the device unless to_cpu."""
    start = time.time()
    adj_type = getattr(opt, "adj_type", "hic")
    if split_adj_dict is None and adj_type in ("hic", "both"):
        split_adj_dict = _load_graph_file(opt, split)
    stage = _stage_for(ChromeModel, optimizer, opt, split)
    stage.load(split_data, split_adj_dict)
    pred, targ, loss = stage.run_split(split, list(split_data), to_cpu=to_cpu)
    elapsed = (time.time() - start) / 60
    return pred, targ, loss, elapsed


class RunLog:
    """the files SaveLogger keeps (utils/evals.py:265-300), minus the per-epoch prediction dumps"""

    def __init__(self, model_name: Optional[str]):
        self.model_name = model_name
        self.best_metric = float("-inf")
        self.best_loss = float("inf")
        self.best_loss_epoch = 0
        if model_name:
            os.makedirs(model_name, exist_ok=True)
            for f in ("train.log", "valid.log", "test.log"):
                open(os.path.join(model_name, f), "w").close()

    def log(self, file_name, epoch, loss, m):
        if not self.model_name or m is None:
            return
        with open(os.path.join(self.model_name, file_name), "a") as f:
            f.write("%s,%s,%s,%s,%s,%s\n" % (epoch, loss, m["mAP"], m["meanAUC"], m["meanAUPR"], m["meanFDR"]))

    def maybe_checkpoint(self, epoch, opt, model, valid_loss, valid_metric_sum):
        if valid_loss < self.best_loss:
            self.best_loss, self.best_loss_epoch = valid_loss, epoch
        if valid_metric_sum >= self.best_metric:
            self.best_metric = valid_metric_sum
            if self.model_name:
                sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
                torch.save({"model": sd, "settings": opt, "epoch": epoch}, os.path.join(self.model_name, "model.chkpt"))
                return True
        return False


def run_model(WindowModel, ChromeModel, train_data, valid_data, test_data, crit, optimizer, scheduler, opt, data_dict,
              logger=None, graphs=None, verbose=True):
    """runner.py:25-62 for the GCN stage.  `graphs`: optional {'train'|'valid'|'test': {chrom: scipy matrix}} instead
    of the pickles under opt.graph_root.  Returns the list of per-epoch dicts {'train','valid','test'} of metrics."""
    log = RunLog(getattr(opt, "model_name", None))
    history = []
    g = graphs or {}
    for epoch in range(1, opt.epochs + 1):
        if scheduler is not None and getattr(opt, "lr_decay2", 0) > 0:
            scheduler.step()                                                                   # runner.py:33-34
        rec = {}
        if not getattr(opt, "load_gcn", False) and not getattr(opt, "test_only", False):
            p, t, loss, el = run_epoch(WindowModel, ChromeModel, train_data, crit, optimizer, epoch, data_dict, opt,
                                       "train", g.get("train"))                                # runner.py:40
            rec["train"] = M.compute_metrics(p, t, loss, opt, el)
            p, t, vloss, el = run_epoch(WindowModel, ChromeModel, valid_data, crit, optimizer, epoch, data_dict, opt,
                                        "valid", g.get("valid"))                               # runner.py:44
            rec["valid"] = M.compute_metrics(p, t, vloss, opt, el)
            vsum = rec["valid"]["meanAUPR"] + rec["valid"]["meanAUPR"] + rec["valid"]["meanFDR"]  # runner.py:46
        else:
            vloss, vsum = 0.0, 0.0
        p, t, tloss, el = run_epoch(WindowModel, ChromeModel, test_data, crit, optimizer, epoch, data_dict, opt, "test",
                                    g.get("test"))                                             # runner.py:50
        rec["test"] = M.compute_metrics(p, t, tloss, opt, el)
        saved = False
        if "valid" in rec:
            saved = log.maybe_checkpoint(epoch, opt, ChromeModel, vloss, vsum)                 # runner.py:57
        log.log("test.log", epoch, tloss, rec["test"])
        log.log("valid.log", epoch, vloss, rec.get("valid"))
        log.log("train.log", epoch, rec["train"]["loss"] if "train" in rec else 0.0, rec.get("train"))
        if verbose:
            print("epoch %d  train loss %.4f  valid meanAUC %.4f meanAUPR %.4f  test meanAUC %.4f%s" % (
                epoch, rec["train"]["loss"] if "train" in rec else float("nan"),
                rec["valid"]["meanAUC"] if "valid" in rec else float("nan"),
                rec["valid"]["meanAUPR"] if "valid" in rec else float("nan"), rec["test"]["meanAUC"],
                "  [checkpoint]" if saved else ""))
        history.append(rec)
    return history
