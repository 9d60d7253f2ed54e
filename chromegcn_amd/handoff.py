"""Feature hand-off from the window encoder to the GCN stage (first half).

The reference accumulates the encoder's per-batch outputs on the host (`torch.cat` onto a growing CPU tensor
every batch, pretrain.py:57-60 -- quadratic copying), regroups them per chromosome with `index_select`
(utils/util_methods.py:183-197), pickles the dict to `chrom_feature_dict_<split>.pt` (:199) and loads it again
in the next run (main.py:30-32).  `FeatureCollector` keeps the batches where the encoder produced them (device
memory), regroups once, and hands `{chrom: {'forward','backward','target'}}` -- the same contract -- straight
to `GCNStage`; the `.pt` file is still available through `save()` for callers that want the reference's artefact.

Plumbing only (torch tensor ops on whatever device the encoder ran on); no arithmetic happens here, so the
regrouped tensors are bit-identical to the reference's (tests/test_handoff.py, golden G6)."""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence

import torch


class FeatureCollector:
    """Drop-in for the `opt.save_feats` branch of pretrain.py:57-63.

        col = FeatureCollector()
        for batch in loader:                      # pretrain.py:24-60
            ...
            col.add(loc, x_out_f, x_out_r, tgt)   # instead of the three torch.cat / all_locs.append lines
        feats = col.finish()                      # what save_feats would have written, still on the device
        stage.load(feats, split_adj_dict)         # or col.to_stage(stage, split_adj_dict)
    """

    def __init__(self, device: Optional[torch.device] = None):
        self.device = torch.device(device) if device is not None else None
        self._f: List[torch.Tensor] = []
        self._r: List[torch.Tensor] = []
        self._t: List[torch.Tensor] = []
        self._chrom_of_row: List[str] = []

    def __len__(self):
        return len(self._chrom_of_row)

    def add(self, loc: Sequence, x_out_f: torch.Tensor, x_out_r: torch.Tensor, target: torch.Tensor):
        """loc: one entry per row, `loc_i[0]` is the chromosome name (pretrain.py:60, util_methods.py:187);
        x_out_f / x_out_r: [b, d] encoder features of the forward / reverse-complement strand; target: [b, C]."""
        b = len(loc)
        if x_out_f.shape[0] != b or x_out_r.shape[0] != b or target.shape[0] != b:
            raise ValueError("FeatureCollector.add: %d locations for batches of %d / %d / %d rows"
                             % (b, x_out_f.shape[0], x_out_r.shape[0], target.shape[0]))
        if x_out_f.shape != x_out_r.shape:
            raise ValueError("FeatureCollector.add: strand feature shapes differ")
        dev = self.device if self.device is not None else x_out_f.device
        self._f.append(x_out_f.detach().to(dev))
        self._r.append(x_out_r.detach().to(dev))
        self._t.append(target.detach().to(dev))
        self._chrom_of_row.extend(str(l[0]) if not isinstance(l, str) else l for l in loc)

    def finish(self) -> Dict[str, Dict[str, torch.Tensor]]:
        """{chrom: {'forward','backward','target'}}: chromosomes in order of first appearance, rows in order of
        appearance inside each chromosome (what utils/util_methods.py:186-197 builds)."""
        if not self._f:
            return {}
        all_f, all_r, all_t = torch.cat(self._f, 0), torch.cat(self._r, 0), torch.cat(self._t, 0)
        index: Dict[str, List[int]] = {}
        for i, ch in enumerate(self._chrom_of_row):
            index.setdefault(ch, []).append(i)
        out = {}
        for ch, rows in index.items():
            idx = torch.tensor(rows, dtype=torch.long, device=all_f.device)
            out[ch] = {"forward": all_f.index_select(0, idx), "backward": all_r.index_select(0, idx),
                       "target": all_t.index_select(0, idx)}
        return out

    def to_stage(self, stage, split_adj_dict=None, only: Optional[Iterable[str]] = None):
        """Registers every collected chromosome with a `finetune.GCNStage` (graph normalisation + upload happen
        there, once); returns the chromosome names in the reference's iteration order."""
        feats = self.finish()
        stage.load(feats, split_adj_dict, only)
        return [c for c in feats if only is None or c in only]

    def save(self, model_name: str, split: str) -> str:
        """Writes the reference's artefact: `<model_name before '.finetune'>/chrom_feature_dict_<split>.pt`
        with CPU tensors (utils/util_methods.py:199)."""
        feats = {ch: {k: v.cpu() for k, v in d.items()} for ch, d in self.finish().items()}
        path = os.path.join(model_name.split(".finetune")[0], "chrom_feature_dict_" + split + ".pt")
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        torch.save(feats, path)
        return path
