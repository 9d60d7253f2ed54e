"""Encoder -> GCN-stage feature hand-off against golden G6 (the reference's save_feats, utils/util_methods.py:183-199)."""
import os

import numpy as np
import pytest
import torch

from chromegcn_amd.handoff import FeatureCollector

G6 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g6_save_feats.npz"))


def fill(col, batch):
    chrom = [str(c) for c in G6["chrom_of_row"]]
    x_f, x_r, t = (torch.from_numpy(G6[k]) for k in ("x_f", "x_r", "targs"))
    for i in range(0, len(chrom), batch):
        loc = [(c, 1000 * j, 1000 * j + 1000) for j, c in enumerate(chrom[i:i + batch], start=i)]
        col.add(loc, x_f[i:i + batch], x_r[i:i + batch], t[i:i + batch])


@pytest.mark.parametrize("batch", [1, 13, 64, 1000])
def test_collector_regroups_like_save_feats(batch):
    col = FeatureCollector()
    fill(col, batch)
    feats = col.finish()
    assert list(feats) == [str(c) for c in G6["order"]]
    for ch, d in feats.items():
        for k in ("forward", "backward", "target"):
            np.testing.assert_array_equal(d[k].numpy(), G6["%s_%s" % (ch, k)])


def test_collector_writes_the_reference_artefact(tmp_path):
    col = FeatureCollector()
    fill(col, 32)
    path = col.save(str(tmp_path / "run.finetune.gcn"), "valid")
    assert path == str(tmp_path / "run" / "chrom_feature_dict_valid.pt")
    saved = torch.load(path)
    assert list(saved) == [str(c) for c in G6["order"]]
    np.testing.assert_array_equal(saved["chr21"]["backward"].numpy(), G6["chr21_backward"])


def test_collector_rejects_ragged_batches():
    col = FeatureCollector()
    with pytest.raises(ValueError):
        col.add([("chr1", 0, 1)], torch.zeros(2, 4), torch.zeros(2, 4), torch.zeros(2, 3))
    assert col.finish() == {}
