"""Device-side feature hand-off (chromegcn_amd.handoff) feeding the stage engine: same predictions as loading the
reference-format dict from the host."""
import numpy as np
import pytest
import torch

import chromegcn_amd as C
from chromegcn_amd import synth
from chromegcn_amd.finetune import GCNStage
from chromegcn_amd.handoff import FeatureCollector

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_collector_on_device_feeds_the_stage():
    feats = {"chr3": synth.chrom_features(260, 128, 11, 3, positive_rate=0.2),
             "chr12": synth.chrom_features(190, 128, 11, 12, positive_rate=0.2)}
    graphs = {"chr3": synth.contact_graph(260, 1500, 3), "chr12": synth.contact_graph(190, 900, 12)}
    # the encoder's view: batches interleave the chromosomes, each chromosome's windows stay in order
    rows = sorted([("chr3", i) for i in range(260)] + [("chr12", i) for i in range(190)], key=lambda r: r[1])
    col = FeatureCollector()
    for i in range(0, len(rows), 48):
        chunk = rows[i:i + 48]
        f = torch.stack([feats[c]["forward"][j] for c, j in chunk]).to(DEV)
        r = torch.stack([feats[c]["backward"][j] for c, j in chunk]).to(DEV)
        t = torch.stack([feats[c]["target"][j] for c, j in chunk]).to(DEV)
        col.add([(c, j * 1000, j * 1000 + 1000) for c, j in chunk], f, r, t)
    torch.manual_seed(1)
    model = C.ChromeGCN(128, 128, 11, 0.2, True, 2).to(DEV)
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    a = GCNStage(model, opt, "hic", DEV)
    order = col.to_stage(a, graphs)
    assert sorted(order) == ["chr12", "chr3"]
    assert a.chroms["chr3"].x.is_cuda and a.chroms["chr3"].n == 260
    pa, ta, la = a.run_split("valid", ["chr3", "chr12"])
    b = GCNStage(model, opt, "hic", DEV)
    b.load(feats, graphs)
    pb, tb, lb = b.run_split("valid", ["chr3", "chr12"])
    np.testing.assert_array_equal(np.asarray(pa), np.asarray(pb))
    np.testing.assert_array_equal(np.asarray(ta), np.asarray(tb))
    assert la == lb
