"""runner.run_model: the reference-shaped epoch driver (runner.py:25-62) -- artefacts and checkpoint format."""
import os
import types

import numpy as np
import pytest
import torch

import chromegcn_amd as C
from chromegcn_amd import runner, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_run_model_writes_reference_style_artifacts(tmp_path):
    data = {"train": {}, "valid": {}, "test": {}}
    graphs = {"train": {}, "valid": {}, "test": {}}
    for i, (c, sp) in enumerate([("chr2", "train"), ("chr4", "train"), ("chr3", "valid"), ("chr1", "test")]):
        n = 300 + 40 * i
        data[sp][c] = synth.chrom_features(n, 128, 9, i, positive_rate=0.3)
        graphs[sp][c] = synth.contact_graph(n, 6 * n, i)
    torch.manual_seed(0)
    model = C.ChromeGCN(128, 128, 9, 0.2, True, 2).to(DEV)
    optim = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    sched = torch.optim.lr_scheduler.StepLR(optim, step_size=2, gamma=0.5)   # main.py:86
    out = str(tmp_path / "run")
    opt = types.SimpleNamespace(epochs=3, adj_type="hic", model_name=out, lr_decay2=1, load_gcn=False, test_only=False)
    hist = runner.run_model(None, model, data["train"], data["valid"], data["test"], None, optim, sched, opt, None,
                            graphs=graphs, verbose=False)
    assert len(hist) == 3 and set(hist[0]) == {"train", "valid", "test"}
    for f in ("train.log", "valid.log", "test.log"):
        lines = open(os.path.join(out, f)).read().strip().split("\n")
        assert len(lines) == 3 and len(lines[0].split(",")) == 6 and lines[2].startswith("3,")
    ck = torch.load(os.path.join(out, "model.chkpt"), weights_only=False)
    assert set(ck) == {"model", "settings", "epoch"} and 1 <= ck["epoch"] <= 3
    ref_keys = ["GC1.weight", "GC1.bias", "W1.weight", "W1.bias", "GC2.weight", "GC2.bias", "W2.weight", "W2.bias",
                "batch_norm.weight", "batch_norm.bias", "batch_norm.running_mean", "batch_norm.running_var",
                "batch_norm.num_batches_tracked", "out.weight", "out.bias"]
    assert list(ck["model"].keys()) == ref_keys                      # what main.py:66-69 (-load_gcn) expects
    m2 = C.ChromeGCN(128, 128, 9, 0.2, True, 2)
    m2.load_state_dict(ck["model"])
    # the learning-rate change of the scheduler reached the captured optimizer kernel (graphs re-captured)
    assert optim.param_groups[0]["lr"] < 0.25
    assert np.isfinite(hist[-1]["valid"]["meanAUC"]) and hist[-1]["train"]["loss"] < hist[0]["train"]["loss"]


def test_cli_trains_from_reference_format_files(tmp_path):
    """chromegcn_amd.train reads chrom_feature_dict_{split}.pt (utils/util_methods.py:183-199) and the graph pickles
    (data/7create_graph_new.py:197-202), optionally takes the head from a window-model checkpoint (main.py:74-81)."""
    import pickle
    from chromegcn_amd import train as T
    feat_dir, graph_root = tmp_path / "cnn_run", tmp_path / "graphs"
    feat_dir.mkdir(); graph_root.mkdir()
    names = {"train": ["chr2", "chr4"], "valid": ["chr3"], "test": ["chr1"]}
    for sp, chroms in names.items():
        feats, graphs = {}, {}
        for i, c in enumerate(chroms):
            n = 200 + 30 * i
            feats[c] = synth.chrom_features(n, 128, 6, hash(c) % 100, positive_rate=0.3)
            graphs[c] = synth.contact_graph(n, 5 * n, i)
        torch.save(feats, str(feat_dir / ("chrom_feature_dict_%s.pt" % sp)))
        with open(str(graph_root / ("%s_graphs_500000_SQRTVCnorm.pkl" % sp)), "wb") as f:
            pickle.dump(graphs, f)
    cnn = {"model": {"module.model.classifier.weight": torch.randn(6, 128), "module.model.classifier.bias": torch.randn(6),
                     "module.model.batch_norm.weight": torch.rand(128) + 0.5, "module.model.batch_norm.bias": torch.randn(128)}}
    torch.save(cnn, str(tmp_path / "cnn.chkpt"))
    out = str(tmp_path / "gcn_run")
    hist = T.main(["-feat_dir", str(feat_dir), "-graph_root", str(graph_root), "-epochs", "2", "-model_name", out,
                   "-cnn_chkpt", str(tmp_path / "cnn.chkpt"), "-gate"])
    assert len(hist) == 2 and os.path.exists(os.path.join(out, "model.chkpt"))
    hist2 = T.main(["-feat_dir", str(feat_dir), "-graph_root", str(graph_root), "-epochs", "1", "-model_name", out + "_eval",
                    "-load_gcn", os.path.join(out, "model.chkpt")])
    assert set(hist2[0]) == {"test"}
