"""Config 5 on the GPU: encoder -> device hand-off -> GCN stage.  The device hand-off must equal the reference's
`.pt` round trip (utils/util_methods.py:183-199 writes, main.py:30-32 reads) bit for bit, and the stage fed either
way must produce identical training results."""
import os

import numpy as np
import pytest
import torch

import chromegcn_amd as C
from chromegcn_amd import e2e, synth
from chromegcn_amd.encoder import StrandPair, WindowEncoder, extract_features
from chromegcn_amd.finetune import GCNStage
from chromegcn_amd.handoff import FeatureCollector

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_device_handoff_equals_pt_round_trip_bitwise(tmp_path):
    L, n_labels, windows = 320, 9, 96
    chroms = ("chr20", "chr22")
    tokens, targets, locs = e2e.synthetic_windows(chroms, windows, L, n_labels, seed=3)
    # interleave the two chromosomes in file order to exercise the regrouping (save_feats keeps order of appearance)
    perm = torch.stack([torch.arange(windows), torch.arange(windows) + windows], 1).reshape(-1)
    tokens, targets, locs = tokens[perm], targets[perm], [locs[i] for i in perm.tolist()]
    torch.manual_seed(1)
    enc = StrandPair(WindowEncoder(n_labels, L)).to(DEV)
    col = extract_features(enc, tokens.to(DEV), targets.to(DEV), locs, FeatureCollector(), batch_size=32)
    feats_dev = col.finish()
    assert list(feats_dev) == list(chroms) and all(v["forward"].is_cuda for v in feats_dev.values())
    # the reference's artefact: written, then read back by "the next run"
    path = col.save(str(tmp_path / "run.finetune.x"), "train")
    assert os.path.basename(path) == "chrom_feature_dict_train.pt"
    feats_pt = torch.load(path)
    for c in chroms:
        for k in ("forward", "backward", "target"):
            assert feats_pt[c][k].device.type == "cpu"
            assert torch.equal(feats_pt[c][k], feats_dev[c][k].cpu()), (c, k)
    # and a stage trained from either source ends up with identical parameters and predictions
    graphs = {c: synth.contact_graph(windows, 400, synth.chrom_seed(c)) for c in chroms}
    outs = []
    for feats in (feats_dev, feats_pt):
        torch.manual_seed(0)
        m = C.ChromeGCN(128, 128, n_labels, 0.0, True, 2).to(DEV)
        with torch.no_grad():
            m.GC1.weight.mul_(40); m.GC2.weight.mul_(40)
        opt = torch.optim.SGD(m.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
        st = GCNStage(m, opt, "hic", DEV, hip_graphs=True)
        st.load(feats, graphs)
        for _ in range(2):
            preds, tg, loss = st.run_split("train")
        outs.append((preds, loss, {k: v.clone() for k, v in m.state_dict().items()}))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    for k in outs[0][2]:
        assert torch.equal(outs[0][2][k], outs[1][2][k]), k


def test_encoder_on_gpu_matches_cpu():
    """stock torch-ROCm ops: the same module on both devices (MIOpen convolutions vs the CPU's), 1e-4"""
    L, n_labels = 600, 7
    torch.manual_seed(2)
    enc = StrandPair(WindowEncoder(n_labels, L)).eval()
    tok = torch.randint(0, 5, (6, L))
    with torch.no_grad():
        cpu = enc(tok)
        gpu = enc.to(DEV)(tok.to(DEV))
    for a, b in zip(cpu[:3], gpu[:3]):
        np.testing.assert_allclose(b.cpu().numpy(), a.numpy(), atol=1e-4, rtol=1e-4)


def test_e2e_pipeline_runs_and_reports_both_rates():
    t, stage, names = e2e.run_pipeline(torch.device(DEV), windows=256, seq_length=320, epochs=2, warmup=1,
                                       chroms=("chr20", "chr22"), batch_size=64)
    assert t["windows"] == 512 and t["feat_device"].startswith("cuda")
    assert t["encoder_s"] > 0 and t["gcn_epoch_s"] > 0 and np.isfinite(t["final_loss"])
    assert names == ["chr20", "chr22"] and all(stage.chroms[c].n == 256 for c in names)
