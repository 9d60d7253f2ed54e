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


# ------------------------------------------------------------------------------------------------------------------
# config 5 across ranks: every rank encodes the windows of the chromosomes it owns and trains them (e2e.run_pipeline
# with a process group).  Two spawned ranks share cuda:0 (gloo: RCCL refuses two ranks on one device).
# ------------------------------------------------------------------------------------------------------------------
E2E_KW = dict(windows=256, seq_length=320, dropout=0.0, epochs=2, warmup=1, chroms=("chr19", "chr20", "chr22"), batch_size=64)


def _e2e_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t, stage, names = e2e.run_pipeline(torch.device("cuda:0"), group=dist.group.WORLD, return_feats=True, **E2E_KW)
    preds, _, loss = stage.run_split("valid", names)
    q.put((rank, t["owned"], {c: {k: v.cpu().numpy() for k, v in f.items()} for c, f in t["feats"].items()},
           {k: v.cpu().numpy() for k, v in stage.model.state_dict().items()}, preds.numpy(), t["encoded_windows_this_rank"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_e2e_sharded_over_two_ranks_features_bitwise_parameters_match_emulation():
    import sys
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_two_rank import free_port
    from chromegcn_amd.dist import plan_shards
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_e2e_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    chroms = E2E_KW["chroms"]
    # every chromosome was encoded by exactly one rank, and no rank encoded windows it does not own
    owned = [set(r[1]) for r in res]
    assert owned[0] | owned[1] == set(chroms) and not (owned[0] & owned[1]) and all(owned)
    assert [r[5] for r in res] == [len(o) * E2E_KW["windows"] for o in owned]
    # ---- single process: the same encoder pass over ALL windows; the ranks' features must be these, bit for bit
    dev = torch.device(DEV)
    n_labels = synth.N_LABELS
    tokens, targets, locs = e2e.synthetic_windows(chroms, E2E_KW["windows"], E2E_KW["seq_length"], n_labels)
    torch.manual_seed(0)
    enc = StrandPair(WindowEncoder(n_labels, E2E_KW["seq_length"])).to(dev)
    model = C.ChromeGCN(128, 128, n_labels, 0.0, True, 2).to(dev)
    with torch.no_grad():
        model.out.load_state_dict(enc.model.classifier.state_dict())
        model.batch_norm.load_state_dict(enc.model.batch_norm.state_dict())
    feats = extract_features(enc, tokens.to(dev), targets.to(dev), locs, FeatureCollector(), E2E_KW["batch_size"]).finish()
    for r in res:
        for c, f in r[2].items():
            for k in ("forward", "backward", "target"):
                assert np.array_equal(f[k], feats[c][k].cpu().numpy()), (r[0], c, k)
    # ---- and the trained parameters: single-process emulation of the sharded epochs (gradients of a step group's
    # chromosomes averaged, one fused step), same kernels, eager
    graphs = {c: synth.contact_graph(E2E_KW["windows"], max(1, int(round(synth.PAIRS_PER_CHROM * E2E_KW["windows"] / synth.chrom_nodes(c)))),
                                     synth.chrom_seed(c)) for c in chroms}
    opt = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    st = GCNStage(model, opt, "hic", dev, hip_graphs=False, input_grad=True, cache_input_aggregation=False)
    st.load(feats, graphs)
    plan = plan_shards({c: st._meta[c][2] for c in chroms}, world)
    assert {c for c in chroms if plan.owner[c] == 0} == owned[0]
    model.train()
    for _ in range(E2E_KW["warmup"] + E2E_KW["epochs"]):
        for group in plan.rounds:
            members = [g for g in group if g is not None]
            acc = None
            for nm in members:
                st._ensure_flat_grad()
                st._fwd_bwd(st.chroms[nm])
                acc = st._flat_grad.clone() if acc is None else acc + st._flat_grad
            st._flat_grad.copy_(acc)
            st._optimizer_step(1.0 / len(members))
    ref = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    for r in res:
        for k, v in ref.items():
            if "running" in k or "num_batches" in k:
                continue   # BatchNorm running statistics are rank-averaged (documented deviation)
            np.testing.assert_allclose(r[3][k], v, rtol=1e-4, atol=1e-4, err_msg="rank %d %s" % (r[0], k))
    for k in res[0][3]:   # both ranks end with the same model and the same full predictions
        assert np.array_equal(res[0][3][k], res[1][3][k]), k
    assert np.array_equal(res[0][4], res[1][4])


def test_e2e_pipeline_at_full_sequence_length():
    """configs[4]'s window shape: 2 000-token windows (config_args.py:39), 4 096 of them on one chromosome slot"""
    t, stage, names = e2e.run_pipeline(torch.device(DEV), windows=4096, seq_length=2000, epochs=2, warmup=1,
                                       chroms=("chr22",), batch_size=64)
    assert t["windows"] == 4096 and names == ["chr22"] and stage.chroms["chr22"].n == 4096
    assert stage.chroms["chr22"].x.shape == (2, 4096, 128) and np.isfinite(t["final_loss"])
    assert t["encoder_s"] > 0 and t["gcn_epoch_s"] > 0


@pytest.mark.timeout(900)
def test_e2e_at_one_real_chromosome(tmp_path):
    """BASELINE.json configs[4] at ONE REAL chromosome's size: chr21 of the synthetic genome -- 5 776 windows
    with peaks x 2 000 tokens (pretrain.py:24-63 pushes them through the encoder in batches of 64), its own 250 000 contact
    pairs -- encoder -> device hand-off -> GCN stage -> multi-label metrics, nothing scaled down.  The device hand-off must
    be the reference's `.pt` round trip (utils/util_methods.py:183-199 writes, main.py:30-32 reads) bit for bit at this
    size too, and the stage fed either way must train to identical parameters."""
    from chromegcn_amd import metrics
    dev = torch.device(DEV)
    n = synth.chrom_nodes("chr21")
    t, stage, names = e2e.run_pipeline(dev, windows="full", seq_length=2000, dropout=0.0, epochs=2, warmup=1, chroms=("chr21",),
                                       batch_size=64, return_feats=True)
    assert names == ["chr21"] and t["windows"] == n == 5776 and t["windows_per_chrom"] == {"chr21": n}
    c = stage.chroms["chr21"]
    assert c.n == n and c.x.shape == (2, n, 128) and c.graph.nnz > 2 * 200000   # the full contact budget (+ diagonal)
    assert t["encoder_s"] > 0 and t["gcn_epoch_s"] > 0 and np.isfinite(t["final_loss"])
    feats_dev = t["feats"]
    assert feats_dev["chr21"]["forward"].is_cuda and feats_dev["chr21"]["forward"].shape == (n, 128)
    # ---- the reference's artefact at this size: written from the same collector contents, read back on the host
    col = FeatureCollector()
    col.add([("chr21", 1000 * i, 1000 * i + 1000) for i in range(n)], feats_dev["chr21"]["forward"],
            feats_dev["chr21"]["backward"], feats_dev["chr21"]["target"])
    path = col.save(str(tmp_path / "run.finetune.x"), "train")
    feats_pt = torch.load(path)
    for k in ("forward", "backward", "target"):
        assert feats_pt["chr21"][k].device.type == "cpu" and torch.equal(feats_pt["chr21"][k], feats_dev["chr21"][k].cpu()), k
    # ---- the stage fed from the file trains to the same parameters as the stage fed on the device (run_pipeline's)
    hic = synth.contact_graph(n, synth.PAIRS_PER_CHROM, synth.chrom_seed("chr21"))
    torch.manual_seed(0)
    enc = StrandPair(WindowEncoder(synth.N_LABELS, 2000)).to(dev)   # the same seed as run_pipeline: the same head init
    m = C.ChromeGCN(128, 128, synth.N_LABELS, 0.0, True, 2).to(dev)
    with torch.no_grad():
        m.out.load_state_dict(enc.model.classifier.state_dict())
        m.batch_norm.load_state_dict(enc.model.batch_norm.state_dict())
    del enc
    opt = torch.optim.SGD(m.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    st = GCNStage(m, opt, "hic", dev, hip_graphs=True, input_grad=True, cache_input_aggregation=False)
    st.load(feats_pt, {"chr21": hic})
    for _ in range(3):   # warmup 1 + epochs 2
        st.run_split("train", names, to_cpu=False)
    for k, v in stage.model.state_dict().items():
        assert torch.equal(v, m.state_dict()[k]), k
    # ---- and the metrics stage on the full chromosome's predictions (runner.py:41)
    preds, targets, loss = stage.run_split("valid", names, to_cpu=False)
    res = metrics.compute_metrics(preds, targets, loss)
    assert preds.shape == (n, synth.N_LABELS) and np.isfinite(loss)
    assert all(np.isfinite(res[k]) for k in ("mAP", "meanAUC", "meanAUPR", "meanFDR", "loss"))
    assert len(res["allAUC"]) == synth.N_LABELS   # 5 776 windows at 5 % positives: every label has both classes
