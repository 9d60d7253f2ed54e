"""The stage loop on the GPU against G4 (recorded from the reference's model + process_graph driven
in finetune.py's order): per-epoch predictions, loss totals, final parameters -- eagerly and through
the captured HIP graphs (which must leave the training state bit-identical to never having warmed up)."""
import types

import numpy as np
import pytest
import torch

import chromegcn_amd as C
from chromegcn_amd.finetune import finetune, run_epoch, GCNStage
from helpers import csr_from, state_from

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _load(z):
    chroms = [str(c) for c in z["chroms"]]
    feats, graphs = {}, {}
    for c in chroms:
        graphs[c] = csr_from(z, c + "_in")
        feats[c] = {"forward": torch.from_numpy(z[c + "_xf"]), "backward": torch.from_numpy(z[c + "_xr"]),
                    "target": torch.from_numpy(z[c + "_tgt"])}
    return chroms, feats, graphs


@pytest.mark.parametrize("hip_graphs", [False, True])
def test_finetune_loop_matches_reference_golden(golden, hip_graphs):
    z = golden("g4_finetune_loop.npz")
    chroms, feats, graphs = _load(z)
    init = state_from(z, "init")
    m = C.ChromeGCN(128, 128, init["out.weight"].shape[0], 0.0, True, 2)
    m.load_state_dict(init)
    m.to(DEV)
    optim = torch.optim.SGD(m.parameters(), lr=0.25, weight_decay=1e-6, momentum=0.9)
    opt = types.SimpleNamespace(adj_type="hic", hip_graphs=hip_graphs)
    ref_tr = z["train_losses"]
    for e in range(2):
        preds, targets, total = finetune(None, m, feats, None, optim, e + 1, None, opt, "train", split_adj_dict=graphs)
        assert preds.device.type == "cpu" and preds.shape == targets.shape
        np.testing.assert_allclose(preds.numpy(), z["train_preds_e%d" % e], atol=1e-4, rtol=1e-4)
        assert abs(total - ref_tr[3 * e:3 * e + 3].sum()) < 2e-4
        np.testing.assert_array_equal(targets.numpy(), np.concatenate([z[c + "_tgt"] for c in chroms]))
    preds, targets, total, elapsed = run_epoch(None, m, feats, None, optim, 3, None, opt, "valid", split_adj_dict=graphs)
    np.testing.assert_allclose(preds.numpy(), z["eval_preds"], atol=1e-4, rtol=1e-4)
    assert abs(total - z["eval_losses"].sum()) < 2e-4 and elapsed >= 0
    final = state_from(z, "final")
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), final[k].numpy(), atol=1e-4, rtol=1e-4, err_msg=k)


def test_captured_step_equals_eager_step_bitwise(golden):
    z = golden("g4_finetune_loop.npz")
    chroms, feats, graphs = _load(z)
    init = state_from(z, "init")
    outs = []
    for hip_graphs in (False, True):
        m = C.ChromeGCN(128, 128, init["out.weight"].shape[0], 0.0, True, 2)
        m.load_state_dict(init); m.to(DEV)
        optim = torch.optim.SGD(m.parameters(), lr=0.25, weight_decay=1e-6, momentum=0.9)
        st = GCNStage(m, optim, "hic", DEV, hip_graphs=hip_graphs, input_grad=True)
        st.load(feats, graphs)
        for _ in range(2):
            for c in chroms:
                loss, probs, dx = st.train_step(c)
        outs.append(({k: v.clone() for k, v in m.state_dict().items()}, loss.clone(), dx.clone()))
    for k in outs[0][0]:
        assert torch.equal(outs[0][0][k], outs[1][0][k]), k
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def test_step_group_path_equals_single_rank_step(golden):
    """train_group (the multi-rank path: captured fwd+bwd, all-reduce, optimizer step) with a group of one
    must reproduce train_step bit for bit."""
    z = golden("g4_finetune_loop.npz")
    chroms, feats, graphs = _load(z)
    init = state_from(z, "init")
    outs = []
    for use_group in (False, True):
        m = C.ChromeGCN(128, 128, init["out.weight"].shape[0], 0.0, True, 2)
        m.load_state_dict(init); m.to(DEV)
        optim = torch.optim.SGD(m.parameters(), lr=0.25, weight_decay=1e-6, momentum=0.9)
        st = GCNStage(m, optim, "hic", DEV, hip_graphs=True)
        st.load(feats, graphs)
        for _ in range(2):
            for c in chroms:
                loss, probs, dx = st.train_group(c, 1) if use_group else st.train_step(c)
        outs.append(({k: v.clone() for k, v in m.state_dict().items()}, loss.clone()))
    for k in outs[0][0]:
        assert torch.equal(outs[0][0][k], outs[1][0][k]), k
    assert torch.equal(outs[0][1], outs[1][1])


def test_finetune_reads_graph_pickle_like_the_reference(tmp_path, golden):
    """finetune.py:20-23: graphs come from <graph_root>/<split>_graphs_<hicsize>_<hicnorm>norm.pkl"""
    import pickle
    z = golden("g4_finetune_loop.npz")
    chroms, feats, graphs = _load(z)
    with open(str(tmp_path / "valid_graphs_500000_SQRTVCnorm.pkl"), "wb") as f:
        pickle.dump(graphs, f)
    init = state_from(z, "final")
    m = C.ChromeGCN(128, 128, init["out.weight"].shape[0], 0.0, True, 2)
    m.load_state_dict(init); m.to(DEV)
    opt = types.SimpleNamespace(adj_type="hic", graph_root=str(tmp_path), hicsize="500000", hicnorm="SQRTVC")
    preds, targets, total = finetune(None, m, feats, None, None, 1, None, opt, "valid")
    np.testing.assert_allclose(preds.numpy(), z["eval_preds"], atol=1e-4, rtol=1e-4)
    assert abs(total - z["eval_losses"].sum()) < 2e-4


def test_training_with_dropout_learns_a_planted_signal():
    """end-to-end sanity with everything on (dropout 0.2, captured graphs, fused SGD): targets planted by a
    teacher network of the same family must become predictable -- the loss has to fall well below chance."""
    from chromegcn_amd import synth
    torch.manual_seed(0)
    n, d, c = 1500, 128, 12
    hic = synth.contact_graph(n, 12000, 3)
    feats = synth.chrom_features(n, d, c, 4)
    teacher = C.ChromeGCN(d, d, c, 0.0, True, 2).to(DEV).eval()
    with torch.no_grad():
        for k, p in teacher.named_parameters():
            if p.dim() == 2:
                p.copy_(torch.randn_like(p) / np.sqrt(p.shape[-1]) * 2.0)
        g = C.process_graph("hic", {"c": hic}, n, "c", device=DEV)
        logits, _ = teacher.forward_strands(torch.stack([feats["forward"], feats["backward"]]).to(DEV), g)
        feats["target"] = ((logits[0] + logits[1]) / 2 > 0).float().cpu()
    model = C.ChromeGCN(d, d, c, 0.2, True, 2).to(DEV)
    optim = torch.optim.SGD(model.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
    st = GCNStage(model, optim, "hic", DEV, hip_graphs=True)
    st.add_chromosome("c", feats, hic)
    first = st.train_step("c")[0].item()
    for _ in range(150):
        loss = st.train_step("c")[0]
    ev, probs = st.eval_step("c")
    acc = ((probs > 0.5).float().cpu() == feats["target"]).float().mean().item()
    assert loss.item() < 0.6 * first and acc > 0.75, (first, loss.item(), ev.item(), acc)


def test_cached_input_aggregation_is_bitwise_identical(golden):
    """A X of the first layer is loop invariant; streaming the cached copy must not change a single bit of the
    training trajectory, the predictions or d loss / d features."""
    z = golden("g4_finetune_loop.npz")
    chroms, feats, graphs = _load(z)
    init = state_from(z, "init")
    outs = []
    for cache in (False, True):
        m = C.ChromeGCN(128, 128, init["out.weight"].shape[0], 0.0, True, 2)
        m.load_state_dict(init); m.to(DEV)
        optim = torch.optim.SGD(m.parameters(), lr=0.25, weight_decay=1e-6, momentum=0.9)
        st = GCNStage(m, optim, "hic", DEV, hip_graphs=True, cache_input_aggregation=cache, input_grad=True)
        st.load(feats, graphs)
        for _ in range(3):
            for c in chroms:
                loss, probs, dx = st.train_step(c)
        ev = [st.eval_step(c)[1].clone() for c in chroms]
        assert (st.chroms[chroms[0]].h1["h"] is not None) == cache
        outs.append(({k: v.clone() for k, v in m.state_dict().items()}, loss.clone(), dx.clone(), ev))
    for k in outs[0][0]:
        assert torch.equal(outs[0][0][k], outs[1][0][k]), k
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    for a, b in zip(outs[0][3], outs[1][3]):
        assert torch.equal(a, b)


def test_cpu_predictions_of_the_drop_in_equal_the_device_arena_bitwise(golden):
    """finetune() returns CPU predictions like the reference (finetune.py:52-53,67).  They leave the device chromosome by
    chromosome on a copy stream behind each step (pinned host arena): what arrives must be the device arena bit for bit,
    in the reference's chromosome order, for a split that is the whole stage and for a subset in another order; the
    tensor of one call must survive the next call (two host buffers alternate)."""
    z = golden("g4_finetune_loop.npz")
    chroms, feats, graphs = _load(z)
    init = state_from(z, "init")
    m = C.ChromeGCN(128, 128, init["out.weight"].shape[0], 0.2, True, 2)
    m.load_state_dict(init); m.to(DEV)
    optim = torch.optim.SGD(m.parameters(), lr=0.05, weight_decay=1e-6, momentum=0.9)
    st = GCNStage(m, optim, "hic", DEV, hip_graphs=True)
    st.load(feats, graphs)
    kept = []
    for e in range(3):
        preds, targets, total = st.run_split("train", chroms)           # to_cpu=True is the default
        torch.cuda.synchronize()
        assert preds.device.type == "cpu" and preds.is_pinned()
        assert torch.equal(preds, st._arena["probs"].cpu())
        dev_preds, dev_targets, dev_total = st.run_split("valid", chroms, to_cpu=False)
        assert torch.equal(targets, dev_targets.cpu())
        kept.append((preds, preds.clone()))
        if e >= 1:   # the tensor returned one call ago is still intact
            assert torch.equal(kept[e - 1][0], kept[e - 1][1])
    assert not torch.equal(kept[0][1], kept[2][1])                      # (the model did train in between)
    # a subset in another order: not a contiguous run of the arena -> concatenated from the host arena
    sub = [chroms[2], chroms[0]]
    preds, targets, total = st.run_split("valid", sub)
    rows = st._arena["rows"]
    want = torch.cat([st._arena["probs"][rows[c][0]:rows[c][1]] for c in sub]).cpu()
    assert torch.equal(preds, want)
    assert torch.equal(targets, torch.cat([feats[c]["target"] for c in sub]))


def test_one_graph_per_split_equals_one_graph_per_chromosome_bitwise(golden):
    """run_split(to_cpu=False) on one rank replays ONE graph for the whole split (GCNStage(epoch_graph=True), the default).
    Same kernels in the same order as one graph per chromosome: parameters, momentum, dropout state, losses and predictions
    must agree bit for bit over several epochs (dropout on), for train and eval splits, for a subset in another order, across
    a learning-rate change (the rate is baked into the captured optimizer kernels) and after a chromosome is reloaded."""
    z = golden("g4_finetune_loop.npz")
    chroms, feats, graphs = _load(z)
    init = state_from(z, "init")
    runs = []
    for whole in (False, True):
        m = C.ChromeGCN(128, 128, init["out.weight"].shape[0], 0.2, True, 2)
        m.load_state_dict(init); m.to(DEV)
        optim = torch.optim.SGD(m.parameters(), lr=0.05, weight_decay=1e-6, momentum=0.9)
        st = GCNStage(m, optim, "hic", DEV, hip_graphs=True, epoch_graph=whole)
        st.load(feats, graphs)
        log = []
        for e in range(4):
            if e == 2:
                optim.param_groups[0]["lr"] = 0.01
            if e == 3:   # new data for one chromosome: every graph that touches it must be re-captured
                c0 = chroms[0]
                st.add_chromosome(c0, {"forward": feats[c0]["forward"] * 0.5, "backward": feats[c0]["backward"] * 0.5,
                                       "target": 1.0 - feats[c0]["target"]}, graphs[c0])
            p, t, loss = st.run_split("train", chroms, to_cpu=False, sync_loss=False)
            log.append((p.clone(), loss.clone()))
            p, t, loss = st.run_split("valid", chroms, to_cpu=False)
            log.append((p.clone(), torch.tensor(loss)))
            p, t, loss = st.run_split("train", [chroms[-1], chroms[0]], to_cpu=False)
            log.append((p.clone(), torch.tensor(loss)))
        keys = [k for k in st._graphs]
        assert any(isinstance(k[0], tuple) for k in keys) == whole, keys
        runs.append((log, {k: v.clone() for k, v in m.state_dict().items()},
                     [optim.state[p_]["momentum_buffer"].clone() for p_ in m.parameters() if p_ in optim.state and "momentum_buffer" in optim.state[p_]]))
    for (pa, la), (pb, lb) in zip(runs[0][0], runs[1][0]):
        assert torch.equal(pa, pb) and torch.equal(la.cpu(), lb.cpu())
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k
    for a, b in zip(runs[0][2], runs[1][2]):
        assert torch.equal(a, b)
