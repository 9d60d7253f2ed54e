"""Module-level drop-in surface on the GPU: GraphConvolution (any width, adj and adj=None), ChromeGCN.forward with
adj=None (models/SubLayers.py:45-48), a seeded randomised sweep of the fused layer against the oracle, non-SGD
optimizers under HIP-graph capture, and the engine's cache invalidation."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import chromegcn_amd as C
from chromegcn_amd import graph as G, ops, synth
from chromegcn_amd.finetune import GCNStage
from oracle import chromegcn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = dict(atol=1e-4, rtol=1e-4)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("d_in,d_out", [(128, 128), (128, 256), (64, 192), (100, 36), (256, 128), (128, 4), (128, 103), (77, 1), (128, 130)])
@pytest.mark.parametrize("adj_kind", ["hic", "both", "coo", "none"])
def test_graph_convolution_module_matches_oracle(d_in, d_out, adj_kind):
    """layers.GraphConvolution.forward(input, adj, deg) = adj @ (input @ W) + b  (models/SubLayers.py:42-52), forward and
    autograd, for the fast widths and for widths only the generic kernel serves; adj as ChromGraph, as the torch sparse
    COO a reference caller passes, and adj=None."""
    n = 700
    rng = np.random.RandomState(d_in + d_out)
    a = O.random_symmetric_graph(n, 6000, 3)
    a_norm = None if adj_kind == "none" else O.normalized_adjacency("both" if adj_kind == "both" else "hic", a, n)
    gc = C.GraphConvolution(d_in, d_out).to(DEV)
    with torch.no_grad():
        gc.weight.copy_(_dev((rng.randn(d_in, d_out) / np.sqrt(d_in)).astype(np.float32)))
        gc.bias.copy_(_dev((rng.randn(d_out) * 0.1).astype(np.float32)))
    x = rng.randn(n, d_in).astype(np.float32)
    gup = rng.randn(n, d_out).astype(np.float32)
    if adj_kind == "none":
        adj = None
    elif adj_kind == "coo":
        adj = O.to_torch_coo(a_norm).to(DEV)          # what the reference's process_graph(...).cuda() hands over
    else:
        adj = G.upload(G.normalize_graph(adj_kind, a, n), DEV)
    xt = _dev(x).requires_grad_(True)
    y = gc(xt, adj, None)
    y.backward(_dev(gup))
    W, b = gc.weight.detach().cpu().numpy().astype(np.float64), gc.bias.detach().cpu().numpy().astype(np.float64)
    s = x.astype(np.float64) @ W
    A = sp.identity(n, format="csr") if a_norm is None else a_norm.astype(np.float64)
    want = A @ s + b
    np.testing.assert_allclose(y.detach().cpu().numpy(), want, **TOL)
    ds = A.T @ gup.astype(np.float64)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), ds @ W.T, **TOL)
    np.testing.assert_allclose(gc.weight.grad.cpu().numpy(), x.astype(np.float64).T @ ds, atol=1e-4 * np.abs(x.astype(np.float64).T @ ds).max(), rtol=1e-4)
    np.testing.assert_allclose(gc.bias.grad.cpu().numpy(), gup.astype(np.float64).sum(0), atol=1e-4 * np.abs(gup.astype(np.float64).sum(0)).max(), rtol=1e-4)


def test_graph_convolution_rejects_widths_the_kernels_cannot_serve_loudly():
    """any out_features works through the module (padded to the next multiple of 4, models/SubLayers.py:8-12 takes any);
    the raw aggregation operator still says what it needs, and widths beyond 4096 columns are refused"""
    g = G.upload(G.normalize_graph("none", None, 10), DEV)
    from chromegcn_amd import ops
    with pytest.raises(RuntimeError, match="multiple of 4"):
        ops.spmm(torch.randn(1, 10, 130, device=DEV), g)
    gc = C.GraphConvolution(8, 4100).to(DEV)
    with pytest.raises(RuntimeError, match="4096"):
        gc(torch.randn(10, 8, device=DEV), g, None)


@pytest.mark.parametrize("layers", [1, 2])
def test_chromegcn_forward_without_adjacency_matches_oracle(layers):
    """ChromeGCN.forward(x, None, None): every GraphConvolution degenerates to X W + b (models/SubLayers.py:45-48)."""
    n, d, c = 333, 128, 19
    torch.manual_seed(5)
    orc = O.GatedGCNOracle(d, c, 0.0, layers)
    with torch.no_grad():
        for k, p in orc.named_parameters():
            if "GC" in k and k.endswith("weight"):
                p.mul_(40)
    m = C.ChromeGCN(d, d, c, 0.0, True, layers)
    m.load_state_dict(orc.state_dict())
    m.to(DEV)
    x = torch.randn(n, d)
    for mode in ("eval", "train"):
        getattr(orc, mode)(); getattr(m, mode)()
        xo = x.clone().requires_grad_(True)
        xh = x.clone().to(DEV).requires_grad_(True)
        _, out_o, (g1o, g2o), _ = orc(xo, None, None)
        x_back, out_h, (g1h, g2h), none = m(xh, None, None)
        assert x_back is xh and none is None
        np.testing.assert_allclose(out_h.detach().cpu().numpy(), out_o.detach().numpy(), **TOL)
        np.testing.assert_allclose(g1h.detach().cpu().numpy(), g1o.detach().numpy(), **TOL)
        assert (g2h is None) == (g2o is None)
        out_o.square().mean().backward()
        out_h.square().mean().backward()
        np.testing.assert_allclose(xh.grad.cpu().numpy(), xo.grad.numpy(), atol=1e-5, rtol=1e-4)
        for (k, po), (_, ph) in zip(orc.named_parameters(), m.named_parameters()):
            np.testing.assert_allclose(ph.grad.cpu().numpy(), po.grad.numpy(), atol=1e-5, rtol=1e-4, err_msg=k)
            po.grad = None; ph.grad = None


def test_randomised_layer_sweep_matches_oracle():
    """seeded, short version of tests/probes/stress_parity.py: random sizes (ragged tiles, 1 .. 3000 nodes), densities,
    adjacency kinds, strands, widths, hub rows; forward + every gradient of the fused layer vs the oracle's numpy math"""
    rng = np.random.RandomState(1234)
    for case in range(24):
        n = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, rng.randint(1, 400), rng.randint(400, 3000)]))
        S = int(rng.choice([1, 2])); d = int(rng.choice([128, 128, 256]))
        adj = str(rng.choice(["hic", "hic", "both", "constant", "none"]))
        if adj == "constant" and n < 8:
            adj = "none"
        pairs = int(rng.choice([0, n // 2, 3 * n, 20 * n, min(60 * n, n * n // 3)]))
        a = O.random_symmetric_graph(n, pairs, int(rng.randint(1 << 30))) if adj in ("hic", "both") else None
        if a is not None and n > 40 and rng.rand() < 0.3:   # a hub row
            a = a.tolil(); hub = int(rng.randint(n)); a[hub, :] = 1; a[:, hub] = 1; a[hub, hub] = 0; a = a.tocsr()
        h = G.normalize_graph(adj, a, n)
        g = G.upload(h, DEV)
        W = (rng.randn(d, d) / np.sqrt(d) * 1.5).astype(np.float32); b = (rng.randn(d) * 0.2).astype(np.float32)
        wg = (rng.randn(d) / np.sqrt(d) * 2).astype(np.float32); cg = np.float32(rng.randn() * 0.3)
        x = rng.randn(S, n, d).astype(np.float32)
        gup = (rng.randn(S, n, d) * 0.1).astype(np.float32); ggate = (rng.randn(S, n) * 0.1).astype(np.float32)
        t = {k: _dev(v).requires_grad_(True) for k, v in dict(x=x, W=W, b=b, wg=wg.reshape(1, d), cg=np.array([cg])).items()}
        xn, gate = ops.gated_layer(t["x"], t["W"], t["b"], t["wg"], t["cg"], g)
        (xn * _dev(gup)).sum().add((gate * _dev(ggate)).sum()).backward()
        spm = h.to_scipy()
        acc = {k: 0.0 for k in ["dW", "db", "dwg", "dcg"]}
        tag = "case %d n=%d S=%d d=%d adj=%s nnz=%d" % (case, n, S, d, adj, h.nnz)
        for s in range(S):
            f = O.layer_forward_np(spm, x[s], W, b, wg, float(cg))
            np.testing.assert_allclose(xn[s].detach().cpu().numpy(), f["Xn"], err_msg=tag, **TOL)
            np.testing.assert_allclose(gate[s].detach().cpu().numpy(), f["g"], err_msg=tag, **TOL)
            bw = O.layer_backward_np(spm, x[s], W, wg, f["Z"], f["g"], gup[s], ggate[s])
            np.testing.assert_allclose(t["x"].grad[s].cpu().numpy(), bw["dX"], err_msg=tag + " dX", **TOL)
            for k in acc:
                acc[k] = acc[k] + bw[k]
        scale = max(1.0, float(np.abs(acc["dW"]).max()))
        for k, got in (("dW", t["W"].grad), ("db", t["b"].grad), ("dwg", t["wg"].grad), ("dcg", t["cg"].grad)):
            np.testing.assert_allclose(got.cpu().numpy().reshape(np.shape(acc[k])), acc[k], rtol=1e-4, atol=1e-4 * scale,
                                       err_msg=tag + " " + k)


def _small_stage(optimizer_name, hip_graphs, seed=0):
    n, d, c = 400, 128, 11
    feats = synth.chrom_features(n, d, c, 5)
    hic = synth.contact_graph(n, 3000, 5)
    torch.manual_seed(seed)
    orc = O.GatedGCNOracle(d, c, 0.0, 2)
    with torch.no_grad():
        orc.GC1.weight.mul_(40); orc.GC2.weight.mul_(40)
    m = C.ChromeGCN(d, d, c, 0.0, True, 2)
    m.load_state_dict(orc.state_dict())
    m.to(DEV)
    mk = {"adam": lambda ps: torch.optim.Adam(ps, betas=(0.9, 0.98), lr=1e-3),           # utils/util_methods.py:20-21
          "sgd": lambda ps: torch.optim.SGD(ps, lr=0.25, momentum=0.9, weight_decay=1e-6)}[optimizer_name]
    st = GCNStage(m, mk(m.parameters()), "hic", DEV, hip_graphs=hip_graphs)
    st.add_chromosome("c", feats, hic)
    return st, m, orc, mk, feats, hic


def test_adam_steps_under_hip_graph_capture_match_oracle_and_eager():
    """the reference's `-optim adam` (get_optimizer, utils/util_methods.py:20-21): torch's Adam.step is not capturable,
    so the engine captures forward+backward only and steps eagerly; three steps vs the oracle and vs the eager engine"""
    st_g, m_g, orc, mk, feats, hic = _small_stage("adam", True)
    st_e, m_e, _, _, _, _ = _small_stage("adam", False)
    oopt = mk(orc.parameters())
    cache = {}
    for step in range(3):
        lg, pg, _ = st_g.train_step("c")
        le, pe, _ = st_e.train_step("c")
        _, _, tot = O.finetune_epoch(orc, {"c": feats}, {"c": hic}, oopt, "train", "hic", adj_cache=cache)
        assert torch.equal(lg, le) and torch.equal(pg, pe)
        assert abs(lg.item() - tot) < 1e-4
    osd = orc.state_dict()
    for (k, vg), (_, ve) in zip(m_g.state_dict().items(), m_e.state_dict().items()):
        assert torch.equal(vg, ve), k
        # Adam divides by sqrt(v): tiny-gradient parameters amplify fp32 differences, hence the looser bound
        np.testing.assert_allclose(vg.cpu().numpy(), osd[k].numpy(), atol=2e-3, rtol=1e-3, err_msg=k)


def test_capture_failure_leaves_the_model_as_it_was():
    st, m, _, _, _, _ = _small_stage("sgd", True)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    st._ensure_flat_grad()
    st._ensure_arena()
    calls = {"n": 0}
    orig = st._fwd_bwd

    def boom(*a, **k):
        out = orig(*a, **k)          # the step itself runs (and mutates the model: the optimizer step rides in it) ...
        calls["n"] += 1
        if calls["n"] >= 3:          # ... twice as warm-up; the third call, under capture, fails
            raise RuntimeError("boom")
        return out
    st._fwd_bwd = boom
    with pytest.raises(RuntimeError):
        st.train_step("c")
    torch.cuda.synchronize()
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k


def test_fused_head_backward_in_eval_mode_raises_a_clear_error():
    st, m, _, _, feats, hic = _small_stage("sgd", False)
    m.eval()
    c = st.chroms["c"]
    x = c.x.clone().requires_grad_(True)
    loss, probs, _ = m.forward_loss(x, c.graph, c.target)
    with pytest.raises(RuntimeError, match="train mode"):
        loss.backward()


def test_stage_rebuilds_a_chromosome_whose_inputs_changed():
    """GCNStage.load keys its device cache on the caller's tensors / graph object (not on the name and size alone)"""
    st, m, _, _, feats, hic = _small_stage("sgd", True)
    st.load({"c": feats}, {"c": hic})
    first = st.chroms["c"]
    st.load({"c": feats}, {"c": hic})
    assert st.chroms["c"] is first                                  # same inputs: reused
    feats2 = {k: v.clone() for k, v in feats.items()}
    feats2["target"] = 1 - feats2["target"]
    st.load({"c": feats2}, {"c": hic})
    assert st.chroms["c"] is not first                              # new tensors under the same name: rebuilt
    assert torch.equal(st.chroms["c"].target.cpu(), feats2["target"])
    second = st.chroms["c"]
    feats2["forward"].add_(1.0)                                     # in-place edit bumps the version counter
    st.load({"c": feats2}, {"c": hic})
    assert st.chroms["c"] is not second
    hic2 = synth.contact_graph(400, 3000, 99)
    third = st.chroms["c"]
    st.load({"c": feats2}, {"c": hic2})
    assert st.chroms["c"] is not third and st.chroms["c"].graph.nnz == hic2.nnz + 400
    loss, _, _ = st.train_step("c")                                 # and the rebuilt chromosome still steps
    assert torch.isfinite(loss)


def test_batchnorm_training_needs_two_rows():
    """train-mode BatchNorm over one window: torch raises (ValueError: Expected more than 1 value per channel);
    the fused head refuses it as a bad argument instead of writing inf/NaN running variance"""
    m = C.ChromeGCN(128, 128, 5, 0.0, True, 1).to(DEV)
    g = G.upload(G.normalize_graph("none", None, 1), DEV)
    x = torch.randn(2, 1, 128, device=DEV)
    m.train()
    with pytest.raises(RuntimeError, match="bad argument"):
        m.forward_loss(x, g, torch.zeros(1, 5, device=DEV))
    assert torch.isfinite(m.batch_norm.running_var).all()


@pytest.mark.parametrize("d,c,layers", [(128, 103, 2), (128, 164, 2), (256, 256, 4), (128, 1, 1)])
def test_module_forward_in_eval_under_no_grad_runs_the_fused_head_and_matches_oracle(d, c, layers, monkeypatch):
    """model.eval(); with torch.no_grad(): model(x, adj) -- how the reference's evaluation and visualisation code calls the
    module (finetune.py:41-42 under the valid / test splits) -- goes through cgcn_head_logits (relu -> BatchNorm with the
    running statistics -> Linear, one kernel); with gradients enabled the same call takes the differentiable torch ops.
    Both must match the oracle module; forward_strands likewise for both strands at once."""
    n = 777
    torch.manual_seed(11)
    orc = O.GatedGCNOracle(d, c, 0.3, layers)
    with torch.no_grad():
        orc.batch_norm.running_mean.normal_(0.1, 0.3)
        orc.batch_norm.running_var.uniform_(0.4, 2.0)
        orc.batch_norm.weight.uniform_(0.5, 1.5)
        orc.batch_norm.bias.normal_(0, 0.2)
    m = C.ChromeGCN(d, d, c, 0.3, True, layers)
    m.load_state_dict(orc.state_dict())
    m.to(DEV)
    adj = synth.contact_graph(n, 4000, 3)
    g = G.upload(G.normalize_graph("hic", adj, n), DEV)
    adj_o = O.process_graph("hic", {"c": adj}, n, "c")
    xf, xr = torch.randn(n, d), torch.randn(n, d)
    orc.eval(); m.eval()
    calls = []
    real = ops.head_logits
    monkeypatch.setattr(ops, "head_logits", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        want_f = orc(xf, adj_o, None)[1].numpy()
        want_r = orc(xr, adj_o, None)[1].numpy()
        got_f = m(xf.to(DEV), g)[1]
        assert len(calls) == 1 and tuple(got_f.shape) == (n, c)
        both = m.forward_strands(torch.stack([xf, xr]).to(DEV), g)[0]
        assert len(calls) == 2 and tuple(both.shape) == (2, n, c)
    np.testing.assert_allclose(got_f.cpu().numpy(), want_f, **TOL)
    np.testing.assert_allclose(both[0].cpu().numpy(), want_f, **TOL)
    np.testing.assert_allclose(both[1].cpu().numpy(), want_r, **TOL)
    rm = m.batch_norm.running_mean.clone()
    got_grad = m(xf.to(DEV), g)[1]                 # gradients enabled: the torch ops (differentiable), same numbers
    assert len(calls) == 2 and got_grad.requires_grad
    np.testing.assert_allclose(got_grad.detach().cpu().numpy(), got_f.cpu().numpy(), atol=2e-5, rtol=1e-5)
    assert torch.equal(rm, m.batch_norm.running_mean)
