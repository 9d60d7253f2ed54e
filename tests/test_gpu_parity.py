"""Parity tests proper: the HIP path (through the C ABI) against the oracle and against the golden
vectors recorded from the reference.  fp32, atol = rtol = 1e-4 (the north-star tolerance) unless a
tighter bound is stated.  Eval mode or dropout = 0 (device RNG streams differ)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

import chromegcn_amd as C
from chromegcn_amd import graph as G
from chromegcn_amd import ops
from oracle import chromegcn_oracle as O
from helpers import csr_from, state_from

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)
DEV = "cuda"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def graph_cases():
    """(name, HostCSR) -- covers implicit/explicit values, empty rows, a dense row, rows longer
    than one and two 64-entry index chunks, n not a multiple of the 16-node tile, n = 1."""
    out = []
    a = O.random_symmetric_graph(257, 1500, 1)
    out.append(("hic257", G.normalize_graph("hic", a, 257)))
    out.append(("both97", G.normalize_graph("both", O.random_symmetric_graph(97, 300, 2), 97)))
    out.append(("const61", G.normalize_graph("constant", None, 61)))
    out.append(("none33", G.normalize_graph("none", None, 33)))
    out.append(("one", G.normalize_graph("hic", sp.csr_matrix((1, 1)), 1)))
    m = np.zeros((300, 300)); m[5, :] = 1; m[:, 5] = 1; m[5, 5] = 0
    rng = np.random.RandomState(3)
    i, j = rng.randint(0, 300, 4000), rng.randint(0, 300, 4000)
    m[i, j] = 1; m[j, i] = 1; np.fill_diagonal(m, 0)
    m[77, :] = 0; m[:, 77] = 0; m[77, 77] = -1  # empty row after +I
    out.append(("dense300", G.normalize_graph("hic", sp.csr_matrix(m), 300)))
    # Hi-C hub: one node in contact with 1500 others (> LONG_ROW = 512 -> split across the workgroup's
    # waves), plus a second hub in the same 8-node tile
    hub = np.zeros((1700, 1700))
    sel = rng.choice(1700, 1500, replace=False)
    hub[40, sel] = 1; hub[sel, 40] = 1
    sel2 = rng.choice(1700, 700, replace=False)
    hub[43, sel2] = 1; hub[sel2, 43] = 1
    i, j = rng.randint(0, 1700, 6000), rng.randint(0, 1700, 6000)
    hub[i, j] = 1; hub[j, i] = 1; np.fill_diagonal(hub, 0)
    out.append(("hub1700", G.normalize_graph("hic", sp.csr_matrix(hub), 1700)))
    out.append(("hubboth", G.normalize_graph("both", sp.csr_matrix(hub[:900, :900]), 900)))
    # an asymmetric, arbitrary-valued operator (what a torch-COO caller may hand in)
    r = sp.random(90, 90, 0.08, format="csr", random_state=4, dtype=np.float32)
    out.append(("asym90", G.host_csr_from_matrix(r)))
    return out


@pytest.mark.parametrize("split_forward", ["fused", "split"], indirect=True)
@pytest.mark.parametrize("S,d", [(1, 128), (2, 128), (1, 256), (2, 256)])
def test_spmm_matches_oracle(S, d, split_forward):
    for name, h in graph_cases():
        g = G.upload(h, DEV)
        rng = np.random.RandomState(10)
        x = rng.randn(S, h.n, d).astype(np.float32)
        xt = dev(x).requires_grad_(True)
        y = ops.spmm(xt, g)
        a = h.to_scipy().astype(np.float64)
        want = np.stack([a @ x[s].astype(np.float64) for s in range(S)])
        np.testing.assert_allclose(y.detach().cpu().numpy(), want, err_msg=name, **TOL)
        dy = rng.randn(S, h.n, d).astype(np.float32)
        y.backward(dev(dy))
        want_dx = np.stack([a.T @ dy[s].astype(np.float64) for s in range(S)])
        np.testing.assert_allclose(xt.grad.cpu().numpy(), want_dx, err_msg=name, **TOL)


def _layer_params(d, seed):
    rng = np.random.RandomState(seed)
    return ((rng.randn(d, d) / np.sqrt(d) * 1.5).astype(np.float32), (rng.randn(d) * 0.2).astype(np.float32),
            (rng.randn(d) / np.sqrt(d) * 2).astype(np.float32), np.float32(rng.randn() * 0.3))


@pytest.fixture
def split_forward(request):
    """route of cgcn_layer_fwd / cgcn_spmm: 'fused' = built-in choice (whole-row gathers at these sizes), 'split' =
    forced feature-sliced route (k_aggregate_sliced; for the layer: into H, then k_layer_dense), which full-size
    chromosomes take by default"""
    from chromegcn_amd import _lib
    lib = _lib.load()
    lib.cgcn_debug_set_fwd_split_bytes(0 if request.param == "split" else -1)
    yield request.param
    lib.cgcn_debug_set_fwd_split_bytes(-1)


@pytest.mark.parametrize("split_forward", ["fused", "split"], indirect=True)
@pytest.mark.parametrize("S,d", [(1, 128), (2, 128), (1, 256), (2, 256)])
def test_gated_layer_forward_backward_matches_oracle(S, d, split_forward):
    for name, h in graph_cases():
        g = G.upload(h, DEV)
        rng = np.random.RandomState(20)
        W, b, wg, cg = _layer_params(d, 21)
        x = rng.randn(S, h.n, d).astype(np.float32)
        gup = (rng.randn(S, h.n, d) * 0.1).astype(np.float32)
        ggate = (rng.randn(S, h.n) * 0.1).astype(np.float32)
        t = {k: dev(v).requires_grad_(True) for k, v in dict(x=x, W=W, b=b, wg=wg.reshape(1, d), cg=np.array([cg])).items()}
        xn, gate = ops.gated_layer(t["x"], t["W"], t["b"], t["wg"], t["cg"], g)
        (xn * dev(gup)).sum().add((gate * dev(ggate)).sum()).backward()
        a = h.to_scipy()
        acc = {k: 0.0 for k in ["dW", "db", "dwg", "dcg"]}
        for s in range(S):
            f = O.layer_forward_np(a, x[s], W, b, wg, float(cg))
            np.testing.assert_allclose(xn[s].detach().cpu().numpy(), f["Xn"], err_msg=name, **TOL)
            np.testing.assert_allclose(gate[s].detach().cpu().numpy(), f["g"], err_msg=name, **TOL)
            bw = O.layer_backward_np(a, x[s], W, wg, f["Z"], f["g"], gup[s], ggate[s])
            np.testing.assert_allclose(t["x"].grad[s].cpu().numpy(), bw["dX"], err_msg=name + " dX", **TOL)
            for k in acc:
                acc[k] = acc[k] + bw[k]
        np.testing.assert_allclose(t["W"].grad.cpu().numpy(), acc["dW"], err_msg=name + " dW", **TOL)
        np.testing.assert_allclose(t["b"].grad.cpu().numpy(), acc["db"], err_msg=name + " db", **TOL)
        np.testing.assert_allclose(t["wg"].grad.cpu().numpy().ravel(), acc["dwg"], err_msg=name + " dwg", **TOL)
        np.testing.assert_allclose(t["cg"].grad.cpu().numpy().ravel()[0], acc["dcg"], err_msg=name + " dcg", **TOL)


def test_ring_backward_over_many_slot_generations():
    """k_bwd_rowlocal_ring walks an 8-slot LDS ring with monotonic FULL / FREE counters: a workgroup of the 256 reuses every
    slot once per 8 x 16 rows.  n = 70 000 windows x 2 strands = 8 750 slots = 34 per workgroup: more than four trips
    round the ring (the full-size oracle cases reach two), a last slot that is not full, and a row range that crosses the
    strand boundary inside a slot.  Identity adjacency ('none', utils/util_methods.py:173-174) keeps the host oracle
    cheap; the row-local kernel does not look at the graph."""
    n, S, d = 70003, 2, 128
    h = G.normalize_graph("none", None, n)
    g = G.upload(h, DEV)
    rng = np.random.RandomState(31)
    W, b, wg, cg = _layer_params(d, 32)
    x = rng.randn(S, n, d).astype(np.float32)
    gup = (rng.randn(S, n, d) * 0.1).astype(np.float32)
    t = {k: dev(v).requires_grad_(True) for k, v in dict(x=x, W=W, b=b, wg=wg.reshape(1, d), cg=np.array([cg])).items()}
    xn, gate = ops.gated_layer(t["x"], t["W"], t["b"], t["wg"], t["cg"], g)
    (xn * dev(gup)).sum().backward()
    a = h.to_scipy()
    acc = {k: 0.0 for k in ["dW", "db", "dwg", "dcg"]}
    for s in range(S):
        f = O.layer_forward_np(a, x[s].astype(np.float64), W.astype(np.float64), b.astype(np.float64), wg.astype(np.float64), float(cg))
        bw = O.layer_backward_np(a, x[s].astype(np.float64), W.astype(np.float64), wg.astype(np.float64), f["Z"], f["g"],
                                 gup[s].astype(np.float64), np.zeros(n))
        np.testing.assert_allclose(t["x"].grad[s].cpu().numpy(), bw["dX"], err_msg="dX strand %d" % s, **TOL)
        for k in acc:
            acc[k] = acc[k] + bw[k]
    rel = lambda got, want: float(np.abs(got - want).max() / np.abs(want).max())
    assert rel(t["W"].grad.cpu().numpy(), acc["dW"]) < 1e-4
    assert rel(t["b"].grad.cpu().numpy(), acc["db"]) < 1e-4
    assert rel(t["wg"].grad.cpu().numpy().ravel(), acc["dwg"]) < 1e-4
    assert abs(t["cg"].grad.cpu().numpy().ravel()[0] - acc["dcg"]) < 1e-4 * max(1.0, abs(acc["dcg"]))


def test_gated_layer_against_reference_golden(golden):
    """G2: values recorded from the reference's own GraphConvolution + gate math."""
    z = golden("g2_gated_layer.npz")
    for name in z["cases"]:
        adj_type = name.split("_")[-1]
        a_in = csr_from(z, name + "_in"); n = a_in.shape[0]
        g = C.process_graph(adj_type, {"c": a_in}, n, "c", device=DEV)
        t = {k: dev(z[name + "_" + k]).requires_grad_(True) for k in ["X", "W", "b", "wg", "cg"]}
        xn, gate = ops.gated_layer(t["X"].unsqueeze(0), t["W"], t["b"], t["wg"], t["cg"], g)
        np.testing.assert_allclose(xn[0].detach().cpu().numpy(), z[name + "_Xn"], err_msg=name, **TOL)
        np.testing.assert_allclose(gate[0].detach().cpu().numpy(), z[name + "_g"].ravel(), err_msg=name, **TOL)
        xn.backward(dev(z[name + "_Gup"]).unsqueeze(0))
        for k, gk in [("X", "dX"), ("W", "dW"), ("b", "db"), ("wg", "dwg"), ("cg", "dcg")]:
            np.testing.assert_allclose(t[k].grad.cpu().numpy(), z[name + "_" + gk], err_msg=name + gk, **TOL)


def test_layer_results_are_bit_reproducible():
    h = G.normalize_graph("hic", O.random_symmetric_graph(500, 4000, 5), 500)
    g = G.upload(h, DEV)
    W, b, wg, cg = _layer_params(128, 6)
    x = dev(np.random.RandomState(7).randn(2, 500, 128).astype(np.float32))
    outs = []
    for _ in range(3):
        t = [dev(v).requires_grad_(True) for v in (W, b, wg.reshape(1, -1), np.array([cg]))]
        xx = x.clone().requires_grad_(True)
        xn, gate = ops.gated_layer(xx, *t, g)
        xn.sum().backward()
        outs.append([xn.detach().clone(), xx.grad.clone()] + [p.grad.clone() for p in t])
    for o in outs[1:]:
        for a, b_ in zip(outs[0], o):
            assert torch.equal(a, b_)


def _load_model(name, z, L_override=None):
    _, d, L = name.split("_")
    d = int(d[1:]); L = int(L[1:])
    init = state_from(z, name + "_init")
    m = C.ChromeGCN(d, d, init["out.weight"].shape[0], 0.0, True, L)
    m.load_state_dict(init)  # reference state_dict keys load as-is
    return m.to(DEV), d, L


def test_model_eval_forward_against_reference_golden(golden):
    z = golden("g3_model.npz")
    for name in z["cases"]:
        m, d, L = _load_model(name, z)
        a_in = csr_from(z, name + "_in"); n = a_in.shape[0]
        g = C.process_graph("hic", {"c": a_in}, n, "c", device=DEV)
        m.eval()
        with torch.no_grad():
            x_in, lf, gates, none = m(dev(z[name + "_xf"]), g, None)
            _, lr, _, _ = m(dev(z[name + "_xr"]), g, None)
            both, _ = m.forward_strands(torch.stack([dev(z[name + "_xf"]), dev(z[name + "_xr"])]), g)
        assert none is None and x_in.shape == (n, d)
        np.testing.assert_allclose(lf.cpu().numpy(), z[name + "_eval_logits_f"], **TOL)
        np.testing.assert_allclose(lr.cpu().numpy(), z[name + "_eval_logits_r"], **TOL)
        np.testing.assert_allclose(both[0].cpu().numpy(), z[name + "_eval_logits_f"], **TOL)
        np.testing.assert_allclose(both[1].cpu().numpy(), z[name + "_eval_logits_r"], **TOL)
        assert gates[0].shape == (n, 1)
        np.testing.assert_allclose(gates[0].cpu().numpy(), z[name + "_eval_g1"], **TOL)
        if L == 2:
            np.testing.assert_allclose(gates[1].cpu().numpy(), z[name + "_eval_g2"], **TOL)
        else:
            assert gates[1] is None


@pytest.mark.parametrize("split_forward", ["fused", "split"], indirect=True)
@pytest.mark.parametrize("batched", [False, True])
def test_model_train_steps_against_reference_golden(golden, batched, split_forward):
    """loss, every gradient (incl. d/dx_in, finetune.py:33-34), parameters and BatchNorm running
    statistics after one and two SGD steps (lr .25, momentum .9, wd 1e-6)."""
    z = golden("g3_model.npz")
    for name in z["cases"]:
        m, d, L = _load_model(name, z)
        a_in = csr_from(z, name + "_in"); n = a_in.shape[0]
        # reference-style caller: a torch sparse COO adjacency
        adj = O.process_graph("hic", {"c": a_in}, n, "c").to(DEV) if not batched else \
            C.process_graph("hic", {"c": a_in}, n, "c", device=DEV)
        tgt = dev(z[name + "_tgt"])
        opt = torch.optim.SGD(m.parameters(), lr=0.25, weight_decay=1e-6, momentum=0.9)
        m.train()
        xf = dev(z[name + "_xf"]).requires_grad_(True); xr = dev(z[name + "_xr"]).requires_grad_(True)

        def step():
            opt.zero_grad()
            if batched:
                p, _ = m.forward_strands(torch.stack([xf, xr]), adj)
                pf, pr = p[0], p[1]
            else:
                _, pf, _, _ = m(xf, adj, None)
                _, pr, _, _ = m(xr, adj, None)
            loss = F.binary_cross_entropy_with_logits((pf + pr) / 2, tgt)
            loss.backward()
            return loss

        loss = step()
        assert abs(loss.item() - float(z[name + "_train_loss"])) < 1e-4
        np.testing.assert_allclose(xf.grad.cpu().numpy(), z[name + "_train_dxf"], atol=1e-4 * np.abs(z[name + "_train_dxf"]).max(), rtol=1e-4)
        np.testing.assert_allclose(xr.grad.cpu().numpy(), z[name + "_train_dxr"], atol=1e-4 * np.abs(z[name + "_train_dxr"]).max(), rtol=1e-4)
        for k, p in m.named_parameters():
            ref = z["%s_grad_%s" % (name, k)]
            # scale-relative 1e-4 for every gradient (tests/test_gpu_fullsize_oracle.py has the float64 analysis)
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=1e-4 * float(np.abs(ref).max()), rtol=1e-4, err_msg=k)
        opt.step()
        post = state_from(z, name + "_post")
        for k, v in m.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), post[k].numpy(), err_msg=k, **TOL)
        xf.grad = None; xr.grad = None
        loss2 = step(); opt.step()
        assert abs(loss2.item() - float(z[name + "_train_loss2"])) < 1e-4
        post2 = state_from(z, name + "_post2")
        for k, v in m.state_dict().items():
            tol = dict(atol=1e-5, rtol=1e-5) if "running" in k else TOL
            np.testing.assert_allclose(v.cpu().numpy(), post2[k].numpy(), err_msg=k, **tol)


def test_deeper_wider_model_against_oracle():
    """config 4 shape family: d = 256, 4 layers -- beyond the reference (its ctor caps at 2 layers);
    oracle = the restatement rule 'repeat ChromeModels.py:42-46'."""
    _deeper_wider_case(5)


def _deeper_wider_case(seed):
    n, d, L, c = 300, 256, 4, 11
    a = O.random_symmetric_graph(n, 2000, 9)
    # (every parameter from a seeded stream: the gate weights and the classifier keep their constructor values, which used
    # to come from whatever state the tests before had left the global generator in -- and with them the size of the gate-bias
    # gradients, single scalars that are sums of cancelling per-row terms)
    torch.manual_seed(seed)
    orc = O.GatedGCNOracle(d, c, 0.0, L)
    g_ = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in orc.named_parameters():
            if "GC" in k and "weight" in k:
                p.copy_(torch.randn(p.shape, generator=g_) / np.sqrt(d) * 1.5)
            elif p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=g_) * 0.2)
    m = C.ChromeGCN(d, d, c, 0.0, True, L)
    m.load_state_dict(orc.state_dict()); m.to(DEV)
    x = torch.randn(2, n, d, generator=g_)
    tgt = (torch.rand(n, c, generator=g_) < 0.1).float()
    adj_cpu = O.process_graph("hic", {"c": a}, n, "c")
    g = C.process_graph("hic", {"c": a}, n, "c", device=DEV)
    orc.train(); m.train()
    xo = x.clone().requires_grad_(True)
    threads = torch.get_num_threads()
    torch.set_num_threads(8)   # (the host oracle's fp32 sums depend on the thread count: tests/test_gpu_fullsize_oracle.py)
    try:
        lo = F.binary_cross_entropy_with_logits((orc(xo[0], adj_cpu)[1] + orc(xo[1], adj_cpu)[1]) / 2, tgt)
        lo.backward()
    finally:
        torch.set_num_threads(threads)
    xg = x.to(DEV).requires_grad_(True)
    p, gates = m.forward_strands(xg, g)
    lg = F.binary_cross_entropy_with_logits((p[0] + p[1]) / 2, tgt.to(DEV))
    lg.backward()
    assert len(gates) == 4 and abs(lo.item() - lg.item()) < 1e-4
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xo.grad.numpy(), atol=1e-4 * xo.grad.abs().max().item(), rtol=1e-4)
    po = dict(orc.named_parameters())
    # scale-relative 1e-4 per tensor; a gate-bias gradient is ONE number -- the sum of n x S signed per-row terms that cancel to a
    # fraction of a percent of their absolute sum (tests/test_gpu_fullsize_oracle.py) -- so its scale is the largest gate-bias
    # gradient of the model, not its own (possibly accidentally tiny) value: two fp32 summation orders of the same terms differ
    # by 1e-4 of a scalar that happens to come out 5x smaller than its siblings, and did (one box, round 6)
    kind_scale = {}
    for k, p_ in po.items():
        kind = k.split(".")[0].rstrip("0123456789") + "." + k.split(".", 1)[1]
        kind_scale[kind] = max(kind_scale.get(kind, 0.0), float(p_.grad.abs().max()))
    for k, pp in m.named_parameters():
        ref = po[k].grad.numpy()
        kind = k.split(".")[0].rstrip("0123456789") + "." + k.split(".", 1)[1]
        scale = kind_scale[kind] if ref.size == 1 else float(np.abs(ref).max())
        np.testing.assert_allclose(pp.grad.cpu().numpy(), ref, atol=1e-4 * scale, rtol=1e-4, err_msg=k)


def test_cpu_inputs_fail_loudly():
    m = C.ChromeGCN(128, 128, 5, 0.0, True, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(4, 128), None)


def test_bad_shapes_are_rejected():
    g = C.process_graph("none", None, 8, "c", device=DEV)
    with pytest.raises(RuntimeError):
        ops.spmm(torch.randn(1, 8, 102, device=DEV), g)      # the bare aggregation takes any width % 4 == 0 ...
    assert torch.allclose(ops.spmm(torch.ones(1, 8, 100, device=DEV), g), torch.ones(1, 8, 100, device=DEV))
    with pytest.raises(RuntimeError):
        ops.spmm(torch.randn(1, 9, 128, device=DEV), g)
    w = torch.randn(100, 100, device=DEV)
    with pytest.raises(RuntimeError):                         # ... the fused gated layer only 128 / 256
        ops.gated_layer(torch.randn(1, 8, 100, device=DEV), w, torch.zeros(100, device=DEV), torch.zeros(1, 100, device=DEV),
                        torch.zeros(1, device=DEV), g)


def test_views_at_odd_storage_offsets_are_accepted():
    """A contiguous view that starts 4 bytes into another tensor's storage is not 16-byte aligned; the kernels'
    vector row accesses need that alignment, so the Python layer copies such inputs once (the raw C ABI rejects them)."""
    n, d = 97, 128
    g = G.upload(G.normalize_graph("hic", O.random_symmetric_graph(n, 300, 2), n), DEV)
    base = torch.randn(n * d + 1, device=DEV)
    x_off = base[1:].view(1, n, d)
    assert x_off.data_ptr() % 16 != 0 and x_off.is_contiguous()
    x_ok = x_off.clone()
    torch.testing.assert_close(ops.spmm(x_off, g), ops.spmm(x_ok, g), rtol=0, atol=0)
    w = torch.randn(d, d, device=DEV) / d ** 0.5
    b, wg, cg = torch.zeros(d, device=DEV), torch.randn(1, d, device=DEV) / d ** 0.5, torch.zeros(1, device=DEV)
    y1, _ = ops.gated_layer(x_off, w, b, wg, cg, g)
    y2, _ = ops.gated_layer(x_ok, w, b, wg, cg, g)
    torch.testing.assert_close(y1, y2, rtol=0, atol=0)
