"""cgcn_sgd_fuse: the optimizer step carried by the last backward launch of a train step (the first layer's gather
kernel) must equal cgcn_layer_bwd followed by cgcn_sgd_step bit for bit -- parameters, momentum, the dropout counter,
and dX (the gather multiplies by the OLD weight although the same launch writes the new one)."""
import ctypes

import numpy as np
import pytest
import torch

import chromegcn_amd as C
from chromegcn_amd import _lib, graph as G, synth
from chromegcn_amd.finetune import GCNStage
from oracle import chromegcn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
P = _lib.ptr


@pytest.mark.parametrize("want_dx", [True, False], ids=["gather_launch", "partial_sum_launch"])
@pytest.mark.parametrize("n,d,momentum,nesterov", [(777, 128, 0.9, False), (12000, 128, 0.9, True), (300, 256, 0.0, False)])
def test_fused_step_equals_separate_step_bitwise(n, d, momentum, nesterov, want_dx):
    lib = _lib.load()
    S = 2
    g = G.upload(G.normalize_graph("hic", synth.contact_graph(n, 6 * n, 3), n), DEV)
    gen = torch.Generator(device=DEV).manual_seed(n)
    rnd = lambda *sh: torch.randn(*sh, device=DEV, generator=gen)
    x, z, h = rnd(S, n, d), torch.tanh(rnd(S, n, d)), rnd(S, n, d)
    gate, dxn = torch.rand(S, n, device=DEV, generator=gen), rnd(S, n, d) * 0.1
    # a flat arena like the engine's: [other params (padding incl.)] [W d*d] [b d] [wg d] [cg 1 (+3 pad)] [more params]
    off_W, off_b, off_wg, off_cg, total = 1000, 1000 + d * d, 1000 + d * d + d, 1000 + d * d + 2 * d, 1000 + d * d + 2 * d + 4 + 2000
    param0 = rnd(total) * 0.1
    grad0 = rnd(total) * 0.01          # "gradients other launches finished"; the layer's own slots get overwritten
    mom0 = rnd(total) * 0.01
    rng0 = torch.tensor([7, 11], dtype=torch.int64, device=DEV)
    ws_bytes = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
    lr, wd, gs = 0.25, 1e-6, 0.5

    def run(fused):
        param, grad, mom, rng = param0.clone(), grad0.clone(), mom0.clone(), rng0.clone()
        dx, dhs = torch.empty_like(x), torch.empty_like(x)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
        W, wg = param[off_W:off_W + d * d], param[off_wg:off_wg + d]
        sg = _lib.SgdFuse(param.data_ptr(), grad.data_ptr(), mom.data_ptr() if momentum else None, total, lr, momentum, wd, gs,
                          1 if nesterov else 0, rng.data_ptr())
        rc = lib.cgcn_layer_bwd(_lib.stream_ptr(), n, S, d, P(g.rowptr_t), P(g.col_t), None, P(g.row_scale), P(x), P(z), P(h),
                                P(gate), W.data_ptr(), wg.data_ptr(), P(dxn), None, P(dx) if want_dx else None, P(dhs),
                                grad[off_W:].data_ptr(), grad[off_b:].data_ptr(), grad[off_wg:].data_ptr(), grad[off_cg:].data_ptr(),
                                0, 0.0, None, 0, None, P(ws), ws_bytes, None, ctypes.byref(sg) if fused else None, None)
        assert rc == 0
        if not fused:
            assert lib.cgcn_sgd_step(_lib.stream_ptr(), total, P(param), P(grad), P(mom) if momentum else None, lr, momentum, wd,
                                     1 if nesterov else 0, gs, P(rng)) == 0
        torch.cuda.synchronize()
        return param, grad, mom, rng, (dx if want_dx else dhs)
    a, b = run(True), run(False)
    for name, ta, tb in zip(("param", "grad", "momentum", "rng", "dX"), a, b):
        assert torch.equal(ta, tb), name
    assert not torch.equal(a[0], param0) and int(a[3][1]) == 12


def test_fuse_request_is_validated():
    lib = _lib.load()
    n, S, d = 64, 2, 128
    g = G.upload(G.normalize_graph("none", None, n), DEV)
    x = torch.randn(S, n, d, device=DEV)
    gate = torch.rand(S, n, device=DEV)
    arena_p, arena_g = torch.zeros(20000, device=DEV), torch.zeros(20000, device=DEV)
    elsewhere = torch.zeros(d * d, device=DEV)
    ws_bytes = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    dx, dhs = torch.empty_like(x), torch.empty_like(x)
    sg = _lib.SgdFuse(arena_p.data_ptr(), arena_g.data_ptr(), None, 20000, 0.1, 0.0, 0.0, 1.0, 0, None)

    def call(dX, dW, accumulate=0, sgd=sg):
        return lib.cgcn_layer_bwd(_lib.stream_ptr(), n, S, d, P(g.rowptr_t), P(g.col_t), None, P(g.row_scale), P(x), P(x), P(x),
                                  P(gate), arena_p[:d * d].data_ptr(), arena_p[17000:].data_ptr(), P(x), None, dX, P(dhs), dW,
                                  arena_g[16384:].data_ptr(), arena_g[16600:].data_ptr(), arena_g[16800:].data_ptr(), accumulate,
                                  0.0, None, 0, None, P(ws), ws_bytes, None, ctypes.byref(sgd) if sgd is not None else None, None)
    assert call(P(dx), arena_g.data_ptr()) == 0
    assert call(None, arena_g.data_ptr()) == 0                       # no gather launch: the partial-sum launch carries the step
    assert call(P(dx), elsewhere.data_ptr()) == -1                   # dW outside the gradient arena
    assert call(P(dx), arena_g.data_ptr(), accumulate=1) == -1       # the step needs final (overwritten) sums
    bad = _lib.SgdFuse(arena_p.data_ptr(), arena_g.data_ptr(), None, 20000, 0.1, 0.9, 0.0, 1.0, 0, None)
    assert call(P(dx), arena_g.data_ptr(), sgd=bad) == -1            # momentum without a momentum buffer
    torch.cuda.synchronize()


def test_engine_fuses_the_step_with_and_without_input_gradient():
    """GCNStage with d loss / d features requested (reference semantics) carries the step in the first layer's gather
    launch, without it in that layer's partial-sum launch: same parameters either way (to fp32 rounding: the two launches
    sum the partial blocks with 512- and 256-thread second stages, i.e. in different orders), and the oracle's."""
    n, d, c = 900, 128, 13
    feats = synth.chrom_features(n, d, c, 5)
    hic = synth.contact_graph(n, 7000, 5)
    outs = []
    for input_grad in (True, False):
        torch.manual_seed(0)
        orc = O.GatedGCNOracle(d, c, 0.0, 2)
        with torch.no_grad():
            orc.GC1.weight.mul_(40); orc.GC2.weight.mul_(40)
        m = C.ChromeGCN(d, d, c, 0.0, True, 2)
        m.load_state_dict(orc.state_dict())
        m.to(DEV)
        opt = torch.optim.SGD(m.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
        st = GCNStage(m, opt, "hic", DEV, hip_graphs=True, input_grad=input_grad, cache_input_aggregation=False)
        st.add_chromosome("c", feats, hic)
        for _ in range(3):
            loss, probs, dx = st.train_step("c")
        assert (dx is not None) == input_grad
        outs.append(({k: v.clone() for k, v in m.state_dict().items()}, loss.clone(), int(m._rng_state[1])))
    for k in outs[0][0]:
        torch.testing.assert_close(outs[0][0][k], outs[1][0][k], rtol=1e-5, atol=1e-6, msg=k)
    assert abs(outs[0][1].item() - outs[1][1].item()) < 1e-6 and outs[0][2] == outs[1][2] == 3
    oopt = O.make_sgd(orc, 0.25)
    cache = {}
    for _ in range(3):
        O.finetune_epoch(orc, {"c": feats}, {"c": hic}, oopt, "train", "hic", adj_cache=cache)
    for k, v in orc.state_dict().items():
        np.testing.assert_allclose(outs[0][0][k].cpu().numpy(), v.numpy(), atol=1e-4, rtol=1e-4, err_msg=k)
