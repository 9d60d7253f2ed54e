"""Fused inter-layer dropout (models/ChromeModels.py:42) and the fused SGD step
(utils/util_methods.py:14-19) on the GPU."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import chromegcn_amd as C
from chromegcn_amd import graph as G
from chromegcn_amd import ops
from oracle import chromegcn_oracle as O
from test_gpu_head import _probe_mask

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _probe_layer_mask(S, n, d, p, rng, layer_id):
    """mask of the layer kernel's output dropout: identity graph, W = 0, b = +30 -> z = 1; gate bias +30 ->
    g = 1 -> Xn = 1 before dropout, so the output IS mask / (1-p)."""
    g = C.process_graph("none", None, n, "c", device=DEV)
    W = torch.zeros(d, d, device=DEV); b = torch.full((d,), 30.0, device=DEV)
    wg = torch.zeros(1, d, device=DEV); cg = torch.full((1,), 30.0, device=DEV)
    xn, _ = ops.gated_layer(torch.zeros(S, n, d, device=DEV), W, b, wg, cg, g, dropout_out=p, rng_state=rng, layer_id=layer_id)
    return (xn > 0.5).cpu()


@pytest.mark.parametrize("scale", [1.0, 1.7])
def test_two_layer_model_with_dropout_matches_float64_with_explicit_masks(scale):
    S, n, d, c, p = 2, 97, 128, 9, 0.2
    a = O.random_symmetric_graph(n, 500, 3)
    graph = C.process_graph("hic", {"c": a}, n, "c", device=DEV)
    A64 = torch.from_numpy(O.normalized_adjacency("hic", a, n).toarray()).double()
    torch.manual_seed(4)
    m = C.ChromeGCN(d, d, c, p, True, 2)
    with torch.no_grad():
        for k, q in m.named_parameters():
            if "GC" in k and k.endswith("weight"):
                q.copy_(torch.randn_like(q) / np.sqrt(d) * 1.5)
            elif q.dim() == 1:
                q.copy_(torch.randn_like(q) * 0.2)
    m64 = copy.deepcopy(m).double()
    m = m.to(DEV).train()
    m._rng_managed = True          # this test pins the step counter itself
    m.seed_dropout(77)
    m._rng_state[1] = 3
    x = torch.randn(S, n, d)
    tgt = (torch.rand(n, c) < 0.2).float()
    mask1 = _probe_layer_mask(S, n, d, p, m._rng_state, 1).double()
    maskh = _probe_mask(S * n, d, p, 77, 3).view(S, n, d).double()
    assert 0.7 < mask1.mean().item() < 0.9 and not torch.equal(mask1, maskh)

    # float64 restatement of ChromeModels.py:34-52 + finetune.py:43-45 with the masks made explicit
    x64 = x.double().requires_grad_(True)
    logits = []
    m64.train()
    for s in range(S):
        h = x64[s]
        for k in (1, 2):
            gc, wk = getattr(m64, "GC%d" % k), getattr(m64, "W%d" % k)
            if k == 2:
                h = h * mask1[s] / (1 - p)
            z = torch.tanh(A64 @ (h @ gc.weight) + gc.bias)
            g = torch.sigmoid(wk(z))
            h = (1 - g) * h + g * z
        y = m64.batch_norm(F.relu(h)) * maskh[s] / (1 - p)
        logits.append(m64.out(y))
    loss64 = F.binary_cross_entropy_with_logits(sum(logits) / S, tgt.double())
    (loss64 * scale).backward()   # scale != 1: the fused head computes its backward half for d loss = 1 and rescales

    xg = x.to(DEV).requires_grad_(True)
    loss, probs, gates = m.forward_loss(xg, graph, tgt.to(DEV))
    (loss * scale).backward()
    assert int(m._rng_state[1].item()) == 3
    assert abs(loss.item() - loss64.item()) < 2e-5
    ref = x64.grad.numpy()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), ref, atol=1e-4 * np.abs(ref).max(), rtol=1e-4)
    p64 = dict(m64.named_parameters())
    for k, q in m.named_parameters():
        r = p64[k].grad.numpy()
        np.testing.assert_allclose(q.grad.cpu().numpy(), r, atol=1e-4 * max(1e-6, np.abs(r).max()), rtol=1e-4, err_msg=k)


def test_standalone_forward_draws_fresh_masks_and_backward_still_matches():
    """outside the engine every forward snapshots + advances the counter: two calls differ, and each
    call's backward uses the mask of its own forward (the f / r call pattern of finetune.py:41-42)."""
    n, d, c, p = 61, 128, 5, 0.5
    graph = C.process_graph("constant", None, n, "c", device=DEV)
    torch.manual_seed(1)
    m = C.ChromeGCN(d, d, c, p, True, 2).to(DEV).train()
    with torch.no_grad():
        m.GC1.weight.copy_(torch.randn(d, d, device=DEV) / np.sqrt(d))
        m.GC2.weight.copy_(torch.randn(d, d, device=DEV) / np.sqrt(d))
    x = torch.randn(n, d, device=DEV)
    torch.manual_seed(5)
    c0 = int(m._rng_state[1].item())
    xa = x.clone().requires_grad_(True); xb = x.clone().requires_grad_(True)
    torch.manual_seed(9); _, oa, _, _ = m(xa, graph, None)
    torch.manual_seed(9); _, ob, _, _ = m(xb, graph, None)   # same torch RNG for the head dropout, new layer mask
    assert int(m._rng_state[1].item()) == c0 + 2
    assert not torch.allclose(oa, ob)
    (oa.sum() + ob.sum()).backward()
    # d(out)/dx is zero exactly where the first call's inter-layer mask AND ... -> simply: the two grads differ
    assert not torch.allclose(xa.grad, xb.grad)
    m.eval()
    with torch.no_grad():
        _, e1, _, _ = m(x, graph, None); _, e2, _, _ = m(x, graph, None)
    assert torch.equal(e1, e2) and int(m._rng_state[1].item()) == c0 + 2


@pytest.mark.parametrize("momentum,wd,nesterov", [(0.9, 1e-6, False), (0.0, 0.0, False), (0.8, 1e-3, True)])
def test_fused_sgd_matches_torch(momentum, wd, nesterov):
    torch.manual_seed(0)
    n = 46825
    p_ref = torch.randn(n, device=DEV); p = p_ref.clone()
    par = nn.Parameter(p_ref)
    opt = torch.optim.SGD([par], lr=0.25, momentum=momentum, weight_decay=wd, nesterov=nesterov)
    mom = torch.zeros(n, device=DEV)
    rng = torch.tensor([1, 10], dtype=torch.int64, device=DEV)
    for step in range(3):
        g = torch.randn(n, device=DEV)
        par.grad = g.clone()
        opt.step()
        ops.sgd_step(p, g * 4.0, mom if momentum else None, 0.25, momentum, wd, nesterov, rng, grad_scale=0.25)
        np.testing.assert_allclose(p.cpu().numpy(), par.detach().cpu().numpy(), atol=1e-6, rtol=1e-6)
    assert int(rng[1].item()) == 13
