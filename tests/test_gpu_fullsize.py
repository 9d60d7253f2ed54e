"""Full-size checks at BASELINE.json's shapes (chr21-like: 5 776 windows / 250 k contact pairs; chr1-like:
29 910 windows), where the CPU oracle is too slow to be the checker: size-independent properties of the
operators and agreement between the fused kernels and their unfused composition on the device."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import chromegcn_amd as C
from chromegcn_amd import graph as G, ops, synth
from chromegcn_amd.finetune import GCNStage

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [("chr21", synth.chrom_nodes("chr21"), 250000, False), ("chr1", synth.chrom_nodes("chr1"), 250000, True)]


@pytest.fixture(scope="module", params=SHAPES, ids=[s[0] for s in SHAPES])
def big(request):
    name, n, pairs, hic_like = request.param
    hic = synth.contact_graph(n, pairs, 5, hic_like)
    return name, n, hic, C.process_graph("hic", {"c": hic}, n, "c", device=DEV)


def test_rows_of_the_normalised_adjacency_sum_to_one(big):
    name, n, hic, g = big
    ones = torch.ones(2, n, 128, device=DEV)
    y = ops.spmm(ones, g)
    assert torch.allclose(y, ones, atol=2e-6)          # D^-1 (A + I) 1 = 1  (utils/util_methods.py:99-106)
    deg = (g.rowptr[1:] - g.rowptr[:-1]).float()
    assert torch.equal(g.row_scale, (1.0 / deg.double()).float()) and bool(g.symmetric)
    assert int(g.nnz) == int(hic.nnz) + n              # +I, no duplicates, nothing lost


def test_spmm_is_linear_and_its_backward_is_the_adjoint(big):
    name, n, hic, g = big
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(2, n, 128, device=DEV, generator=gen)
    y = torch.randn(2, n, 128, device=DEV, generator=gen)
    a, b = 0.7, -1.3
    lhs = ops.spmm(a * x + b * y, g)
    rhs = a * ops.spmm(x, g) + b * ops.spmm(y, g)
    assert torch.allclose(lhs, rhs, atol=1e-5, rtol=1e-5)
    xr = x.clone().requires_grad_(True)
    ops.spmm(xr, g).backward(y)                                  # xr.grad = A^T y
    dot1 = (ops.spmm(x, g).double() * y.double()).sum().item()   # <A x, y>
    dot2 = (x.double() * xr.grad.double()).sum().item()          # <x, A^T y>
    assert abs(dot1 - dot2) < 1e-6 * max(1.0, abs(dot1)) + 1e-3


def test_fused_layer_equals_unfused_composition(big):
    """cgcn_layer_fwd/bwd against  spmm kernel + torch GEMM/elementwise ops  in the REFERENCE's op order
    A (X W) + b (models/SubLayers.py:43-50, ChromeModels.py:38-40), forward and every gradient."""
    name, n, hic, g = big
    d = 128
    gen = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(2, n, d, device=DEV, generator=gen)
    W = (torch.randn(d, d, device=DEV, generator=gen) / d ** 0.5 * 1.5)
    b = torch.randn(d, device=DEV, generator=gen) * 0.2
    wg = torch.randn(1, d, device=DEV, generator=gen) / d ** 0.5 * 2
    cg = torch.randn(1, device=DEV, generator=gen) * 0.3
    gup = torch.randn(2, n, d, device=DEV, generator=gen) * 0.1
    t1 = [t.clone().requires_grad_(True) for t in (x, W, b, wg, cg)]
    xn, gate = ops.gated_layer(*t1, g)
    xn.backward(gup)
    t2 = [t.clone().requires_grad_(True) for t in (x, W, b, wg, cg)]
    u = ops.spmm(torch.matmul(t2[0], t2[1]), g) + t2[2]
    z = torch.tanh(u)
    gt = torch.sigmoid(F.linear(z, t2[3], t2[4]))
    ref = (1 - gt) * t2[0] + gt * z
    ref.backward(gup)
    assert torch.allclose(xn, ref, atol=1e-4, rtol=1e-4)
    assert torch.allclose(gate, gt.squeeze(-1), atol=1e-4, rtol=1e-4)
    for a_, b_, nm in zip(t1, t2, ["dX", "dW", "db", "dwg", "dcg"]):
        scale = max(1.0, b_.grad.abs().max().item())
        assert torch.allclose(a_.grad, b_.grad, atol=1e-4 * scale, rtol=1e-4), nm


def test_whole_step_is_bit_reproducible_and_matches_torch_head(big):
    """two independent engines replaying the captured step give identical bits; the fused head + sinks path
    agrees with the torch-head path (forward_strands + F.binary_cross_entropy_with_logits) to 1e-4."""
    name, n, hic, g = big
    feats = synth.chrom_features(n, 128, 103, 3)
    res = []
    for fused in (True, True, False):
        torch.manual_seed(0)
        m = C.ChromeGCN(128, 128, 103, 0.0, True, 2).to(DEV)
        with torch.no_grad():
            m.GC1.weight.mul_(40); m.GC2.weight.mul_(40)   # the reference init (gain 0.02) leaves z ~ 0
        opt = torch.optim.SGD(m.parameters(), lr=0.25, momentum=0.9, weight_decay=1e-6)
        st = GCNStage(m, opt, "hic", DEV, hip_graphs=True, fused_head=fused, input_grad=True)
        st.add_chromosome("c", feats, hic)
        for _ in range(3):
            loss, probs, dx = st.train_step("c")
        res.append((loss.clone(), probs.clone(), dx.clone(), {k: v.clone() for k, v in m.state_dict().items()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    for k in res[0][3]:
        assert torch.equal(res[0][3][k], res[1][3][k]), k
    assert abs(res[0][0].item() - res[2][0].item()) < 1e-4
    assert torch.allclose(res[0][1], res[2][1], atol=1e-4, rtol=1e-4)
    for k in res[0][3]:
        assert torch.allclose(res[0][3][k].float(), res[2][3][k].float(), atol=1e-4, rtol=1e-4), k
