"""torch.library registration, CPU side: the operators exist under torch.ops.chromegcn with the documented schemas,
and their fake (meta) implementations propagate shapes without touching a device or the library."""
import torch
from torch._subclasses.fake_tensor import FakeTensorMode

from chromegcn_amd import torch_ops  # noqa: F401  (registers the operators)


def test_operators_are_registered_with_functional_schemas():
    ops = torch.ops.chromegcn
    for name in ("spmm", "gated_layer", "gated_layer_backward", "head_loss", "head_loss_backward", "head_logits", "sgd_step"):
        assert hasattr(ops, name), name
    assert "Tensor(a" not in str(ops.gated_layer.default._schema)       # functional: autograd formulas are registered
    assert "Tensor(a" not in str(ops.head_loss.default._schema)
    s = str(ops.sgd_step.default._schema)
    assert "Tensor(a0!) param" in s and "-> ()" in s                     # the optimizer step mutates, returns nothing


def test_fake_implementations_propagate_shapes():
    with FakeTensorMode():
        dev = "cuda"
        S, n, d, C, nnz = 2, 100, 128, 7, 500
        x = torch.empty(S, n, d, device=dev)
        rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
        col = torch.empty(nnz, dtype=torch.int32, device=dev)
        rs = torch.empty(n, device=dev)
        y = torch.ops.chromegcn.spmm(x, rp, col, None, rs, rp, col, None)
        assert tuple(y.shape) == (S, n, d)
        y2 = torch.ops.chromegcn.spmm(torch.empty(1, n, 36, device=dev), rp, col, None, rs, rp, col, None)
        assert tuple(y2.shape) == (1, n, 36)
        W, b = torch.empty(d, d, device=dev), torch.empty(d, device=dev)
        wg, cg = torch.empty(1, d, device=dev), torch.empty(1, device=dev)
        xn, g, z, h = torch.ops.chromegcn.gated_layer(x, W, b, wg, cg, rp, col, None, rs, rp, col, None, 0.0, 0.0, None, 1)
        assert tuple(xn.shape) == (S, n, d) and tuple(g.shape) == (S, n) and z.shape == h.shape == x.shape
        outs = torch.ops.chromegcn.gated_layer_backward(xn, None, x, z, h, g, W, wg, rp, col, None, rs, 0.0, None, 1, True)
        assert [tuple(o.shape) for o in outs] == [(S, n, d), (d, d), (d,), (d,), (1,), (S, n, d)]
        outs = torch.ops.chromegcn.gated_layer_backward(xn, None, x, z, h, g, W, wg, rp, col, None, rs, 0.0, None, 1, False)
        assert outs[0].numel() == 0
        bw, bb = torch.empty(d, device=dev), torch.empty(d, device=dev)
        Wo, bo = torch.empty(C, d, device=dev), torch.empty(C, device=dev)
        rm, rv = torch.empty(d, device=dev), torch.empty(d, device=dev)
        tgt = torch.empty(n, C, device=dev)
        assert tuple(torch.ops.chromegcn.head_logits(xn, bw, bb, rm, rv, 1e-5, Wo, bo).shape) == (S, n, C)
        for training in (True, False):
            loss, probs, sm, si, dp, nrm, nrv = torch.ops.chromegcn.head_loss(xn, bw, bb, Wo, bo, tgt, rm, rv, 0.1, 1e-5,
                                                                              training, 0.0, None)
            assert loss.dim() == 0 and tuple(probs.shape) == (n, C) and nrm.shape == rm.shape and nrv.shape == rv.shape
            assert (tuple(sm.shape) == (S, d) and tuple(dp.shape) == (n, C)) if training else (sm.numel() == 0 and dp.numel() == 0)
        outs = torch.ops.chromegcn.head_loss_backward(torch.empty((), device=dev), xn, bw, bb, Wo, dp, sm, si, 0.0, None)
        p = torch.empty(1000, device=dev)
        assert torch.ops.chromegcn.sgd_step(p, p.clone(), p.clone(), 0.1, 0.9, 1e-6, False, 1.0, None) is None
