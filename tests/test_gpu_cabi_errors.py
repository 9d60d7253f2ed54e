"""Error behaviour of the C ABI (include/chromegcn.h): every entry point returns a negative code instead of
launching on bad input, names it through cgcn_strerror, and treats empty inputs as a no-op."""
import ctypes

import numpy as np
import pytest
import torch

from chromegcn_amd import _lib, graph as G, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
OK, BAD_ARG, UNSUPPORTED, LAUNCH, WORKSPACE = 0, -1, -2, -3, -4


@pytest.fixture(scope="module")
def env():
    lib = _lib.load()
    n, S, d = 70, 2, 128
    g = G.upload(G.normalize_graph("hic", synth.contact_graph(n, 300, 3), n), DEV)
    t = lambda *shape: torch.randn(*shape, device=DEV)
    return dict(lib=lib, n=n, S=S, d=d, g=g, x=t(S, n, d), y=t(S, n, d), W=t(d, d), b=t(d), wg=t(d), cg=t(1),
                gate=t(S, n), z=t(S, n, d), h=t(S, n, d))


def fwd(e, **over):
    a = dict(e); a.update(over)
    P, g = _lib.ptr, a["g"]
    return a["lib"].cgcn_layer_fwd(_lib.stream_ptr(), a["n"], a["S"], a["d"], P(g.rowptr), P(g.col), None, P(g.row_scale),
                                   a.get("xptr", P(a["x"])), P(a["W"]), P(a["b"]), P(a["wg"]), P(a["cg"]), a.get("yptr", P(a["y"])),
                                   P(a["z"]), P(a["h"]), P(a["gate"]), a.get("p", 0.0), None, 0, None, None, 0, None)


def test_strerror_names_every_code(env):
    lib = env["lib"]
    msgs = {c: lib.cgcn_strerror(c).decode() for c in (OK, BAD_ARG, UNSUPPORTED, LAUNCH, WORKSPACE, -99)}
    assert len(set(msgs.values())) == 6 and all(msgs.values())
    assert lib.cgcn_abi_version() == _lib.ABI_VERSION


def test_layer_fwd_rejects_bad_input_without_launching(env):
    assert fwd(env) == OK
    before = env["y"].clone()
    assert fwd(env, d=64) == UNSUPPORTED                      # d not in {128, 256}
    assert fwd(env, S=3) == UNSUPPORTED
    assert fwd(env, n=-1) == BAD_ARG
    assert fwd(env, xptr=None) == BAD_ARG                     # null feature pointer
    assert fwd(env, yptr=_lib.ptr(env["x"])) == BAD_ARG       # output aliases the input
    assert fwd(env, xptr=_lib.ptr(env["x"]) + 4) == BAD_ARG   # not 16-byte aligned
    assert fwd(env, p=0.3) == BAD_ARG                         # dropout without an rng state
    assert fwd(env, n=0) == OK                                # empty chromosome: nothing to do
    torch.cuda.synchronize()
    assert torch.equal(env["y"], before)


def test_workspaces_are_checked(env):
    lib, n, S, d, g = env["lib"], env["n"], env["S"], env["d"], env["g"]
    P = _lib.ptr
    need = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
    assert need > 0 and lib.cgcn_layer_bwd_workspace_bytes(n, S, 100) == 0
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    dx, dhs = torch.empty_like(env["x"]), torch.empty_like(env["x"])
    dW, db, dwg, dcg = torch.empty(d, d, device=DEV), torch.empty(d, device=DEV), torch.empty(d, device=DEV), torch.empty(1, device=DEV)

    def bwd(ws_bytes, dxn=env["y"], wsp=P(ws)):
        return lib.cgcn_layer_bwd(_lib.stream_ptr(), n, S, d, P(g.rowptr_t), P(g.col_t), None, P(g.row_scale), P(env["x"]), P(env["z"]),
                                  P(env["h"]), P(env["gate"]), P(env["W"]), P(env["wg"]), _lib.ptr(dxn), None, P(dx), P(dhs), P(dW), P(db),
                                  P(dwg), P(dcg), 0, 0.0, None, 0, None, wsp, ws_bytes, None, None, None)
    assert bwd(need) == OK
    assert bwd(need - 1) == WORKSPACE
    assert bwd(need, wsp=None) == WORKSPACE
    assert bwd(need, dxn=None) == BAD_ARG                     # neither dXn nor a head state: no source of the gradient
    C = 11
    assert lib.cgcn_head_workspace_bytes(n, S, d, C) > 0
    assert lib.cgcn_head_workspace_bytes(n, S, d, 0) == 0 and lib.cgcn_head_workspace_bytes(n, S, d, 257) == 0
    assert lib.cgcn_metrics_workspace_bytes(10, 0) == 0 and lib.cgcn_metrics_workspace_bytes(10, 3) > 0
    rows = ctypes.c_int(0)
    assert lib.cgcn_layer_fwd_colstats_plan(n, S, d, _lib.COLSTATS_RECORDS, ctypes.byref(rows)) == (n + rows.value - 1) // rows.value
    assert lib.cgcn_layer_fwd_colstats_plan(n, S, 64, _lib.COLSTATS_RECORDS, ctypes.byref(rows)) == 0
    assert lib.cgcn_layer_fwd_colstats_plan(n, S, d, 7, ctypes.byref(rows)) == 0                       # no such mode
    tiles = lib.cgcn_layer_fwd_colstats_plan(n, S, d, _lib.COLSTATS_ACCUMULATE, ctypes.byref(rows))    # any table size, d = 128 / 256
    assert rows.value == -1 and tiles * S * d * 2 * 4 >= (8 * S * d * 2 + 1) * 8
    assert lib.cgcn_layer_fwd_colstats_plan(1, S, d, _lib.COLSTATS_ACCUMULATE, ctypes.byref(rows)) == 1 and rows.value > 0   # n < 2: records


def test_sgd_and_spmm_argument_checks(env):
    lib = env["lib"]
    P = _lib.ptr
    p, g_, m = torch.zeros(8, device=DEV), torch.ones(8, device=DEV), torch.zeros(8, device=DEV)
    assert lib.cgcn_sgd_step(_lib.stream_ptr(), 8, P(p), P(g_), P(m), 0.1, 0.9, 0.0, 0, 1.0, None) == OK
    assert lib.cgcn_sgd_step(_lib.stream_ptr(), 8, P(p), P(g_), None, 0.1, 0.9, 0.0, 0, 1.0, None) == BAD_ARG   # momentum without a buffer
    assert lib.cgcn_sgd_step(_lib.stream_ptr(), 8, P(p), P(g_), None, 0.1, 0.0, 0.0, 1, 1.0, None) == BAD_ARG   # nesterov without momentum
    assert lib.cgcn_sgd_step(_lib.stream_ptr(), -1, P(p), P(g_), P(m), 0.1, 0.9, 0.0, 0, 1.0, None) == UNSUPPORTED
    assert lib.cgcn_sgd_step(_lib.stream_ptr(), 0, None, None, None, 0.1, 0.0, 0.0, 0, 1.0, None) == OK
    torch.cuda.synchronize()
    assert torch.allclose(p, torch.full((8,), -0.1, device=DEV))
    g = env["g"]
    y = torch.empty_like(env["x"])
    assert lib.cgcn_spmm(_lib.stream_ptr(), env["n"], env["n"], env["S"], env["d"], P(g.rowptr), P(g.col), None, P(g.row_scale), P(env["x"]), P(y), None) == OK
    assert lib.cgcn_spmm(_lib.stream_ptr(), env["n"], env["n"], env["S"], env["d"], None, P(g.col), None, P(g.row_scale), P(env["x"]), P(y), None) == BAD_ARG
    assert lib.cgcn_spmm(_lib.stream_ptr(), env["n"], env["n"], env["S"], 98, P(g.rowptr), P(g.col), None, P(g.row_scale), P(env["x"]), P(y), None) == UNSUPPORTED


@pytest.fixture(params=["accumulate", "records"])
def mode(request):
    return request.param


def test_colstats_on_a_split_size_table_needs_an_aggregation_buffer(env, mode):
    """the plan reports merged records (or integer totals) for tables that take the two-launch route; the fused
    kernel (no H, no H_in) would write one record per 16/S-node tile -- past the caller's buffer.  Rejected in both modes, and
    -- the mode being an ARGUMENT since ABI v24 -- whatever the process-wide route threshold says at the time of the call;
    nothing is written past what was planned."""
    lib = env["lib"]
    P = _lib.ptr
    n, S, d = 8192, 2, 128                      # 8 MiB table: split-size
    rows = ctypes.c_int(0)
    tiles = lib.cgcn_layer_fwd_colstats_plan(n, S, d, _lib.COLSTATS_ACCUMULATE if mode == "accumulate" else _lib.COLSTATS_RECORDS, ctypes.byref(rows))
    if mode == "records":
        assert rows.value > 16 // S and tiles == (n + rows.value - 1) // rows.value
    else:                                       # accumulate mode: the buffer holds the integer totals
        assert rows.value == -1 and tiles * S * d * 2 * 4 >= (8 * S * d * 2 + 1) * 8
    g = G.upload(G.normalize_graph("hic", synth.contact_graph(n, 20000, 5), n), DEV)
    x, y, z, h = (torch.randn(S, n, d, device=DEV) for _ in range(4))
    gate = torch.empty(S, n, device=DEV)
    cs = torch.full((tiles * S * d * 2 + 4096,), 7.0, device=DEV)   # guard zone behind the records
    def call(hptr, r=None):
        return lib.cgcn_layer_fwd(_lib.stream_ptr(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(x), P(env["W"]),
                                  P(env["b"]), P(env["wg"]), P(env["cg"]), P(y), P(z), hptr, P(gate), 0.0, None, 0, None, P(cs),
                                  rows.value if r is None else r, None)
    assert call(None) == BAD_ARG
    assert call(P(h), 0) == BAD_ARG and call(P(h), 3) == BAD_ARG and call(P(h), -4) == BAD_ARG   # not a plan's value
    for split_bytes in (-1, 0, 1 << 40):        # the threshold moved between plan and call: same records, no overrun
        lib.cgcn_debug_set_fwd_split_bytes(split_bytes)
        try:
            assert call(None) == BAD_ARG
            cs.fill_(7.0)
            assert call(P(h)) == OK
            torch.cuda.synchronize()
            assert bool((cs[tiles * S * d * 2:] == 7.0).all())             # nothing written past the planned records
            assert not bool((cs[:S * d * 2] == 7.0).all())                 # ... and something inside them
        finally:
            lib.cgcn_debug_set_fwd_split_bytes(-1)


def test_zero_only_and_prezeroed_accumulate_modes(env):
    """colstats_rows = -2: the call produces no statistics and its first launch zeroes the totals for a LATER call;
    colstats_rows = -3: accumulate into such totals on ANY route -- here the fused one-launch forward of a small table, without an
    H buffer -- and the decoded totals are the sums of relu(Xn) (and equal the two-launch route's, which zeroes for itself: -1)."""
    lib = env["lib"]
    P = _lib.ptr
    n, S, d = 3001, 2, 128                       # 3 MB table: the fused route
    rows = ctypes.c_int(0)
    tiles = lib.cgcn_layer_fwd_colstats_plan(n, S, d, _lib.COLSTATS_ACCUMULATE, ctypes.byref(rows))
    assert rows.value == -1
    g = G.upload(G.normalize_graph("hic", synth.contact_graph(n, 9000, 5), n), DEV)
    x = torch.randn(S, n, d, device=DEV)
    y, z, h = (torch.empty_like(x) for _ in range(3))
    gate = torch.empty(S, n, device=DEV)
    words = 8 * S * d * 2 * 2 + 6                # forward block + backward block (cgcn_common.hpp)
    def fresh():
        return torch.full((tiles * S * d * 2 + 1024,), float("nan"), device=DEV)
    def call(cs, r, hptr=None, hin=None, xin=x):
        return lib.cgcn_layer_fwd(_lib.stream_ptr(), n, S, d, P(g.rowptr), P(g.col), None, P(g.row_scale), P(xin), P(env["W"]),
                                  P(env["b"]), P(env["wg"]), P(env["cg"]), P(y), P(z), hptr, P(gate), 0.0, None, 0, hin, P(cs), r, None)
    def decode(cs):
        t = cs[:8 * S * d * 2 * 2].view(torch.int64).view(8, S, d, 2).sum(0).double() / 2.0 ** 32
        return t[..., 0], t[..., 1]
    r = torch.relu
    for zero_route in ("fused", "two_launch", "h_in"):
        cs = fresh()
        if zero_route == "fused":
            assert call(cs, _lib.COLSTATS_ROWS_ZERO_ONLY) == OK
        elif zero_route == "two_launch":
            lib.cgcn_debug_set_fwd_split_bytes(0)
            try:
                assert call(cs, _lib.COLSTATS_ROWS_ZERO_ONLY, hptr=P(h)) == OK
            finally:
                lib.cgcn_debug_set_fwd_split_bytes(-1)
        else:
            hin = torch.randn_like(x)
            assert call(cs, _lib.COLSTATS_ROWS_ZERO_ONLY, hin=P(hin)) == OK
        torch.cuda.synchronize()
        assert bool((cs[:words * 2].view(torch.int32) == 0).all()), zero_route     # zeroed: both blocks and the header words
        assert bool(torch.isnan(cs[words * 2:]).all())                                # ... and nothing behind them
        assert bool(torch.isfinite(y).all())
    # accumulate into the zeroed totals on the fused route (no H): sums of relu(Xn) over the nodes
    assert call(cs, _lib.COLSTATS_ROWS_ACCUMULATE_ZEROED) == OK
    torch.cuda.synchronize()
    s1, s2 = decode(cs)
    want1, want2 = r(y).double().sum(1), (r(y).double() ** 2).sum(1)
    np.testing.assert_allclose(s1.cpu().numpy(), want1.cpu().numpy(), rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(s2.cpu().numpy(), want2.cpu().numpy(), rtol=1e-6, atol=1e-5)
    assert bool(torch.isnan(cs[words * 2:]).all())
    # the two-launch route in its own accumulate mode (-1: zeroes for itself) arrives at the same totals
    cs2 = fresh()
    assert call(cs2, _lib.COLSTATS_ROWS_ACCUMULATE) == BAD_ARG           # needs an H buffer
    assert call(cs2, _lib.COLSTATS_ROWS_ACCUMULATE, hptr=P(h)) == OK
    torch.cuda.synchronize()
    t1, t2 = decode(cs2)
    np.testing.assert_allclose(t1.cpu().numpy(), s1.cpu().numpy(), rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(t2.cpu().numpy(), s2.cpu().numpy(), rtol=1e-6, atol=1e-5)
    assert call(cs2, -4) == BAD_ARG


def test_fused_sgd_rejects_input_dropout_with_an_input_gradient(env):
    """the launch carrying cgcn_sgd_fuse advances rng_state[1] while its gather workgroups would read it."""
    lib, n, S, d, g = env["lib"], env["n"], env["S"], env["d"], env["g"]
    P = _lib.ptr
    need = lib.cgcn_layer_bwd_workspace_bytes(n, S, d)
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    dx, dhs = torch.empty_like(env["x"]), torch.empty_like(env["x"])
    cnt = d * d + 2 * d + 4
    param, grad, mom = torch.zeros(cnt, device=DEV), torch.zeros(cnt, device=DEV), torch.zeros(cnt, device=DEV)
    rng = torch.tensor([5, 0], dtype=torch.int64, device=DEV)
    sg = _lib.SgdFuse(param.data_ptr(), grad.data_ptr(), mom.data_ptr(), cnt, 0.1, 0.9, 0.0, 1.0, 0, rng.data_ptr())
    dW, db, dwg, dcg = grad[:d * d], grad[d * d:d * d + d], grad[d * d + d:d * d + 2 * d], grad[d * d + 2 * d:d * d + 2 * d + 1]

    def bwd(p_in, dxp):
        return lib.cgcn_layer_bwd(_lib.stream_ptr(), n, S, d, P(g.rowptr_t), P(g.col_t), None, P(g.row_scale), P(env["x"]), P(env["z"]),
                                  P(env["h"]), P(env["gate"]), P(env["W"]), P(env["wg"]), P(env["y"]), None, dxp, P(dhs), P(dW), P(db),
                                  P(dwg), P(dcg), 0, p_in, P(rng), 1, None, P(ws), need, None, ctypes.byref(sg), None)
    assert bwd(0.25, P(dx)) == BAD_ARG
    torch.cuda.synchronize()
    assert int(rng[1].item()) == 0
    assert bwd(0.0, P(dx)) == OK
    torch.cuda.synchronize()
    assert int(rng[1].item()) == 1


def test_head_logits_argument_checks(env):
    """cgcn_head_logits: unsupported widths / label counts and NULL pointers come back as error codes, n = 0 launches nothing."""
    lib, P, st = env["lib"], _lib.ptr, _lib.stream_ptr
    n, d, C = 64, 128, 7
    x = torch.randn(1, n, d, device="cuda")
    v = torch.ones(d, device="cuda")
    W = torch.randn(C, d, device="cuda"); b = torch.zeros(C, device="cuda")
    out = torch.full((1, n, C), 7.0, device="cuda")
    ok = lambda **k: lib.cgcn_head_logits(st(), k.get("n", n), k.get("S", 1), k.get("d", d), k.get("C", C), P(x), P(v), P(v), P(v),
                                         k.get("rv", P(v)), 1e-5, P(W), P(b), k.get("out", P(out)))
    assert ok() == OK
    torch.cuda.synchronize()
    want = (torch.relu(x[0]) - 1) / (1 + 1e-5) ** 0.5 + 1   # BatchNorm with mean = var = weight = bias = 1
    assert torch.allclose(out[0], want @ W.t() + b, atol=1e-4, rtol=1e-4)
    assert ok(d=96) == UNSUPPORTED and ok(C=257) == UNSUPPORTED and ok(S=3) == UNSUPPORTED
    assert ok(rv=None) == BAD_ARG and ok(out=None) == BAD_ARG and ok(n=-1) == BAD_ARG
    out.fill_(7.0)
    assert ok(n=0) == OK
    torch.cuda.synchronize()
    assert bool((out == 7.0).all())
