"""The C-ABI library loads and exports every symbol include/chromegcn.h declares (no GPU needed,
no compute calls)."""
import os
import re

from chromegcn_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "chromegcn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cgcn_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_symbols() == _lib.exported_symbols()


def test_library_builds_loads_and_exports_everything():
    _build.build_library()
    lib = _lib.load()
    for name in declared_symbols():
        assert getattr(lib, name) is not None
    assert lib.cgcn_abi_version() == _lib.ABI_VERSION
    assert lib.cgcn_strerror(0) == b"ok"
    assert b"unsupported" in lib.cgcn_strerror(-2)
    # pure host-side argument logic (no launch): workspace sizing and shape rejection
    assert lib.cgcn_layer_bwd_workspace_bytes(5000, 2, 128) > 0
    assert lib.cgcn_layer_bwd_workspace_bytes(5000, 3, 128) == 0
    assert lib.cgcn_layer_bwd_workspace_bytes(5000, 2, 100) == 0
    assert lib.cgcn_spmm(None, 10, 10, 1, 102, None, None, None, None, None, None, None) == -2   # width not a multiple of 4
    assert lib.cgcn_spmm(None, 10, 10, 3, 128, None, None, None, None, None, None, None) == -2   # strands
    assert lib.cgcn_spmm(None, 10, 10, 1, 100, None, None, None, None, None, None, None) == -1   # any width % 4: null pointers
    assert lib.cgcn_spmm(None, 10, 10, 1, 128, None, None, None, None, None, None, None) == -1


def test_route_queries_are_pure_host_logic():
    """cgcn_debug_layer_{fwd,bwd}_route (ABI v19): which kernels a call would launch; nothing is launched, no GPU needed."""
    import ctypes
    lib = _lib.load()
    # forward: fused below the split threshold (6 MiB of feature table), two launches above, or for a hub-heavy graph
    assert lib.cgcn_debug_layer_fwd_route(5776, 2, 128, None, 0) == 0       # 5.9 MB
    assert lib.cgcn_debug_layer_fwd_route(29910, 2, 128, None, 0) == 1      # 30.6 MB
    assert lib.cgcn_debug_layer_fwd_route(5776, 2, 256, None, 0) == 1       # d = 256, both strands (11.8 MB): the same threshold since round 6
    assert lib.cgcn_debug_layer_fwd_route(3000, 2, 256, None, 0) == 0 and lib.cgcn_debug_layer_fwd_route(29910, 2, 256, None, 0) == 1
    assert lib.cgcn_debug_layer_fwd_route(5776, 2, 128, None, 8) == 0 and lib.cgcn_debug_layer_fwd_route(5776, 2, 128, None, 24) == 1   # merged records
    assert lib.cgcn_debug_layer_fwd_route(5776, 2, 128, None, -1) == 1      # accumulate mode: the two-launch route at every size

    class Aux(ctypes.Structure):
        _fields_ = [("col16", ctypes.c_void_p), ("row_order", ctypes.c_void_p), ("max_row_len", ctypes.c_int32)]
    hub = Aux(None, None, 9000)
    assert lib.cgcn_debug_layer_fwd_route(5776, 2, 128, ctypes.byref(hub), 0) == 1
    lib.cgcn_debug_set_fwd_split_bytes(0)
    try:
        assert lib.cgcn_debug_layer_fwd_route(64, 1, 128, None, 0) == 1     # the test hook forces the split at every size
    finally:
        lib.cgcn_debug_set_fwd_split_bytes(-1)
    # backward: the ring kernel at d = 128, the column-slab kernel (k_bwd_rowlocal256s) at d = 256
    assert lib.cgcn_debug_layer_bwd_route(5776, 2, 128) == 2 and lib.cgcn_debug_layer_bwd_route(29910, 1, 128) == 2
    assert lib.cgcn_debug_layer_bwd_route(5776, 2, 256) == 0
    assert lib.cgcn_debug_layer_bwd_route(5776, 3, 128) == -2 and lib.cgcn_debug_layer_fwd_route(5776, 2, 100, None, 0) == -2
    # one partial record per workgroup of the row-local launch: at most one per CU, 16-row slots at d = 128
    rec = (128 * 128 + 2 * 128 + 4) * 4
    assert lib.cgcn_layer_bwd_workspace_bytes(5776, 2, 128) == 256 * rec
    assert lib.cgcn_layer_bwd_workspace_bytes(40, 2, 128) == 5 * rec
