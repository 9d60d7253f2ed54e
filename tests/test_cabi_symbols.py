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
