"""ctypes binding of libchromegcn_hip.so -- the C ABI in include/chromegcn.h.

There is NO fallback: if the library is missing or a call fails, this raises.  torch is imported
first so that the HIP runtime already mapped by torch (soname libamdhip64.so.7) is the one the
library binds to; streams and device pointers are then interchangeable."""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL: shares torch's HIP runtime)

from . import _build

_c_int = ctypes.c_int
_c_vp = ctypes.c_void_p
_c_sz = ctypes.c_size_t
_c_float = ctypes.c_float
_c_uint = ctypes.c_uint

_SIGNATURES = {
    # name: (restype, argtypes)
    "cgcn_abi_version": (_c_int, []),
    "cgcn_strerror": (ctypes.c_char_p, [_c_int]),
    "cgcn_spmm": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "cgcn_layer_fwd": (_c_int, [_c_vp, _c_int, _c_int, _c_int] + [_c_vp] * 13 + [_c_float, _c_vp, _c_uint, _c_vp, _c_vp, _c_int, _c_vp]),
    "cgcn_layer_fwd_colstats_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_vp]),
    "cgcn_debug_set_fwd_split_bytes": (None, [ctypes.c_longlong]),
    "cgcn_debug_set_products": (None, [ctypes.c_int]),
    "cgcn_debug_get_products": (ctypes.c_int, []),
    "cgcn_debug_layer_fwd_route": (_c_int, [_c_int, _c_int, _c_int, _c_vp, _c_int]),
    "cgcn_debug_layer_bwd_route": (_c_int, [_c_int, _c_int, _c_int]),
    "cgcn_layer_bwd_workspace_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "cgcn_layer_bwd": (_c_int, [_c_vp, _c_int, _c_int, _c_int] + [_c_vp] * 18 + [_c_int, _c_float, _c_vp, _c_uint, _c_vp, _c_vp, _c_sz, _c_vp, _c_vp, _c_vp]),
    "cgcn_debug_layer_bwd_phases": (_c_int, [_c_vp, _c_int, _c_int, _c_int] + [_c_vp] * 18 + [_c_int, _c_float, _c_vp, _c_uint, _c_vp, _c_vp, _c_sz, _c_int, _c_vp]),
    "cgcn_head_workspace_bytes": (_c_sz, [_c_int] * 4),
    "cgcn_head_workspace_layout": (_c_int, [_c_int] * 4 + [ctypes.POINTER(_c_sz)] * 3),
    "cgcn_head_bwd_partials": (_c_int, [_c_int]),
    "cgcn_head_fwd": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp] * 6 + [_c_float, _c_float, _c_int] + [_c_vp] * 3
                      + [_c_float] + [_c_vp] * 7 + [_c_sz]),
    "cgcn_head_logits": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp] * 5 + [_c_float] + [_c_vp] * 3),
    "cgcn_head_train": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp] * 6 + [_c_float, _c_float] + [_c_vp] * 3 + [_c_float]
                        + [_c_vp] * 6 + [_c_int, _c_int, _c_vp, _c_sz]),
    "cgcn_debug_head_train_phases": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp] * 6 + [_c_float, _c_float] + [_c_vp] * 3 + [_c_float]
                                     + [_c_vp] * 6 + [_c_int, _c_int, _c_vp, _c_sz, _c_int]),
    "cgcn_head_bwd": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp] * 8 + [_c_float] + [_c_vp] * 6 + [_c_int, _c_vp, _c_sz]),
    "cgcn_sddmm": (_c_int, [_c_vp, _c_int, _c_int, _c_int] + [_c_vp] * 5 + [_c_int]),
    "cgcn_saliency_normalize": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp]),
    "cgcn_graph_count": (_c_int, [_c_vp, _c_int, _c_int] + [_c_vp] * 5),
    "cgcn_graph_fill": (_c_int, [_c_vp, _c_int, _c_int] + [_c_vp] * 8),
    "cgcn_metrics_workspace_bytes": (_c_sz, [ctypes.c_longlong, _c_int]),
    "cgcn_multilabel_metrics": (_c_int, [_c_vp, ctypes.c_longlong, _c_int, _c_vp, _c_vp, _c_float, _c_vp, _c_vp, _c_sz]),
    "cgcn_multilabel_metrics_nonneg": (_c_int, [_c_vp, ctypes.c_longlong, _c_int, _c_vp, _c_vp, _c_float, _c_vp, _c_vp, _c_vp, _c_sz]),
    "cgcn_sgd_step": (_c_int, [_c_vp, ctypes.c_longlong, _c_vp, _c_vp, _c_vp, _c_float, _c_float, _c_float, _c_int, _c_float, _c_vp]),
}
ABI_VERSION = 25
COLSTATS_RECORDS, COLSTATS_ACCUMULATE = 0, 1   # include/chromegcn.h: CGCN_COLSTATS_*
COLSTATS_ROWS_ACCUMULATE, COLSTATS_ROWS_ZERO_ONLY, COLSTATS_ROWS_ACCUMULATE_ZEROED = -1, -2, -3   # CGCN_COLSTATS_ROWS_*
_lib = None


class ChromeGCNLibraryError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGNATURES)


def load(build_if_missing=False):
    """Load (once) and return the ctypes handle.  Raises ChromeGCNLibraryError if the library is missing or was
    built from different sources than the tree holds.  Never compiles: building is explicit
    (`python -m chromegcn_amd._build` / `__graft_entry__.build()`), because this may run in a process that has
    already initialised the GPU, under a profiler, or as one of N ranks (chromegcn_amd/_build.py).
    `build_if_missing` is accepted for old callers and ignored."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("CHROMEGCN_LIB") or _build.LIB  # override: tuning experiments load a variant build
    if not os.path.exists(path):
        raise ChromeGCNLibraryError(
            "chromegcn_amd: %s is missing -- build it first: python -m chromegcn_amd._build (or "
            "__graft_entry__.build()).  There is no CPU/torch fallback for the HIP path." % path)
    if path == _build.LIB and _build.is_stale():
        raise ChromeGCNLibraryError(
            "chromegcn_amd: %s is stale (sources changed since it was built) -- rebuild: "
            "python -m chromegcn_amd._build" % path)
    _lib = open_library(path)
    return _lib


def open_library(path):
    """ctypes handle of the library at `path` with every signature of include/chromegcn.h set (not cached: tuning
    tools open a variant build beside the in-tree one)."""
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ChromeGCNLibraryError("chromegcn_amd: symbol %s missing from %s" % (name, path)) from e
        fn.restype = res
        fn.argtypes = args
    got = lib.cgcn_abi_version()
    if got != ABI_VERSION:
        raise ChromeGCNLibraryError("chromegcn_amd: ABI version %d != expected %d; rebuild" % (got, ABI_VERSION))
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().cgcn_strerror(rc).decode()
        raise RuntimeError("chromegcn_amd: %s failed: %s (code %d)" % (what, msg, rc))


class HeadGrad(ctypes.Structure):
    """mirror of cgcn_head_grad (include/chromegcn.h)"""
    _fields_ = [("dym", _c_vp), ("bnc", _c_vp), ("save_mean", _c_vp), ("save_invstd", _c_vp), ("bn_w", _c_vp),
                ("dropout_p", _c_float), ("rng_state", _c_vp), ("part", _c_vp), ("n_partials", _c_int), ("C", _c_int),
                ("dW_out", _c_vp), ("db_out", _c_vp), ("accumulate", _c_int), ("dloss", _c_vp), ("dbn_w", _c_vp),
                ("dbn_b", _c_vp), ("stat_acc", _c_vp)]


class SgdFuse(ctypes.Structure):
    """mirror of cgcn_sgd_fuse (include/chromegcn.h)"""
    _fields_ = [("param", _c_vp), ("grad", _c_vp), ("momentum_buf", _c_vp), ("count", ctypes.c_longlong),
                ("lr", _c_float), ("momentum", _c_float), ("weight_decay", _c_float), ("grad_scale", _c_float),
                ("nesterov", _c_int), ("rng_state", _c_vp)]


def ptr(t):
    """device pointer of a tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


_aux_streams = {}


def aux_stream_ptr():
    """a per-device side stream for cgcn_layer_bwd's concurrent partial reduction.  OFF by default: measured on
    MI355X (chr21-like step, HIP graph) the fork/join edges cost more than the 7.5 us reduction they hide
    (0.295 ms with, 0.277 ms without).  CHROMEGCN_AUX_STREAM=1 turns it on."""
    if not os.environ.get("CHROMEGCN_AUX_STREAM"):
        return None
    dev = torch.cuda.current_device()
    s = _aux_streams.get(dev)
    if s is None:
        s = _aux_streams[dev] = torch.cuda.Stream(device=dev)
    return s.cuda_stream
