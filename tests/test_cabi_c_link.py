"""The boundary is a C ABI: include/chromegcn.h must compile as plain C (gcc, -std=c99 -pedantic) and a C program must
link against libchromegcn_hip.so and call its host-side entry points (no GPU needed, no compute calls)."""
import os
import shutil
import subprocess

import pytest

from chromegcn_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "chromegcn.h"

int main(void) {
    if (cgcn_abi_version() != CGCN_ABI_VERSION) { printf("abi mismatch\n"); return 1; }
    if (strcmp(cgcn_strerror(CGCN_OK), "ok") != 0) return 2;
    if (cgcn_layer_bwd_workspace_bytes(5000, 2, 128) == 0) return 3;          /* supported shape */
    if (cgcn_layer_bwd_workspace_bytes(5000, 2, 100) != 0) return 4;          /* unsupported width */
    if (cgcn_head_workspace_bytes(5000, 2, 128, 103) == 0) return 5;
    if (cgcn_metrics_workspace_bytes(1000, 103) == 0) return 6;
    /* argument checking happens before any launch: NULL pointers are rejected without touching a device */
    if (cgcn_spmm((cgcn_stream_t)0, 10, 10, 1, 128, NULL, NULL, NULL, NULL, NULL, NULL, NULL) != CGCN_ERR_BAD_ARG) return 7;
    if (cgcn_spmm((cgcn_stream_t)0, 10, 10, 1, 130, NULL, NULL, NULL, NULL, NULL, NULL, NULL) != CGCN_ERR_UNSUPPORTED) return 8;
    {   /* the optional per-graph facts are a plain host struct: built in C, passed by address, argument checks unchanged */
        cgcn_graph_aux aux = {NULL, NULL, 0};
        aux.max_row_len = 5000;
        if (cgcn_spmm((cgcn_stream_t)0, 10, 10, 1, 128, NULL, NULL, NULL, NULL, NULL, NULL, &aux) != CGCN_ERR_BAD_ARG) return 9;
        if (sizeof(aux.col16) != sizeof(void *) || sizeof(aux.max_row_len) != 4) return 10;
    }
    /* struct layouts, for the ctypes mirrors (chromegcn_amd/graph.py GraphAux, _lib.py HeadGrad / SgdFuse) */
    printf("layout cgcn_graph_aux %zu", sizeof(cgcn_graph_aux));
    printf(" %zu %zu %zu %zu %zu %zu %zu %zu\n", offsetof(cgcn_graph_aux, col16), offsetof(cgcn_graph_aux, row_order),
           offsetof(cgcn_graph_aux, max_row_len), offsetof(cgcn_graph_aux, band_halfwidth), offsetof(cgcn_graph_aux, bp_rowptr),
           offsetof(cgcn_graph_aux, bp_col), offsetof(cgcn_graph_aux, bp_col16), offsetof(cgcn_graph_aux, bp_row_order));
    printf("layout cgcn_head_grad %zu %zu %zu %zu %zu %zu\n", sizeof(cgcn_head_grad), offsetof(cgcn_head_grad, dropout_p),
           offsetof(cgcn_head_grad, n_partials), offsetof(cgcn_head_grad, accumulate), offsetof(cgcn_head_grad, dbn_b),
           offsetof(cgcn_head_grad, stat_acc));
    printf("layout cgcn_sgd_fuse %zu %zu %zu %zu\n", sizeof(cgcn_sgd_fuse), offsetof(cgcn_sgd_fuse, count), offsetof(cgcn_sgd_fuse, lr),
           offsetof(cgcn_sgd_fuse, rng_state));
    printf("c-abi ok v%d\n", cgcn_abi_version());
    return 0;
}
'''


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    _build.build_library()
    src = tmp_path / "cabi.c"
    src.write_text(SRC)
    exe = tmp_path / "cabi"
    libdir = os.path.dirname(_build.LIB)
    rocm = "/opt/rocm/lib"
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L" + libdir, "-lchromegcn_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath," + rocm, "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + rocm + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "c-abi ok v" in out.stdout
    # the ctypes mirrors of the host structs have the C compiler's layout (a field added to one side only would shift every
    # pointer behind it: the library would read garbage addresses)
    import ctypes
    from chromegcn_amd import _lib, graph as G
    lay = {l.split()[1]: [int(v) for v in l.split()[2:]] for l in out.stdout.splitlines() if l.startswith("layout ")}
    A = G.GraphAux
    assert lay["cgcn_graph_aux"] == [ctypes.sizeof(A)] + [getattr(A, f).offset for f in
                                                          ("col16", "row_order", "max_row_len", "band_halfwidth", "bp_rowptr", "bp_col", "bp_col16", "bp_row_order")]
    assert [f for f, _ in A._fields_] == ["col16", "row_order", "max_row_len", "band_halfwidth", "bp_rowptr", "bp_col", "bp_col16", "bp_row_order"]
    H = _lib.HeadGrad
    assert lay["cgcn_head_grad"] == [ctypes.sizeof(H), H.dropout_p.offset, H.n_partials.offset, H.accumulate.offset, H.dbn_b.offset,
                                     H.stat_acc.offset]
    F = _lib.SgdFuse
    assert lay["cgcn_sgd_fuse"] == [ctypes.sizeof(F), F.count.offset, F.lr.offset, F.rng_state.offset]
