"""The boundary is a C ABI: include/chromegcn.h must compile as plain C (gcc, -std=c99 -pedantic) and a C program must
link against libchromegcn_hip.so and call its host-side entry points (no GPU needed, no compute calls)."""
import os
import shutil
import subprocess

import pytest

from chromegcn_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
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
