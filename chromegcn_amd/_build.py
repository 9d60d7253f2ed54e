"""Builds libchromegcn_hip.so (gfx950 only) in-tree with hipcc.  Called by __graft_entry__.build()
and lazily by chromegcn_amd._lib when the library is missing and hipcc is available."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = [os.path.join(PKG, "csrc", "cgcn_kernels.hip"), os.path.join(PKG, "csrc", "cgcn_head.hip"),
       os.path.join(PKG, "csrc", "cgcn_graph.hip"), os.path.join(PKG, "csrc", "cgcn_metrics.hip")]
HDR = [os.path.join(ROOT, "include", "chromegcn.h"), os.path.join(PKG, "csrc", "cgcn_common.hpp")]
LIB = os.path.join(PKG, "libchromegcn_hip.so")


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SRC + HDR if os.path.exists(f))


def build_library(force=False, verbose=False, out=None):
    if out is None and not force and not is_stale():
        return LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("chromegcn_amd: hipcc not found; cannot build libchromegcn_hip.so")
    extra = os.environ.get("CGCN_EXTRA_FLAGS", "").split()  # tuning experiments only (tools/)
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include")] + extra + SRC + ["-o", (out or LIB) + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace((out or LIB) + ".tmp", out or LIB)
    return out or LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
