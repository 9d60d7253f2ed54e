"""Builds libchromegcn_hip.so (gfx950 only) in-tree with hipcc.

Building is always an explicit act: `python -m chromegcn_amd._build`, `__graft_entry__.build()`, or the test
session's start-up hook (tests/conftest.py, before anything touches the GPU).  `chromegcn_amd._lib.load()` never
shells out to the compiler: a process that has initialised the GPU (or runs under a profiler's preload) must not
spawn hipcc, and N ranks of one job must not race to write the same file.

Staleness is decided by CONTENT, not by mtime: the build writes the hash of every source, header and flag next to
the library (`libchromegcn_hip.so.srchash`); a snapshot copied to another box keeps matching whatever the copy did
to the timestamps."""
import hashlib
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = [os.path.join(PKG, "csrc", "cgcn_kernels.hip"), os.path.join(PKG, "csrc", "cgcn_head.hip"),
       os.path.join(PKG, "csrc", "cgcn_graph.hip"), os.path.join(PKG, "csrc", "cgcn_metrics.hip")]
HDR = [os.path.join(ROOT, "include", "chromegcn.h"), os.path.join(PKG, "csrc", "cgcn_common.hpp")]
LIB = os.path.join(PKG, "libchromegcn_hip.so")
HASH = LIB + ".srchash"
BASE_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _extra_flags():
    return os.environ.get("CGCN_EXTRA_FLAGS", "").split()  # tuning experiments only (tools/)


def source_hash(extra=None) -> str:
    h = hashlib.sha256()
    h.update(" ".join(BASE_FLAGS + (list(extra) if extra is not None else [])).encode())
    for f in SRC + HDR:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


# Test-only builds of the SAME sources (tests/test_gpu_ring_stress.py): the ring kernel's row team or matrix team held
# back by wave- and slot-dependent delays.  They live beside the product library under variants/ (git-ignored like it,
# shipped to the GPU box like it), are opened explicitly by path (`_lib.open_library`) and never replace it.
TEST_VARIANTS = {
    "ring_slow_row": ["-DCGCN_EXPERIMENT_BUILD", "-DRING_TEST_SLOW_ROW=1"],
    "ring_slow_matrix": ["-DCGCN_EXPERIMENT_BUILD", "-DRING_TEST_SLOW_MATRIX=1"],
}


def variant_path(name: str) -> str:
    return os.path.join(ROOT, "variants", "libcgcn_test_%s.so" % name)


def variant_is_stale(name: str) -> bool:
    p = variant_path(name)
    try:
        with open(p + ".srchash") as f:
            return not os.path.exists(p) or f.read().strip() != source_hash(TEST_VARIANTS[name])
    except OSError:
        return True


def build_test_variants(force=False, verbose=False):
    """Build (side by side) every stale test variant; returns their paths.  Only kernels.hip differs, but a variant is a
    whole library so that it can be opened beside the product one."""
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("chromegcn_amd: hipcc not found; cannot build the test variants")
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    jobs = []
    for name, flags in TEST_VARIANTS.items():
        if not force and not variant_is_stale(name):
            continue
        target = variant_path(name)
        tmp = "%s.tmp.%d" % (target, os.getpid())
        cmd = [hipcc] + BASE_FLAGS + ["-I" + os.path.join(ROOT, "include")] + flags + SRC + ["-o", tmp]
        if verbose:
            print(" ".join(cmd))
        jobs.append((name, target, tmp, subprocess.Popen(cmd)))
    for name, target, tmp, p in jobs:
        try:
            if p.wait() != 0:
                raise RuntimeError("chromegcn_amd: building test variant %s failed" % name)
            os.replace(tmp, target)
            with open(target + ".srchash", "w") as f:
                f.write(source_hash(TEST_VARIANTS[name]) + "\n")
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return [variant_path(n) for n in TEST_VARIANTS]


def is_stale() -> bool:
    """True when the in-tree library is missing or was built from other sources than the ones in the tree."""
    if not os.path.exists(LIB) or not os.path.exists(HASH):
        return True
    try:
        with open(HASH) as f:
            return f.read().strip() != source_hash([])
    except OSError:
        return True


def build_library(force=False, verbose=False, out=None):
    extra = _extra_flags()
    if extra and out is None:
        # a tuning build must never replace the product library (and be recorded as the canonical fresh build): a
        # CGCN_EXTRA_FLAGS left over in the environment would otherwise silently swap kernels under every test
        raise RuntimeError("chromegcn_amd: CGCN_EXTRA_FLAGS=%r is set; tuning builds need an explicit variant path "
                           "(build_library(out='variants/libcgcn_<name>.so')), unset it to build the in-tree library"
                           % " ".join(extra))
    if out is None and not force and not is_stale():
        return LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("chromegcn_amd: hipcc not found; cannot build libchromegcn_hip.so")
    target = out or LIB
    tmp = "%s.tmp.%d" % (target, os.getpid())  # per-process temporary: concurrent builders never share a file
    cmd = [hipcc] + BASE_FLAGS + ["-I" + os.path.join(ROOT, "include")] + extra + SRC + ["-o", tmp]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, target)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    if out is None:
        with open(HASH + ".tmp.%d" % os.getpid(), "w") as f:
            f.write(source_hash([]) + "\n")
        os.replace(HASH + ".tmp.%d" % os.getpid(), HASH)
    return target


if __name__ == "__main__":
    import sys
    print(build_library(force=True, verbose=True))
    if "--test-variants" in sys.argv:
        print(build_test_variants(force=True, verbose=True))
