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
# what the compiler reports for every kernel of the build (registers, spills, scratch, LDS, occupancy), written next to the
# library by build_library: tests/test_kernel_resources.py fails on any spilled register or scratch byte
RESOURCES = LIB + ".resources.json"
BASE_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Rpass-analysis=kernel-resource-usage"]


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
        jobs.append((name, target, tmp, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
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


def parse_resource_remarks(text: str):
    """-Rpass-analysis=kernel-resource-usage remarks -> [{name, vgprs, agprs, spill, sgpr_spill, scratch, lds, occupancy}] (one per kernel)"""
    import re
    rows, cur = [], None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "VGPRs Spill": "spill", "SGPRs Spill": "sgpr_spill",
            "ScratchSize [bytes/lane]": "scratch", "LDS Size [bytes/block]": "lds", "Occupancy [waves/SIMD]": "occupancy"}
    for ln in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?: \[[^\]]+\])?): (\S+)", ln)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None and k in keys:
            try:
                cur[keys[k]] = int(v)
            except ValueError:
                cur[keys[k]] = v
    return rows


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
        res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        remarks = [ln for ln in res.stderr.splitlines() if "remark:" in ln]
        other = [ln for ln in res.stderr.splitlines() if "remark:" not in ln]
        if other and (verbose or res.returncode != 0):
            print("\n".join(other))
        if res.returncode != 0:
            raise subprocess.CalledProcessError(res.returncode, cmd)
        os.replace(tmp, target)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    if out is None:
        import json
        with open(RESOURCES + ".tmp.%d" % os.getpid(), "w") as f:
            json.dump({"source_hash": source_hash([]), "kernels": parse_resource_remarks("\n".join(remarks))}, f, indent=0)
        os.replace(RESOURCES + ".tmp.%d" % os.getpid(), RESOURCES)
        with open(HASH + ".tmp.%d" % os.getpid(), "w") as f:
            f.write(source_hash([]) + "\n")
        os.replace(HASH + ".tmp.%d" % os.getpid(), HASH)
    return target


def kernel_resources():
    """The compiler's resource report of the in-tree library's kernels, or None when there is none for the current sources."""
    import json
    try:
        with open(RESOURCES) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    return d["kernels"] if d.get("source_hash") == source_hash([]) else None


if __name__ == "__main__":
    import sys
    print(build_library(force=True, verbose=True))
    if "--test-variants" in sys.argv:
        print(build_test_variants(force=True, verbose=True))
