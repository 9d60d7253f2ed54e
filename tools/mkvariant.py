#!/usr/bin/env python3
"""Build tuning variants of the library: python tools/mkvariant.py name='-DFLAG=1 -DOTHER=2' ...  -> variants/libcgcn_<name>.so
(up to 4 compiles side by side; the in-tree library is never touched)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
jobs = []
for a in sys.argv[1:]:
    name, flags = a.split("=", 1)
    env = dict(os.environ, CGCN_EXTRA_FLAGS=flags)
    code = "from chromegcn_amd import _build; print(_build.build_library(out='variants/libcgcn_%s.so'))" % name
    jobs.append((name, subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=env)))
    if len(jobs) >= 4:
        n, p = jobs.pop(0)
        if p.wait() != 0:
            sys.exit("variant %s failed" % n)
for n, p in jobs:
    if p.wait() != 0:
        sys.exit("variant %s failed" % n)
