#!/bin/bash
# per-kernel VGPRs / spills / scratch / LDS / occupancy of one source file (compiler remarks): bash tools/regs.sh <file.hip> [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -I"$(dirname "$0")/../include" "$@" "$f" -o /tmp/regs_$$.so -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys, subprocess
cur = None; rows = []
for ln in sys.stdin:
    if "error" in ln: print(ln.rstrip())
    m = re.search(r"remark:\s+(Function Name|VGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k.split(" [")[0]] = v
for r in rows:
    nm = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-48s vgpr %4s spill %3s scratch %4s lds %6s occ %s" % (nm[:48], r.get("VGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize"), r.get("LDS Size"), r.get("Occupancy")))
'
rm -f /tmp/regs_$$.so
