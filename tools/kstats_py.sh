#!/bin/bash
# per-kernel time of any python tool under rocprofv3 (tuning tool): bash tools/kstats_py.sh <outtag> <script.py> [args]
tag=$1; shift
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python3 $R/"$@" > /tmp/ks_$tag.log 2>&1
f=$(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { tail -5 /tmp/ks_$tag.log; exit 1; }
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    nm = r["Name"]
    if "rocprim" in nm:
        m = re.search(r"(onesweep\w*|radix_sort\w*|histogram\w*|scan\w*|merge\w*)", nm)
        nm = "rocprim:" + (m.group(1) if m else "?") + " " + nm[-40:]
    print("%-64s calls %5s avg %8.1f us  total %8.2f ms  %5s%%" % (nm[:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
