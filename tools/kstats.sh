#!/bin/bash
# per-kernel time of one bench workload (tuning tool): bash tools/kstats.sh <outtag> [bench args]; CHROMEGCN_LIB selects a variant
tag=$1; shift
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > /tmp/ks_$tag.log 2>&1
f=$(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { tail -5 /tmp/ks_$tag.log; exit 1; }
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-60s calls %5s avg %8.1f us  total %8.2f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
