#!/bin/bash
# GPU busy time vs wall time of the timed epochs (tuning tool): sum of kernel durations per step from a kernel trace
# against bench.py's ms_per_step.  bash tools/busy.sh [bench args]
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/busy
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/busy -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 20 --warmup 5 "$@" > /tmp/busy.log 2>&1
f=$(find /tmp/busy -name "*kernel_stats.csv" | head -1)
python3 - "$f" /tmp/busy.log <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
line = [l for l in open(sys.argv[2]) if l.startswith("{")][-1]
d = json.loads(line)
steps = d["steps"] + d["warmup"] + 1          # + the capture pass
tot = sum(float(r["TotalDurationNs"]) for r in rows if not r["Name"].startswith("void at::") and "copyBuffer" not in r["Name"]) / 1e6
print("ms_per_step %.3f   kernel time per step (all epochs incl. warm-up and capture) %.3f ms" % (d["ms_per_step"], tot / steps))
for r in rows[:10]:
    print("  %-56s per step %7.3f ms  avg %7.1f us" % (r["Name"][:56], float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
PY
