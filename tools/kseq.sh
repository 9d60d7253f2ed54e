#!/bin/bash
# kernel sequence of the last replayed step (tuning tool): bash tools/kseq.sh [bench args]  -> names + durations in launch order
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kseq
rocprofv3 --kernel-trace --output-format csv -d /tmp/kseq -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 3 --warmup 1 "$@" > /tmp/kseq.log 2>&1
f=$(find /tmp/kseq -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# last step: walk back from the end to the previous k_layer_fwd/k_aggregate... simply print the last 40 launches
for r in rows[-int(__import__("os").environ.get("KSEQ_N","40")):]:
    print("%-70s %7.1f us  grid %s wg %s" % (r["Kernel_Name"][:70], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
PY
