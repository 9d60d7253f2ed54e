#!/usr/bin/env python3
"""Mean per-launch counter values per kernel from a rocprofv3 --pmc csv directory (tuning tool)."""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
vals, dur, seen = defaultdict(lambda: defaultdict(list)), defaultdict(list), set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:48]
    vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
names = sorted({c for k in vals for c in vals[k]})
print("kernel,us," + ",".join(names))
for k in sorted(vals, key=lambda k: -sum(dur[k])):
    if "k_" not in k: continue
    print(k + ",%.1f," % (sum(dur[k]) / len(dur[k])) + ",".join("%d" % (sum(vals[k][c]) / max(1, len(vals[k][c]))) for c in names))
