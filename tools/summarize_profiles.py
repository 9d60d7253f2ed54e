#!/usr/bin/env python3
"""Turns the raw rocprofv3 CSVs of tools/profile_round.sh into the small per-kernel summaries kept under profiles/."""
import csv, glob, json, os, sys
from collections import defaultdict


def short(name):
    name = name.replace(",", ";")
    return name.split("(")[0].strip()


def find(d, pattern):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    return hits[0] if hits else None


def counters(d):
    """{kernel: {counter: [values]}}, {kernel: [durations_us]}"""
    f = find(d, "*counter_collection.csv")
    vals, durs = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    if not f:
        return vals, durs
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return vals, durs


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out = os.path.join(root, "summary")
    os.makedirs(out, exist_ok=True)
    f = find(os.path.join(root, "stats"), "*kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(out, f"{tag}_kernel_stats_bench_chr21.csv"), "w") as o:
            o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline\n")
            o.write("# (headline train steps + engine-default steps + bench.py's isolated k_layer_fwd launches for the roofline + eval)\n")
            o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for r in rows:
                o.write("%s,%s,%s,%s,%s,%s,%s\n" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
    traffic = {}
    per = {}
    for cname, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        vals, _ = counters(os.path.join(root, sub))
        with open(os.path.join(out, f"{tag}_pmc_{cname.lower()}_bench_chr21.csv"), "w") as o:
            o.write(f"# rocprofv3 --pmc {cname} --kernel-trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline ; mean per launch, unit KB (x1024 = bytes)\n")
            o.write("kernel,counter,launches,mean_KB\n")
            for k in sorted(vals, key=lambda k: -sum(vals[k][cname])):
                v = vals[k][cname]
                o.write("%s,%s,%d,%.1f\n" % (k, cname, len(v), sum(v) / len(v)))
                per.setdefault(k, {})[cname] = sum(v) / len(v)
    key = [k for k in per if k.startswith("void k_layer_fwd<2; 128; 1; false; false")]
    if key and "FETCH_SIZE" in per[key[0]] and "WRITE_SIZE" in per[key[0]]:
        fs, ws = per[key[0]]["FETCH_SIZE"], per[key[0]]["WRITE_SIZE"]
        traffic = {"chr21_d128": (2 * fs + ws) * 1024,
                   "_note": "k_layer_fwd<2,128,1,false,false>: (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per launch; FETCH_SIZE=%.0f KB WRITE_SIZE=%.0f KB "
                            "(rocprofv3 --pmc, separate passes, %s kernels). gfx950: FETCH_SIZE reports half the bytes of wide (16 B/lane) reads, hence the "
                            "factor 2. FETCH_SIZE counts L2->fabric reads and includes Infinity-Cache hits, so this is traffic beyond L2, not HBM-only." % (fs, ws, tag)}
        json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    vals, durs = counters(os.path.join(root, "sq"))
    names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD"]
    with open(os.path.join(out, f"{tag}_pmc_sq_bench_chr21.csv"), "w") as o:
        o.write("# rocprofv3 --pmc " + " ".join(names) + " --kernel-trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline ; mean per launch; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles\n")
        o.write("kernel,us," + ",".join(names) + "\n")
        for k in sorted(vals, key=lambda k: -sum(durs[k])):
            if not k.startswith(("void k_", "k_")):
                continue
            o.write("%s,%.1f,%s\n" % (k, sum(durs[k]) / len(durs[k]), ",".join("%d" % (sum(vals[k][c]) / max(1, len(vals[k][c]))) for c in names)))
    vals, durs = counters(os.path.join(root, "mfma"))
    names = ["SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES"]
    if vals:
        with open(os.path.join(out, f"{tag}_pmc_mfma_bench_chr21.csv"), "w") as o:
            o.write("# rocprofv3 --pmc " + " ".join(names) + " --kernel-trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline ; mean per launch.\n")
            o.write("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz): fraction of the chip's MFMA issue slots in use\n")
            o.write("kernel,us," + ",".join(names) + ",mfma_util\n")
            for k in sorted(vals, key=lambda k: -sum(durs[k])):
                if not k.startswith(("void k_", "k_")):
                    continue
                us = sum(durs[k]) / len(durs[k])
                mean = {c: sum(vals[k][c]) / max(1, len(vals[k][c])) for c in names}
                o.write("%s,%.1f,%s,%.3f\n" % (k, us, ",".join("%d" % mean[c] for c in names), mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * us * 2400.0)))
    print("summaries in", out, os.listdir(out))


if __name__ == "__main__":
    main()
