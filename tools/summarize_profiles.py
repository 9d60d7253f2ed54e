#!/usr/bin/env python3
"""Turns the raw rocprofv3 CSVs of tools/profile_round.sh into the small per-kernel summaries kept under profiles/:
    <tag>_<wl>_kernel_stats.csv, <tag>_<wl>_pmc_{fetch_size,write_size,sq,mfma,l2}.csv, and the workload's entry of
    traffic.json ({"<wl>_d<d>": {"bytes_per_launch", "tag", "fetch_KB", "write_KB", "launches"}})."""
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


FWD_KERNELS = ("void k_layer_fwd<", "void k_aggregate_sliced<", "void k_layer_dense<")


def is_fwd_kernel(k):
    """kernels of one cgcn_layer_fwd call: the fused k_layer_fwd, or k_aggregate_sliced + k_layer_dense (split route)"""
    return k.startswith(FWD_KERNELS)


def main():
    root, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
    bargs = sys.argv[4] if len(sys.argv) > 4 else ""
    d = 256 if "--d 256" in bargs else 128
    out = os.path.join(root, "summary")
    os.makedirs(out, exist_ok=True)
    cmd = "python3 bench.py --no-cpu-baseline --no-extras %s" % bargs
    f = find(os.path.join(root, "stats"), "*kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(out, f"{tag}_{wl}_kernel_stats.csv"), "w") as o:
            o.write("# rocprofv3 --kernel-trace --stats -- %s --no-roofline --steps 10 --warmup 3\n" % cmd)
            o.write("# (epoch / step replays only: --no-roofline, no isolated launches)\n")
            o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for r in rows:
                o.write("%s,%s,%s,%s,%s,%s,%s\n" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
    per = {}
    for cname, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        vals, _ = counters(os.path.join(root, sub))
        with open(os.path.join(out, f"{tag}_{wl}_pmc_{cname.lower()}.csv"), "w") as o:
            o.write(f"# rocprofv3 --pmc {cname} --kernel-trace -- {cmd} --steps 3 --warmup 1 ; mean per launch, unit KB (x1024 = bytes)\n")
            o.write("kernel,counter,launches,mean_KB\n")
            for k in sorted(vals, key=lambda k: -sum(vals[k][cname])):
                v = vals[k][cname]
                o.write("%s,%s,%d,%.1f\n" % (k, cname, len(v), sum(v) / len(v)))
                per.setdefault(k, {})[cname] = (sum(v), len(v))
    # traffic beyond L2 per launch of every hand-written kernel of the step (bench.py looks its dominant kernel up here)
    per_kernel = {}
    for k, v in per.items():
        if not k.startswith(("void k_", "k_")) or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        fs, ws = v["FETCH_SIZE"][0] / v["FETCH_SIZE"][1], v["WRITE_SIZE"][0] / v["WRITE_SIZE"][1]
        per_kernel[k] = {"bytes_per_launch": (2 * fs + ws) * 1024, "fetch_KB": fs, "write_KB": ws, "launches": v["FETCH_SIZE"][1]}
    if per_kernel:
        json.dump({"%s_d%d" % (wl, d): {"tag": tag, "per_kernel": per_kernel},
                   "_note": "per launch, mean over the launches of one bench run: (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes "
                            "(rocprofv3 --pmc, separate passes). gfx950: FETCH_SIZE reports half the bytes of wide (16 B/lane) "
                            "reads, hence the factor 2 (MI355X_MICROARCH.md, HBM). FETCH_SIZE counts L2->fabric reads and includes "
                            "Infinity-Cache hits: traffic beyond L2, not HBM-only."},
                  open(os.path.join(out, "traffic_%s.json" % wl), "w"), indent=1)
    vals, durs = counters(os.path.join(root, "sq"))
    names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD"]
    if vals:
        with open(os.path.join(out, f"{tag}_{wl}_pmc_sq.csv"), "w") as o:
            o.write("# rocprofv3 --pmc " + " ".join(names) + " --kernel-trace -- %s --steps 3 --warmup 1 ; mean per launch; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles\n" % cmd)
            o.write("kernel,launches,us," + ",".join(names) + "\n")
            for k in sorted(vals, key=lambda k: -sum(durs[k])):
                if not k.startswith(("void k_", "k_")):
                    continue
                o.write("%s,%d,%.1f,%s\n" % (k, len(durs[k]), sum(durs[k]) / len(durs[k]), ",".join("%d" % (sum(vals[k][c]) / max(1, len(vals[k][c]))) for c in names)))
    vals, durs = counters(os.path.join(root, "mfma"))
    names = ["SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES"]
    if vals:
        with open(os.path.join(out, f"{tag}_{wl}_pmc_mfma.csv"), "w") as o:
            o.write("# rocprofv3 --pmc " + " ".join(names) + " --kernel-trace -- %s --steps 3 --warmup 1 ; mean per launch.\n" % cmd)
            o.write("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz): fraction of the chip's MFMA issue slots in use\n")
            o.write("kernel,launches,us," + ",".join(names) + ",mfma_util\n")
            for k in sorted(vals, key=lambda k: -sum(durs[k])):
                if not k.startswith(("void k_", "k_")):
                    continue
                us = sum(durs[k]) / len(durs[k])
                mean = {c: sum(vals[k][c]) / max(1, len(vals[k][c])) for c in names}
                o.write("%s,%d,%.1f,%s,%.3f\n" % (k, len(durs[k]), us, ",".join("%d" % mean[c] for c in names), mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * us * 2400.0)))
    vals, durs = counters(os.path.join(root, "l2"))
    if vals:
        with open(os.path.join(out, f"{tag}_{wl}_pmc_l2.csv"), "w") as o:
            o.write("# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -- %s --steps 3 --warmup 1 ; mean per launch; hit rate = HIT / (HIT + MISS)\n" % cmd)
            o.write("kernel,launches,us,TCC_HIT_sum,TCC_MISS_sum,l2_hit_rate\n")
            for k in sorted(vals, key=lambda k: -sum(durs[k])):
                if not k.startswith(("void k_", "k_")):
                    continue
                h = sum(vals[k]["TCC_HIT_sum"]) / max(1, len(vals[k]["TCC_HIT_sum"]))
                m = sum(vals[k]["TCC_MISS_sum"]) / max(1, len(vals[k]["TCC_MISS_sum"]))
                o.write("%s,%d,%.1f,%d,%d,%.3f\n" % (k, len(durs[k]), sum(durs[k]) / len(durs[k]), h, m, h / max(1.0, h + m)))
    print("summaries in", out, os.listdir(out))


if __name__ == "__main__":
    main()
