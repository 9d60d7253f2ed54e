"""Copy the summaries of a tools/final_profiles.sh run from gpurun_out/ into profiles/ (tracked) and rebuild
profiles/traffic.json:  python tools/collect_profiles.py <tag> [<old tag to remove>]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    old = sys.argv[2] if len(sys.argv) > 2 else None
    prof = os.path.join(ROOT, "profiles")
    if old:
        for f in glob.glob(os.path.join(prof, old + "_*")):
            # only what the new round's run regenerates goes: per-round kernel-stat / counter summaries and bench lines; every
            # hand-written record (*.txt: experiments, probes, launch-floor notes ...) stays -- DESIGN.md cites them
            if f.endswith(".csv") or (f.endswith(".json") and "_bench_line_" in f):
                os.remove(f)
    traffic = {}
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_*", "summary"))):
        for f in glob.glob(os.path.join(d, "*")):
            name = os.path.basename(f)
            if name.startswith("traffic_"):
                traffic.update(json.load(open(f)))
            else:
                shutil.copy(f, os.path.join(prof, name))
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", tag, "bench_*.json")):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            wl = os.path.basename(f)[len("bench_"):-len(".json")]
            open(os.path.join(prof, "%s_bench_line_%s.json" % (tag, wl)), "w").write(lines[-1] + "\n")
    for extra in ("saliency_kernel_stats.csv", "pytest.log"):
        src = os.path.join(ROOT, "gpurun_out", tag, extra)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(prof, "%s_%s" % (tag, "pytest_gpu.log" if extra == "pytest.log" else extra)))
    if "chr21_d256L4_d256" in traffic:   # bench.py keys its lookup by workload (+ generator) + width
        traffic["chr21_d256"] = traffic.pop("chr21_d256L4_d256")
    note = traffic.pop("_note", None)
    out = dict(sorted(traffic.items()))
    if note:
        out["_note"] = note
    # the library sources these counters were measured on: bench.py reports a stored value only for the same sources
    # (recorded on the GPU box from the very snapshot that was profiled: tools/final_profiles.sh; an argument overrides it)
    hfile = os.path.join(ROOT, "gpurun_out", tag, "src_hash.txt")
    if len(sys.argv) > 3:
        out["_src_hash"] = sys.argv[3]
    elif os.path.exists(hfile):
        out["_src_hash"] = open(hfile).read().strip()
    else:
        sys.path.insert(0, ROOT)
        from chromegcn_amd import _build
        out["_src_hash"] = _build.source_hash([])
    json.dump(out, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
    # the bench lines were printed before this traffic.json existed: their `roofline.traffic` is the stored value of the
    # previous profile run; point the copies at the values collected from THIS run's PMC passes
    def lookup(key, kernel):
        ent = out.get(key)
        if not isinstance(ent, dict) or not kernel:
            return None
        base = kernel.split(":")[0].split("(")[0].strip()
        for k, v in (ent.get("per_kernel") or {}).items():
            if k.startswith("void " + base + "<") or k.startswith(base):
                return v.get("bytes_per_launch")
        return None

    for f in glob.glob(os.path.join(prof, "%s_bench_line_*.json" % tag)):
        d = json.loads(open(f).read())
        wl = os.path.basename(f)[len(tag + "_bench_line_"):-len(".json")]
        key = "chr21_d256" if wl == "chr21_d256L4" else wl + "_d128"
        src = "stored profile value (profiles/traffic.json, tag %s: the PMC passes of the same run; 2*FETCH_SIZE + WRITE_SIZE per launch, beyond-L2 bytes incl. Infinity-Cache hits), not measured inside bench.py" % tag
        for r in [d.get("roofline")] + list(d.get("roofline_top3") or []):
            if not r:
                continue
            t = lookup(key, r.get("kernel"))
            r["traffic"], r["traffic_source"] = t, (src if t is not None else None)
        open(f, "w").write(json.dumps(d) + "\n")
    print("traffic keys:", [k for k in out if not k.startswith("_")], "src_hash", out["_src_hash"][:16])


if __name__ == "__main__":
    main()
