#!/bin/bash
# Collects the rocprofv3 evidence for one workload on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag> <workload-name> [bench.py args ...]
#   e.g. tools/profile_round.sh r02 genome
#        tools/profile_round.sh r02 chr1_hic --workload chr1 --hic-like
# One kernel-trace/stats pass and four separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*, MFMA/LDS + L2 hit/miss), never
# combined with sys/hip/hsa traces.  The program after `--` is python3 itself (no wrapper hop); the library must
# already be built (chromegcn_amd._lib never compiles).  Raw output: gpurun_out/<tag>_<wl>/ ; summaries:
# gpurun_out/<tag>_<wl>/summary/ (copy those to profiles/).
set -u
TAG=${1:-r02}; WL=${2:-genome}; shift 2 || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}_${WL}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-extras $*"
# the PMC passes skip the isolated roofline launches too (they would be averaged into the per-kernel counters)
BP="$B --no-roofline"
echo "$B" > "$O/cmd.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 $B --no-roofline --steps 10 --warmup 3 > "$O/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/fetch" -- python3 $BP --steps 3 --warmup 1 > "$O/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/write" -- python3 $BP --steps 3 --warmup 1 > "$O/write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d "$O/sq" -- python3 $BP --steps 3 --warmup 1 > "$O/sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$O/mfma" -- python3 $BP --steps 3 --warmup 1 > "$O/mfma.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$O/l2" -- python3 $BP --steps 3 --warmup 1 > "$O/l2.log" 2>&1
python3 "$R/tools/summarize_profiles.py" "$O" "$TAG" "$WL" "$*"
# the raw per-dispatch CSVs are tens of MB per pass; gpurun merges at most 64 MiB back: keep the summaries and logs only
rm -rf "$O/stats" "$O/fetch" "$O/write" "$O/sq" "$O/mfma" "$O/l2"
