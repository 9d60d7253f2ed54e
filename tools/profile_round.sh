#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag>      e.g. r01f
# kernel-trace/stats pass and four separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*, MFMA/LDS), never combined with
# sys/hip/hsa traces.  Raw output: gpurun_out/<tag>/ ; summaries: gpurun_out/<tag>/summary/ (copy those to profiles/).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$O/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/fetch" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$O/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/write" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$O/write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d "$O/sq" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$O/sq.log" 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$O/mfma" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$O/mfma.log" 2>&1
python3 "$R/tools/summarize_profiles.py" "$O" "$TAG"
