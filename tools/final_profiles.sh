set -u
T=$1
mkdir -p gpurun_out/$T
python -c "from chromegcn_amd import _build; print(_build.source_hash([]))" > gpurun_out/$T/src_hash.txt
if [ -z "${SKIP_TESTS:-}" ]; then python -m pytest tests -m gpu -q > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest.log; tail -3 gpurun_out/$T/pytest.log; fi
( time python bench.py > gpurun_out/$T/bench_genome.json 2> gpurun_out/$T/bench_genome.err ) 2> gpurun_out/$T/bench_genome.time
python bench.py --hic-like --no-cpu-baseline > gpurun_out/$T/bench_genome_hic.json 2>/dev/null
for w in chr21 chr1 config1; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/$T/bench_$w.json 2>/dev/null; python bench.py --workload $w --hic-like --no-cpu-baseline > gpurun_out/$T/bench_${w}_hic.json 2>/dev/null; done
python bench.py --generator hub --no-cpu-baseline > gpurun_out/$T/bench_genome_hub.json 2>/dev/null
for a in both constant none; do python bench.py --adj-type $a --no-cpu-baseline --no-extras > gpurun_out/$T/bench_genome_adj_$a.json 2>/dev/null; done
for w in chr21 chr1; do python bench.py --workload $w --generator hub --no-cpu-baseline > gpurun_out/$T/bench_${w}_hub.json 2>/dev/null; done
python bench.py --workload chr21 --d 256 --layers 4 --no-cpu-baseline > gpurun_out/$T/bench_chr21_d256L4.json 2>/dev/null
python bench.py --d 256 --layers 4 --no-cpu-baseline > gpurun_out/$T/bench_genome_d256L4.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --share-gpu --no-cpu-baseline --no-extras > gpurun_out/$T/bench_genome_2ranks_one_gpu_gloo.json 2>/dev/null
python bench.py --workload e2e > gpurun_out/$T/bench_e2e.json 2>/dev/null
python bench.py --workload e2e --e2e-windows 0 > gpurun_out/$T/bench_e2e_chr21_full.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sal_prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sal_prof -- python3 $GRAFT_REPO_ROOT/tools/saliency_bench.py > $GRAFT_REPO_ROOT/gpurun_out/$T/saliency.log 2>&1; f=$(find /tmp/sal_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- python3 tools/saliency_bench.py (adjacency saliency of a chr21-size and a chr1-size chromosome, 6 calls each)"; cat "$f"; } > $GRAFT_REPO_ROOT/gpurun_out/$T/saliency_kernel_stats.csv )
bash tools/profile_round.sh $T genome
bash tools/profile_round.sh $T genome_hic --hic-like
bash tools/profile_round.sh $T chr21 --workload chr21
bash tools/profile_round.sh $T chr21_hic --workload chr21 --hic-like
bash tools/profile_round.sh $T chr1 --workload chr1
bash tools/profile_round.sh $T chr1_hic --workload chr1 --hic-like
bash tools/profile_round.sh $T chr21_d256L4 --workload chr21 --d 256 --layers 4
bash tools/profile_round.sh $T genome_hub --generator hub
bash tools/profile_round.sh $T genome_constant --adj-type constant
bash tools/profile_round.sh $T genome_both --adj-type both
for f in gpurun_out/$T/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,2), 'M win/s')
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
