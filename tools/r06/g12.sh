mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_sgd_fuse.py tests/test_gpu_ring_stress.py tests/test_gpu_band.py tests/test_gpu_loop.py -x -q -m gpu 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --no-extras --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"; }
for rep in 1 2 3; do for v in base wide0; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$GRAFT_REPO_ROOT/variants/libcgcn_$v.so; fi
  echo -n "$v: "; run --workload chr21 --d 256 --layers 4 --steps 100; run --d 256 --layers 4 --steps 10; run --workload config1 --steps 200; echo
done; done 2>&1 | tee gpurun_out/r06/ab_wide_riders.txt
unset CHROMEGCN_LIB
bash tools/kstats.sh w1 --workload chr21 --d 256 --layers 4 --no-roofline --steps 10 --warmup 3 | head -6
