mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_cabi_errors.py tests/test_gpu_stat_acc.py tests/test_gpu_head.py tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_epoch_oracle.py tests/test_gpu_loop.py tests/test_gpu_modules.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2; do
  for wl in "chr21 --steps 200" "config1 --steps 200" "genome --steps 20" "chr21 --hic-like --steps 200"; do
    echo -n "$wl: "
    python bench.py --no-cpu-baseline --no-extras --no-roofline --warmup 5 --workload $wl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r06/prezero_ab.txt
bash tools/kstats.sh pz --workload chr21 --no-roofline --steps 10 --warmup 3 | head -9
