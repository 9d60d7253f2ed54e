mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_stat_acc.py tests/test_gpu_cabi_errors.py -x -q -m gpu > gpurun_out/r06/t_acc.log 2>&1; tail -4 gpurun_out/r06/t_acc.log
for rep in 1 2; do
for acc in 1 0; do
  for wl in "chr21 --d 256 --layers 4 --steps 50" "chr21 --steps 100" "config1 --steps 100" "genome --steps 20"; do
    echo -n "acc=$acc $wl: "
    CGCN_STAT_ACC=$acc python bench.py --no-cpu-baseline --no-extras --no-roofline --warmup 5 --workload $wl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"
  done
done; done 2>&1 | tee gpurun_out/r06/acc_ab.txt
bash tools/kstats.sh d256 --workload chr21 --d 256 --layers 4 --no-roofline --steps 10 --warmup 3 | tee gpurun_out/r06/kstats_d256_acc.txt
bash tools/kstats.sh chr21 --workload chr21 --no-roofline --steps 10 --warmup 3 | tee gpurun_out/r06/kstats_chr21_acc.txt
