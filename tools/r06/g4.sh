mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_stat_acc.py tests/test_gpu_cabi_errors.py tests/test_gpu_head.py -x -q -m gpu > gpurun_out/r06/t_acc.log 2>&1; tail -15 gpurun_out/r06/t_acc.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_band.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_modules.py tests/test_gpu_loop.py -x -q -m gpu > gpurun_out/r06/t_par.log 2>&1; tail -5 gpurun_out/r06/t_par.log
for acc in 1 0; do
  CGCN_STAT_ACC=$acc python bench.py --no-cpu-baseline --no-extras --workload chr21 --d 256 --layers 4 --no-roofline --steps 50 --warmup 5 > gpurun_out/r06/b_d256_acc$acc.json 2> gpurun_out/r06/b_d256_acc$acc.err
  CGCN_STAT_ACC=$acc python bench.py --no-cpu-baseline --no-extras --workload chr21 --no-roofline --steps 100 --warmup 5 > gpurun_out/r06/b_chr21_acc$acc.json 2>> gpurun_out/r06/b_d256_acc$acc.err
  CGCN_STAT_ACC=$acc python bench.py --no-cpu-baseline --no-extras --workload config1 --no-roofline --steps 100 --warmup 5 > gpurun_out/r06/b_config1_acc$acc.json 2>> gpurun_out/r06/b_d256_acc$acc.err
  CGCN_STAT_ACC=$acc python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 20 --warmup 5 > gpurun_out/r06/b_genome_acc$acc.json 2>> gpurun_out/r06/b_d256_acc$acc.err
done
grep -H -o '"ms_per_step": [0-9.]*' gpurun_out/r06/b_*acc*.json
tail -3 gpurun_out/r06/b_d256_acc1.err
