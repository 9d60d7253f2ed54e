mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_stat_acc.py tests/test_gpu_head.py tests/test_gpu_epoch_oracle.py -x -q -m gpu > gpurun_out/r06/t_acc.log 2>&1; tail -4 gpurun_out/r06/t_acc.log
for rep in 1 2 3; do
for acc in 1 0; do
  echo -n "acc=$acc genome: "
  CGCN_STAT_ACC=$acc python bench.py --no-cpu-baseline --no-extras --no-roofline --warmup 5 --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r06/acc_ab2.txt
CGCN_STAT_ACC=1 bash tools/kstats.sh gen1 --no-roofline --steps 10 --warmup 3 | tee gpurun_out/r06/kstats_genome_acc1.txt
CGCN_STAT_ACC=0 bash tools/kstats.sh gen0 --no-roofline --steps 10 --warmup 3 | tee gpurun_out/r06/kstats_genome_acc0.txt
