mkdir -p gpurun_out/r04d
python tools/kdense.py dold=variants/libcgcn_dold.so base=chromegcn_amd/libchromegcn_hip.so io8pf4=variants/libcgcn_io8pf4.so io4pf3=variants/libcgcn_io4pf3.so dold2=variants/libcgcn_dold.so base2=chromegcn_amd/libchromegcn_hip.so --n=16264,29910 > gpurun_out/r04d/kdense.txt 2>&1
cut -c1-200 gpurun_out/r04d/kdense.txt
python tools/kring.py old=variants/libcgcn_old.so base=chromegcn_amd/libchromegcn_hip.so r8=variants/libcgcn_r8.so old2=variants/libcgcn_old.so base2=chromegcn_amd/libchromegcn_hip.so r8b=variants/libcgcn_r8.so --n=16264,29910 > gpurun_out/r04d/kring.txt 2>&1
cut -c1-110 gpurun_out/r04d/kring.txt
timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04d/pytest.txt; cat gpurun_out/r04d/pytest.txt
AB_REPS=2 AB_WL="genome" AB_VARIANTS="old dold r8d r8 io8pf4 io4pf3" bash tools/ab.sh > gpurun_out/r04d/ab.txt 2>&1; cat gpurun_out/r04d/ab.txt
for v in base wide; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$GRAFT_REPO_ROOT/variants/libcgcn_$v.so; fi
  for wl in "chr21" "chr21 --generator hic_like" "chr1 --generator hic_like"; do
    python bench.py --workload $wl --d 256 --layers 4 --no-extras --no-cpu-baseline --no-roofline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$wl', round(d['ms_per_step'],4))"
  done
done > gpurun_out/r04d/d256.txt 2>&1; cat gpurun_out/r04d/d256.txt
