mkdir -p gpurun_out/r06
python tools/kdense.py base=chromegcn_amd/libchromegcn_hip.so --d=256 --n=5776,5781,16264,29910,37 > gpurun_out/r06/kdense256.txt 2>&1
python tools/kdense.py base=chromegcn_amd/libchromegcn_hip.so --d=128 --n=16264 >> gpurun_out/r06/kdense256.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_band.py tests/test_gpu_sliced_routes.py -x -q -m gpu -k "256 or d256" > gpurun_out/r06/t1.log 2>&1; tail -3 gpurun_out/r06/t1.log
python -m pytest tests/test_gpu_fullsize_oracle.py -x -q -m gpu -k "256" > gpurun_out/r06/t2.log 2>&1; tail -3 gpurun_out/r06/t2.log
for v in default 0; do
  if [ $v = default ]; then unset CGCN_FWD_SPLIT_BYTES; else export CGCN_FWD_SPLIT_BYTES=0; fi
  python bench.py --no-cpu-baseline --no-extras --workload chr21 --d 256 --layers 4 --no-roofline --steps 50 --warmup 5 > gpurun_out/r06/b_d256_split_$v.json 2> gpurun_out/r06/b_d256_split_$v.err
  python bench.py --no-cpu-baseline --no-extras --workload chr21 --d 256 --layers 4 --hic-like --no-roofline --steps 50 --warmup 5 > gpurun_out/r06/b_d256_hic_split_$v.json 2>> gpurun_out/r06/b_d256_split_$v.err
done
unset CGCN_FWD_SPLIT_BYTES
grep -h -o '"ms_per_step": [0-9.]*' gpurun_out/r06/b_d256*.json
cat gpurun_out/r06/kdense256.txt
