mkdir -p gpurun_out/r04f
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fullsize_oracle.py -m gpu -x -q -k "chr21_hub" 2>&1 | tail -80 > gpurun_out/r04f/hub_fail.txt; tail -45 gpurun_out/r04f/hub_fail.txt | cut -c1-220
cd /tmp && export TMPDIR=/tmp
for v in base norow; do
  if [ $v = base ]; then L=chromegcn_amd/libchromegcn_hip.so; else L=variants/libcgcn_$v.so; fi
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r04f/pmc_$v -- python3 $R/tools/kring.py $v=$L --n=29910 > $R/gpurun_out/r04f/pmc_$v.log 2>&1
  python3 $R/tools/pmc_quick.py $R/gpurun_out/r04f/pmc_$v > $R/gpurun_out/r04f/pmc_$v.csv 2>&1; cat $R/gpurun_out/r04f/pmc_$v.csv | head -8
  rm -rf $R/gpurun_out/r04f/pmc_$v
done
for v in base nt7; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$R/variants/libcgcn_$v.so; fi
  for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r04f/nt_${v}_$n -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 3 --warmup 1 > $R/gpurun_out/r04f/nt_${v}_$n.log 2>&1
    python3 $R/tools/pmc_quick.py $R/gpurun_out/r04f/nt_${v}_$n > $R/gpurun_out/r04f/nt_${v}_$n.csv 2>&1; head -8 $R/gpurun_out/r04f/nt_${v}_$n.csv
    rm -rf $R/gpurun_out/r04f/nt_${v}_$n
  done
done
unset CHROMEGCN_LIB
cd $R
timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r04f/pytest.txt; cat gpurun_out/r04f/pytest.txt
