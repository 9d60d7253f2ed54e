mkdir -p gpurun_out/r04c
python tools/kdense.py dold=variants/libcgcn_dold.so base=chromegcn_amd/libchromegcn_hip.so dold2=variants/libcgcn_dold.so base2=chromegcn_amd/libchromegcn_hip.so --n=5776,16264,29910 > gpurun_out/r04c/kdense.txt 2>&1
cat gpurun_out/r04c/kdense.txt | cut -c1-330
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04c/pytest.txt; cat gpurun_out/r04c/pytest.txt
timeout 600 python -m pytest tests/test_gpu_fullsize_oracle.py -m gpu -q -s -k hub 2>&1 | grep -v "^$" | tail -150 > gpurun_out/r04c/hub_parity.txt; tail -5 gpurun_out/r04c/hub_parity.txt
for v in base nomfma norow; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$GRAFT_REPO_ROOT/variants/libcgcn_$v.so; fi
  python bench.py --workload chr1 --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), d['roofline']['all_kernels_us'])"
done > gpurun_out/r04c/forms.txt 2>&1; cat gpurun_out/r04c/forms.txt
unset CHROMEGCN_LIB
AB_REPS=2 AB_WL="genome" AB_VARIANTS="old dold nt7 nt2 nt5 pf2" bash tools/ab.sh > gpurun_out/r04c/ab.txt 2>&1; cat gpurun_out/r04c/ab.txt
