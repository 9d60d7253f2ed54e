mkdir -p gpurun_out/r04e
python tools/kring.py old=variants/libcgcn_old.so base=chromegcn_amd/libchromegcn_hip.so pf2=variants/libcgcn_pf2.so noslp=variants/libcgcn_noslp.so prio0=variants/libcgcn_prio0.so nowait=variants/libcgcn_nowait.so nomfma=variants/libcgcn_nomfma.so norow=variants/libcgcn_norow.so old2=variants/libcgcn_old.so base2=chromegcn_amd/libchromegcn_hip.so --n=5776,16264,29910 > gpurun_out/r04e/kring.txt 2>&1
cut -c1-110 gpurun_out/r04e/kring.txt
timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r04e/pytest.txt; cat gpurun_out/r04e/pytest.txt
AB_REPS=2 AB_WL="genome chr21 chr1" AB_VARIANTS="old pf2 noslp prio0" bash tools/ab.sh > gpurun_out/r04e/ab.txt 2>&1; cat gpurun_out/r04e/ab.txt
for v in base nomfma norow nowait; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$GRAFT_REPO_ROOT/variants/libcgcn_$v.so; fi
  python bench.py --workload chr1 --no-extras --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), d['roofline']['all_kernels_us'])"
done > gpurun_out/r04e/forms.txt 2>&1; cat gpurun_out/r04e/forms.txt
