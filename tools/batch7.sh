mkdir -p gpurun_out/r04g
python tools/kring.py noearly=variants/libcgcn_noearly.so base=chromegcn_amd/libchromegcn_hip.so ob3=variants/libcgcn_ob3.so norow=variants/libcgcn_norow.so ob3norow=variants/libcgcn_ob3norow.so noearlynorow=variants/libcgcn_noearlynorow.so base2=chromegcn_amd/libchromegcn_hip.so noearly2=variants/libcgcn_noearly.so ob3b=variants/libcgcn_ob3.so --n=5776,16264,29910 > gpurun_out/r04g/kring.txt 2>&1
cut -c1-110 gpurun_out/r04g/kring.txt
timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r04g/pytest.txt; cat gpurun_out/r04g/pytest.txt
AB_REPS=3 AB_WL="genome chr1" AB_VARIANTS="noearly ob3 nt7" bash tools/ab.sh > gpurun_out/r04g/ab.txt 2>&1; cat gpurun_out/r04g/ab.txt
