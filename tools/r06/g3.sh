mkdir -p gpurun_out/r06
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -I include -I chromegcn_amd/csrc tools/micro/tanh_check.hip -o /tmp/tanh_check 2>/dev/null && /tmp/tanh_check > gpurun_out/r06/tanh_check.txt 2>&1; cat gpurun_out/r06/tanh_check.txt
V="base=chromegcn_amd/libchromegcn_hip.so notanh=variants/libcgcn_notanh.so nomfma=variants/libcgcn_nomfma.so norow=variants/libcgcn_norow.so allx=variants/libcgcn_allx.so"
python tools/kdense.py base=chromegcn_amd/libchromegcn_hip.so --d=256 --n=37,5781,16264 > gpurun_out/r06/kdense256_v3.txt 2>&1
python tools/kdense.py $V --d=256 --n=5776,29910 >> gpurun_out/r06/kdense256_v3.txt 2>&1
cut -c1-330 gpurun_out/r06/kdense256_v3.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_band.py tests/test_gpu_sliced_routes.py tests/test_gpu_modules.py -x -q -m gpu > gpurun_out/r06/t1.log 2>&1; tail -3 gpurun_out/r06/t1.log
