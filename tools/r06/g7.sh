mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_saliency.py -x -q -m gpu > gpurun_out/r06/t_sal.log 2>&1; tail -8 gpurun_out/r06/t_sal.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r06/bench_full.json 2> gpurun_out/r06/bench_full.err; tail -3 gpurun_out/r06/bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_full.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'])
print(json.dumps(d['cpu_baseline'])[:700])
r=d['roofline']; print({k:r[k] for k in ('kernel','frac','achieved','gather_GBps','gather_reference_GBps','frac_of_gather_reference','avg_kernel_us')})
print(r['all_kernels_us'])
print(d.get('saliency_ms'), json.dumps(d.get('sddmm_roofline'))[:600])
PY
