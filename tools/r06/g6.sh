mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_epoch_oracle.py -x -q -m gpu -s > gpurun_out/r06/t_epoch.log 2>&1; tail -8 gpurun_out/r06/t_epoch.log
python -m pytest tests/test_gpu_two_rank.py -x -q -m gpu > gpurun_out/r06/t_ranks.log 2>&1; tail -8 gpurun_out/r06/t_ranks.log
python -m pytest tests/test_bench_launcher.py -x -q -m gpu -k "eight" > gpurun_out/r06/t_launch8.log 2>&1; tail -8 gpurun_out/r06/t_launch8.log
