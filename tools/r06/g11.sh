mkdir -p gpurun_out/r06
CHROMEGCN_LIB=$GRAFT_REPO_ROOT/variants/libcgcn_hash2.so python -m pytest tests/test_gpu_dropout_sgd.py tests/test_gpu_head.py -x -q -m gpu 2>&1 | tail -3
AB_VARIANTS="hash2" AB_REPS=3 AB_WL="genome chr21" bash tools/ab.sh 2>&1 | tee gpurun_out/r06/ab_hash2.txt
