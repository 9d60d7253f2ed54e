mkdir -p gpurun_out/r06
KS_D=256 CHROMEGCN_LIB=$GRAFT_REPO_ROOT/variants/libcgcn_hft.so python tools/khead.py --stamps 2>&1 | tee gpurun_out/r06/head256_stamps.txt
