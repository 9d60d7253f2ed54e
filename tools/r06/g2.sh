mkdir -p gpurun_out/r06
V="base=chromegcn_amd/libchromegcn_hip.so nostag=variants/libcgcn_nostag.so notanh=variants/libcgcn_notanh.so nomfma=variants/libcgcn_nomfma.so norow=variants/libcgcn_norow.so allx=variants/libcgcn_allx.so"
python tools/kdense.py $V --d=256 --n=5776,29910 > gpurun_out/r06/kdense256_decomp.txt 2>&1
KT=1 CHROMEGCN_LIB=$GRAFT_REPO_ROOT/variants/libcgcn_kt.so python tools/kdense_stamps.py 256 5776 29910 >> gpurun_out/r06/kdense256_decomp.txt 2>&1
cat gpurun_out/r06/kdense256_decomp.txt | cut -c1-200
