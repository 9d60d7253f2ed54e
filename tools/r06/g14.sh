mkdir -p gpurun_out/r06
for v in base nobacc nostage noticket hrsx_all; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$GRAFT_REPO_ROOT/variants/libcgcn_$v.so; fi
  echo "== $v"; python tools/khead_acc.py 15182 29910 2>/dev/null
done | tee gpurun_out/r06/head_acc_decomp.txt
