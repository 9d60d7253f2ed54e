cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in base hsg0; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$R/variants/libcgcn_$v.so; fi
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d=/tmp/pp_${v}_$(echo $c | tr ' ' '_'); rm -rf $d
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 3 --warmup 1 > $d.log 2>&1
    echo "== $v $c"; python3 $R/tools/pmc_quick.py $d | grep -i "kernel,\|k_bwd_sliced\|k_bwd_rowlocal_ring<true\|k_aggregate"
  done
done
