# the k_bwd_sliced vs k_aggregate_sliced gap: in-epoch kernel averages of decomposition builds + one PMC pass
mkdir -p gpurun_out/r06
R=$GRAFT_REPO_ROOT
{
for v in base noriders nodxn nostore nt0 noriders_nt0 bsx_all; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$R/variants/libcgcn_$v.so; fi
  echo "== $v"
  bash tools/kstats.sh gap_$v --no-roofline --steps 10 --warmup 3 | grep "k_bwd_sliced\|k_aggregate_sliced\|k_bwd_rowlocal_ring<true"
done
cd /tmp && export TMPDIR=/tmp
for v in base noriders_nt0; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$R/variants/libcgcn_$v.so; fi
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    d=/tmp/pp_${v}_$(echo $c | tr ' ' '_'); rm -rf $d
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 3 --warmup 1 > $d.log 2>&1
    echo "== pmc $v $c"; python3 $R/tools/pmc_quick.py $d | grep -i "kernel,\|k_bwd_sliced\|k_aggregate"
  done
done
} > $R/gpurun_out/r06/bwd_sliced_gap_raw.txt 2>&1
cat $R/gpurun_out/r06/bwd_sliced_gap_raw.txt
