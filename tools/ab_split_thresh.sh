run() { python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"; }
for rep in 1 2; do
for v in default 0 4194304; do
  if [ $v = default ]; then unset CGCN_FWD_SPLIT_BYTES; else export CGCN_FWD_SPLIT_BYTES=$v; fi
  echo -n "split_bytes=$v: "
  run; run --workload chr21 --steps 200; run --workload config1 --steps 200; echo
done; done
