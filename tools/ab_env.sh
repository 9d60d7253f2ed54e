run() { python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"; }
for rep in 1 2; do
echo -n "default: "; run; echo
echo -n "split8MiB: "; CGCN_FWD_SPLIT_BYTES=8388608 run; echo
echo -n "split0: "; CGCN_FWD_SPLIT_BYTES=0 run; echo
echo -n "fp32: "; CGCN_PRODUCTS=fp32 run; echo
echo -n "fp32+8MiB: "; CGCN_PRODUCTS=fp32 CGCN_FWD_SPLIT_BYTES=8388608 run; echo
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
