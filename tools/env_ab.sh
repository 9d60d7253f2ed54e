run() { python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"; }
for rep in 1 2; do
echo -n "base: "; run; run --workload chr21 --steps 200; echo
echo -n "DEV_KERNARG=1: "; HIP_FORCE_DEV_KERNARG=1 run; HIP_FORCE_DEV_KERNARG=1 run --workload chr21 --steps 200; echo
echo -n "DEV_KERNARG=0: "; HIP_FORCE_DEV_KERNARG=0 run; HIP_FORCE_DEV_KERNARG=0 run --workload chr21 --steps 200; echo
echo -n "PACKET_CAPTURE=1: "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 run; DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 run --workload chr21 --steps 200; echo
echo -n "PACKET_CAPTURE=0: "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run --workload chr21 --steps 200; echo
echo -n "GPU_MAX_HW_QUEUES=1: "; GPU_MAX_HW_QUEUES=1 run; GPU_MAX_HW_QUEUES=1 run --workload chr21 --steps 200; echo
echo -n "AMD_SERIALIZE_KERNEL=0 DIRECT_DISPATCH=1: "; AMD_DIRECT_DISPATCH=1 run; AMD_DIRECT_DISPATCH=1 run --workload chr21 --steps 200; echo
done
