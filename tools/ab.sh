#!/bin/bash
# A/B of library variants on one GPU box (tuning tool): AB_VARIANTS="name ..." bash tools/ab.sh compares the in-tree
# build (base) with variants/libcgcn_<name>.so on five workloads (chr21, config1, hic-like, chr1, d=256 L=4), twice.
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"; }
for rep in 1 2; do
for v in base ${AB_VARIANTS}; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$GRAFT_REPO_ROOT/variants/libcgcn_$v.so; fi
  echo -n "$v: "; run --steps 200 --warmup 10; run --workload config1 --steps 100 --warmup 5; run --hic-like --steps 100 --warmup 5; run --workload chr1 --steps 50 --warmup 5; run --d 256 --layers 4 --steps 30 --warmup 5; echo
done; done
