#!/bin/bash
# A/B of the two products forms (CGCN_PRODUCTS=fp32 | split) on one GPU box, interleaved: genome epoch, chr21-like, chr1-like
# and config 1 train steps.  AB_REPS (default 3), AB_WL="genome chr21 chr1 config1".  Prints ms per step.
run() { python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"; }
for rep in $(seq 1 ${AB_REPS:-3}); do
for v in fp32 split; do
  export CGCN_PRODUCTS=$v
  echo -n "$v: "
  for wl in ${AB_WL:-genome chr21 chr1 config1}; do
    case $wl in genome) run;; chr21) run --workload chr21 --steps 200;; chr1) run --workload chr1 --steps 50;; config1) run --workload config1 --steps 200;; esac
  done; echo
done; done
