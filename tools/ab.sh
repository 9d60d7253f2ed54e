#!/bin/bash
# A/B of library variants on one GPU box (tuning tool): AB_VARIANTS="name ..." bash tools/ab.sh compares the in-tree
# build (base) with variants/libcgcn_<name>.so (built with CGCN_EXTRA_FLAGS=... python -c "from chromegcn_amd import
# _build; _build.build_library(out='variants/libcgcn_<name>.so')") on the genome epoch, a chr21-like and a chr1-like
# train step, AB_REPS (default 2) times, interleaved.  AB_WL="genome chr21 chr1" selects the workloads.  Prints ms per step.
run() { python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], end=' ')"; }
for rep in $(seq 1 ${AB_REPS:-2}); do
for v in base ${AB_VARIANTS}; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$GRAFT_REPO_ROOT/variants/libcgcn_$v.so; fi
  echo -n "$v: "
  for wl in ${AB_WL:-genome chr21 chr1}; do
    case $wl in genome) run;; chr21) run --workload chr21 --steps 200;; chr1) run --workload chr1 --steps 50;; esac
  done; echo
done; done
