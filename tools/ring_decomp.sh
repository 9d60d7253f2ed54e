#!/bin/bash
# Decomposition of k_bwd_rowlocal_ring (tuning tool): isolated launch times of the plain and the head form with the row team
# alone (-DRING_SKIP_MFMA), the matrix team alone (-DRING_SKIP_ROWTEAM) and without flag waits (-DRING_NO_WAIT; garbage
# results, pure co-run interference).  Variants: python tools/mkvariant.py rowonly='-DCGCN_EXPERIMENT_BUILD -DRING_SKIP_MFMA=1' matonly='-DCGCN_EXPERIMENT_BUILD -DRING_SKIP_ROWTEAM=1' nowait='-DCGCN_EXPERIMENT_BUILD -DRING_NO_WAIT=1'
for v in base rowonly matonly nowait base; do
  if [ $v = base ]; then unset CHROMEGCN_LIB; else export CHROMEGCN_LIB=$GRAFT_REPO_ROOT/variants/libcgcn_$v.so; fi
  for wl in chr1 genome; do
    python bench.py --no-cpu-baseline --no-extras --steps 10 --workload $wl 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_us']
print('$v $wl', {x:k[x] for x in k if 'ring' in x}, round(d['ms_per_step'],4))"
  done
done
