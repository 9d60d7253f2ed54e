#!/bin/bash
# A/B of an environment switch on the genome epoch (tuning tool): ENV_AB="VAR=a VAR=b ..." bash tools/env_ab2.sh [bench args]
run() { python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 40 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f (median %.4f)' % (d['ms_per_step'], d['step_ms']['median']), end='  ')"; }
for rep in $(seq 1 ${AB_REPS:-3}); do
  for kv in $ENV_AB; do echo -n "$kv: "; ( export $kv; run "$@" ); echo; done
done
