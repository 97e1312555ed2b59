#!/bin/bash
# interleaved comparison of library builds on the adaptive workloads: tools/ab_adaptive.sh "<lib1> <lib2> ..." "<workload> ..."
LIBS=$1; WL=${2:-"ch_sbm_100_tsit5 ad_64_tsit5"}
for r in 1 2; do
  for w in $WL; do
    for lib in $LIBS; do
      PDEOPT_LIB=$PWD/$lib timeout 120 python bench.py --no-cpu-baseline --no-parity-spot --steps 5 --warmup 1 --workload $w 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$lib', '$w', round(d['us_per_trial_step'],2), 'us/trial step (wall)', round(d['roofline']['us_per_trial_step_device'],2), 'device', d['config']['trial_steps_per_solve'], d['config']['kernel'])
except Exception as e:
    print('$lib', '$w', 'FAILED', e)"
    done
  done
done
