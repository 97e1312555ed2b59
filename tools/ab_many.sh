#!/bin/bash
# interleaved comparison of several library builds: tools/ab_many.sh "<lib1> <lib2> ..." [bench args]
LIBS=$1; shift
for r in 1 2 3; do
  for lib in $LIBS; do
    PDEOPT_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-parity-spot --no-api --steps 5 --warmup 2 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(d['value'],1), 'env-steps/s', round(d['ms_per_step'],2), 'ms/step')"
  done
done
