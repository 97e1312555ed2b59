#!/bin/bash
# TIMING ONLY: full kernel (ablate 0) and its memory/LDS skeleton (ablate 7) against the group size
for g in 2 4 8 16 32; do for a in 0 7; do
  python bench.py --no-cpu-baseline --steps 4 --warmup 1 --ablate $a --group-envs $g "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('group $g ablate $a', round(d['ms_per_step'],2), 'ms/step')"
done; done
