#!/bin/bash
# TIMING ONLY: cost attribution of the fused pair kernel by phase ablation
# needs the hooks compiled in:  tools/mkvariant.sh ablate -DPDEOPT_PAIR_ABLATE
# bits: 1 mu passes, 2 marches, 4 ring, 8 tile loads, 16 stores
for r in 1 2; do
for a in ${ABLATE_LIST:-0 1 2 4 7 8 16 24 15 23 31}; do
  PDEOPT_LIB=$PWD/variants/lib_ablate.so python bench.py --no-cpu-baseline --steps 4 --warmup 1 --ablate $a "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ablate $a', round(d['ms_per_step'],2), 'ms/step')"
done; done
