#!/bin/bash
# interleaved A/B of the fused CH pair kernel at 16-row (256 threads) and 32-row (512 threads) tiles
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ragged or explicit_trajectory or batch_equals or mass_conservation" 2>&1 | tail -3
for r in 1 2 3; do
  for rows in 16 32; do
    python bench.py --no-cpu-baseline --steps 5 --warmup 2 --tile-rows $rows "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows$rows', round(d['value'],1), 'env-steps/s', round(d['ms_per_step'],2), 'ms/step', d['config'].get('kernel'))"
  done
done
