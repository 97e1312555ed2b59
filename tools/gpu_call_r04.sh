#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp13.txt
: > $O
echo "== headline (CH 1024^2 x 32): environment groups x streams" >> $O
for r in 1 2; do
for ge in 0 4 8 16 -1; do
  for gs in 0 1 2; do
    timeout 120 python bench.py --group-envs $ge --group-streams $gs --no-cpu-baseline --no-parity-spot --no-api --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group-envs $ge streams $gs', round(d['value'],0), 'env-steps/s', d['roofline'].get('concurrent_launches'), round(d['roofline']['avg_launch_us'],1))" >> $O 2>&1
  done
done
done
cat $O
