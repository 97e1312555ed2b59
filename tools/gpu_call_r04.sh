#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp12.txt
: > $O
echo "== config 2 (AC 512^2 x 64): environment groups x streams" >> $O
for r in 1 2; do
for ge in 0 8 16 32 -1; do
  for gs in 0 1 2; do
    timeout 120 python bench.py --workload ac_rk4_512_f32 --group-envs $ge --group-streams $gs --no-cpu-baseline --no-parity-spot --no-api --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group-envs $ge streams $gs', round(d['value'],0), 'env-steps/s', d['config'].get('kernel'), d['roofline'].get('concurrent_launches'), round(d['roofline']['avg_launch_us'],1))" >> $O 2>&1
  done
done
done
cat $O
