#!/bin/bash
mkdir -p gpurun_out
run() { timeout 300 python bench.py --no-cpu-baseline --no-api --steps 5 --warmup 2 "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', round(d['value'],1), 'env-steps/s', round(d['ms_per_step'],2), 'ms/step', d.get('parity_spot_ok'), d['roofline'].get('concurrent_launches'))"; }
for r in 1 2 3; do
run --workload ch_imex_1024_f32 --group-streams 2
run --workload ch_imex_1024_f32 --group-streams 1
run --workload ch_imex_1024_f32 --group-streams 2 --group-envs 8
done 2>&1 | tee gpurun_out/ab_group_streams_imex.txt
