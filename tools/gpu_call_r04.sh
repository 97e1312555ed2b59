#!/bin/bash
# one GPU call of round 4's experiments (outputs under gpurun_out/)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp6.txt
: > $O
export PDEOPT_LIB=$PWD/variants/lib_nofence.so
echo "== multi-XCD environments WITHOUT release / acquire fences (scoped accesses only): tests" >> $O
timeout 900 python -m pytest tests/test_gpu_coop_adaptive.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 >> $O
echo "== larger grids, no fences" >> $O
timeout 600 python tools/adaptive_coop_bench.py 1.0 "CH periodic" f32 2>&1 | cut -c1-60,150-330 >> $O
echo "== notebook solve on more workgroups (tile edge forced), no fences" >> $O
for tile in 0 15 13 11; do
  for r in 1 2; do
    PDEOPT_COOP_TILE=$tile timeout 120 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --workload ch_sbm_100_tsit5 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('tile $tile', round(d['us_per_trial_step'],2), 'us/trial step', d.get('parity_spot_ok'), d['config']['kernel'])
except Exception as e:
    print('tile $tile FAILED', e)" >> $O 2>&1
  done
done
cat $O | cut -c1-300
