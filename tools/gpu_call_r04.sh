#!/bin/bash
# one GPU call of round 4's experiments (outputs under gpurun_out/)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp8.txt
: > $O
echo "== fixed-step multi-workgroup tests" >> $O
timeout 900 python -m pytest tests/test_gpu_coop_fixed.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | tail -15 >> $O
echo "== single-environment latency" >> $O
timeout 300 python tools/single_env_latency.py >> $O 2>&1
echo "== small grids" >> $O
timeout 300 python tools/small_grid_bench.py ch >> $O 2>&1
cat $O | cut -c1-300
