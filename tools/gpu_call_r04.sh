#!/bin/bash
# one GPU call of round 4's experiments (outputs under gpurun_out/)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp5.txt
: > $O
echo "== adaptive solves of larger periodic grids: multi-workgroup kernel (forced / auto) vs host-driven" >> $O
timeout 900 python tools/adaptive_coop_bench.py 1.0 "CH periodic" f32 >> $O 2>&1
echo "== 3-D tests" >> $O
timeout 900 python -m pytest tests/test_gpu_3d.py tests/test_gpu_adaptive.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 >> $O
cat $O | cut -c1-420
