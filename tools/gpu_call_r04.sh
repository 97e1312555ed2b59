#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp10.txt
: > $O
timeout 600 python tools/coop_fixed_tile_sweep.py >> $O 2>&1
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2 >> $O
grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head -20 >> $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O
cat $O | cut -c1-200
