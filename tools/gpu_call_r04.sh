#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp11.txt
: > $O
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2 >> $O
grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head -20 >> $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $O
timeout 400 python tools/small_grid_bench.py ch > gpurun_out/small_grid_ch.txt 2>&1
grep "float32   96^2\|float32  128^2\|float64   64^2\|float64   96^2\|float32   64^2 x   1" gpurun_out/small_grid_ch.txt >> $O
timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200 >> $O
cat $O | cut -c1-220
