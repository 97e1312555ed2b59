#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp9.txt
: > $O
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2 >> $O
grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head -20 >> $O
grep -n "^E  " gpurun_out/pytest_gpu.log | head -30 | cut -c1-300 >> $O
echo "== 3-D: launch-order block map (previous build) vs XCD slabs" >> $O
for r in 1 2; do
  for lib in variants/lib_cprof_final.so pde_opt_amd/libpdeopt_hip.so; do
    PDEOPT_LIB=$PWD/$lib timeout 300 python bench.py --workload ch3d_rk4_128_f32 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value'],1), 'env-steps/s', d.get('parity_spot_ok'), d['config'].get('kernel'))" >> $O 2>&1
  done
done
cat $O | cut -c1-300
