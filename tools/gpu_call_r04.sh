#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp16.txt
: > $O
PDEOPT_LIB=$PWD/variants/lib_smalldb.so timeout 900 python -m pytest tests/test_gpu_small.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2 >> $O
for r in 1 2 3; do
  for lib in pde_opt_amd/libpdeopt_hip.so variants/lib_smalldb.so; do
    for w in ch_rk4_64_f32_small ac_rk4_64_f32_small; do
      PDEOPT_LIB=$PWD/$lib timeout 120 python bench.py --workload $w --no-cpu-baseline --no-parity-spot --no-api --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$w', round(d['value'],0), 'env-steps/s', round(d['ms_per_step'],4), 'ms', d['config'].get('kernel'))" >> $O 2>&1
    done
  done
done
cat $O
