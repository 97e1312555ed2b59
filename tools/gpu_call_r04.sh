#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp15.txt
: > $O
PDEOPT_LIB=$PWD/variants/lib_nbwait.so timeout 600 python -m pytest tests/test_gpu_coop_fixed.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2 >> $O
for r in 1 2 3; do
  for lib in variants/lib_allwait.so variants/lib_nbwait.so; do
    for w in ch_rk4_96_f32_1env ch_rk4_128_f32_1env; do
      PDEOPT_LIB=$PWD/$lib timeout 120 python bench.py --workload $w --no-cpu-baseline --no-parity-spot --no-api --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$w', round(d['ms_per_step'],4), 'ms per 100 substeps', d['config'].get('kernel'))" >> $O 2>&1
    done
  done
done
cat $O
