#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_env.py -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do for lib in pde_opt_amd/libpdeopt_hip.so variants/lib_obs_nosplit.so; do
PDEOPT_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-parity-spot --steps 8 --warmup 2 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib raw', round(d['value'],1), 'api', round(d['api_value'],1), 'api_device_obs', round(d['api_value_device_obs'],1))"
done; done 2>&1 | tee gpurun_out/ab_obs_split.txt
