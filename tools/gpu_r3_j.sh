#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groups.py -q -m gpu -x -k "cahn_hilliard_single_pass or ch_rk4_quad or headline" 2>&1 | tail -3
bash tools/ab_many.sh "pde_opt_amd/libpdeopt_hip.so variants/lib_ch4_nopersist.so" --workload ch_rk4_1024_f32 2>&1 | tee gpurun_out/ab_ch_quad_persist.txt
for gs in 1; do PDEOPT_LIB=$PWD/pde_opt_amd/libpdeopt_hip.so python bench.py --no-cpu-baseline --no-parity-spot --no-api --steps 5 --warmup 2 --group-streams $gs | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('persist group-streams=$gs', round(d['value'],1), 'env-steps/s', round(d['ms_per_step'],2))"; done
