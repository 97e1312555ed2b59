#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_groups.py -q -m gpu -k "ac or allen or AC or fuzz or quad" 2>&1 | tail -4
bash tools/ab_many.sh "pde_opt_amd/libpdeopt_hip.so variants/lib_ac4_nohelp.so" --workload ac_rk4_512_f32 2>&1 | tee gpurun_out/ab_ac4_helpers.txt
