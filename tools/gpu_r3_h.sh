#!/bin/bash
# round 3: persistent, software-pipelined Allen-Cahn single-pass kernel -- parity tests, then A/B against the committed kernel
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_small.py tests/test_gpu_env.py -q -m gpu -k "ac or allen or AC or fuzz or quad or env" 2>&1 | tail -8 > gpurun_out/pytest_h.log
cat gpurun_out/pytest_h.log
timeout 900 bash tools/ab_many.sh "pde_opt_amd/libpdeopt_hip.so variants/lib_ac4_base.so variants/lib_ac4_wg2.so" --workload ac_rk4_512_f32 2>&1 | tee gpurun_out/ab_ac4_persist.txt
