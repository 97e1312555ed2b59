#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cahn_hilliard_single_pass" 2>&1 | tail -3
bash tools/ab_many.sh "pde_opt_amd/libpdeopt_hip.so variants/lib_ch4_mu1.so" --workload ch_rk4_1024_f32 2>&1 | tee gpurun_out/ab_ch_quad_mu2.txt
