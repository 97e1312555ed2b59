#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
bash tools/ab_many.sh "pde_opt_amd/libpdeopt_hip.so variants/lib_cols8.so" --workload ch_imex_1024_f32 > gpurun_out/exp14.txt 2>&1
cat gpurun_out/exp14.txt
