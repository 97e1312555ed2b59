#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
PDEOPT_LIB=$PWD/variants/lib_wide1.so timeout 300 python -m pytest tests/test_gpu_decomp.py tests/test_gpu_groups.py -q -m gpu -k "loopback_equals or grouped_equals or full_batch" 2>&1 | tail -3
timeout 900 bash tools/ab_many.sh "variants/lib_wide0.so variants/lib_wide1.so" 2>&1 | tee gpurun_out/ab_wide.txt
