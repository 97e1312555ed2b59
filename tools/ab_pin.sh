#!/bin/bash
# round 3: which pinning of the flux arithmetic keeps decomposed == monolithic bitwise, and what does it cost?
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for v in 0 1 2 3; do
  echo "== pin$v tests"
  PDEOPT_LIB=$PWD/variants/lib_pin$v.so timeout 600 python -m pytest tests/test_gpu_decomp.py -q -m gpu -k "virtual_ranks or loopback_equals or multi_tile" 2>&1 | tail -8 | grep -E "passed|failed|FAILED" 
done
bash tools/ab_many.sh "variants/lib_pin0.so variants/lib_pin1.so variants/lib_pin2.so variants/lib_pin3.so" 2>&1 | tee gpurun_out/ab_pin.txt
