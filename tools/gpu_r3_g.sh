#!/bin/bash
# round 3: in-kernel adaptive solve -- its tests, then every GPU test that drives Tsit5 or the small kernels
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_adaptive.py tests/test_gpu_small.py tests/test_gpu_parity.py tests/test_gpu_sbm.py tests/test_gpu_env.py -q -m gpu 2>&1 | tail -25 > gpurun_out/pytest_g.log
cat gpurun_out/pytest_g.log
