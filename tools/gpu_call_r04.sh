#!/bin/bash
# one GPU call of round 4's experiments (outputs under gpurun_out/)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp4.txt
: > $O
echo "== 3-D tests" >> $O
timeout 900 python -m pytest tests/test_gpu_3d.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 >> $O
echo "== 3-D: two-pass kernels (kernel path 1) vs brick kernel" >> $O
for r in 1 2; do
  for kp in 1 0; do
    timeout 300 python bench.py --workload ch3d_rk4_128_f32 --kernel-path $kp --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel path $kp', round(d['value'],1), 'env-steps/s', d.get('parity_spot_ok'), d['config'].get('kernel'), 'avg launch us', round(d['roofline']['avg_launch_us'],1))" >> $O 2>&1
  done
done
echo "== adaptive: previous build (h1noring) vs this build" >> $O
bash tools/ab_adaptive.sh "variants/lib_h1noring.so pde_opt_amd/libpdeopt_hip.so" "ch_sbm_100_tsit5 ch_sbm_100_tsit5_f64 ad_64_tsit5" >> $O 2>&1
echo "== tick profiles" >> $O
for lib in cprof_base cprof_lat3; do
  echo $lib >> $O
  PDEOPT_LIB=$PWD/variants/lib_$lib.so timeout 120 python bench.py --workload ch_sbm_100_tsit5 --steps 1 --warmup 1 --no-cpu-baseline --no-parity-spot 2>&1 | grep "coop prof" | tail -1 >> $O
done
echo "== GPU test suite" >> $O
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3 >> $O
cat $O | cut -c1-220
