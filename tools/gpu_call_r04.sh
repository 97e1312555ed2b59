#!/bin/bash
# one GPU call of round 4's experiments (outputs under gpurun_out/)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp7.txt
: > $O
echo "== GPU test suite" >> $O
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3 >> $O
echo "== adaptive workloads, this build" >> $O
bash tools/ab_adaptive.sh "pde_opt_amd/libpdeopt_hip.so" "ch_sbm_100_tsit5 ch_sbm_100_tsit5_theta ch_sbm_100_tsit5_f64 ad_64_tsit5" >> $O 2>&1
echo "== tick profile" >> $O
PDEOPT_LIB=$PWD/variants/lib_cprof_final.so timeout 120 python bench.py --workload ch_sbm_100_tsit5 --steps 1 --warmup 1 --no-cpu-baseline --no-parity-spot 2>&1 | grep "coop prof" | tail -1 >> $O
echo "== all notebook-sized cases" >> $O
timeout 900 python tools/adaptive_coop_bench.py 1.0 "" 2>&1 | cut -c1-64,150-330 >> $O
cat $O | cut -c1-300
