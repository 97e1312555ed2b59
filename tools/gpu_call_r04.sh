#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/round
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2
grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head
for w in ch_sbm_100_tsit5 ch_sbm_100_tsit5_theta ch_sbm_100_tsit5_f64 ad_64_tsit5 ch_rk4_96_f32_1env ch_rk4_128_f32_1env; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/round/bench_$w.json
  python -c "
import json; l=json.loads(open('gpurun_out/round/bench_$w.json').read().strip().splitlines()[-1]); print('$w', round(l['value']), round(l['ms_per_step'],3), l.get('us_per_trial_step',''), l.get('parity_spot_ok'))"
done
timeout 300 python tools/small_grid_bench.py ch > gpurun_out/small_grid_ch.txt 2>&1
grep "float32   96^2 x   1\|float32  128^2 x   1\|float32   96^2 x  16\|float64   64^2 x   1" gpurun_out/small_grid_ch.txt | cut -c1-170
