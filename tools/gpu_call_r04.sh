#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/round
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2
grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head
for w in ch_rk4_64_f32_small ac_rk4_64_f32_small ch_rk4_128_f32_small; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/round/bench_$w.json
  python -c "
import json; l=json.loads(open('gpurun_out/round/bench_$w.json').read().strip().splitlines()[-1]); print('$w', round(l['value']), l.get('parity_spot_ok'), l['config']['kernel'])"
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
