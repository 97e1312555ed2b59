#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/round
w=ch_rk4_128_f32_1env
timeout 400 bash tools/trace_only.sh round/$w --workload $w > gpurun_out/round/${w}_trace_summary.txt 2>&1
timeout 900 bash tools/pmc_traffic.sh round/pmc_$w --workload $w > /dev/null 2>&1
python tools/pmc_to_json.py $w r04 "tsit5_coop_kernel" gpurun_out/round/pmc_$w/pmc_fetch gpurun_out/round/pmc_$w/pmc_write gpurun_out/round/pmc_$w/pmc_valu | tail -15
cp profiles/pmc_r04.json gpurun_out/round/pmc_r04_with_1env.json
timeout 300 python bench.py --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/round/bench_$w.json
head -8 gpurun_out/round/${w}_trace_summary.txt | cut -c1-160
python -c "
import json; l=json.loads(open('gpurun_out/round/bench_$w.json').read().strip().splitlines()[-1]); print(l['value'], l['roofline'])" | cut -c1-1500
