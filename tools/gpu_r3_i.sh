#!/bin/bash
# round 3: side-by-side environment groups -- tests, then the default bench line
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_groups.py tests/test_gpu_env.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -8 > gpurun_out/pytest_i.log
cat gpurun_out/pytest_i.log
timeout 600 python bench.py > gpurun_out/bench_default_i.json 2> gpurun_out/bench_default_i.err; tail -c 1500 gpurun_out/bench_default_i.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_default_i.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], "api", d.get("api_value"), "spot", d.get("parity_spot_ok"), d.get("parity_spot_rel_err"))
print({k:r[k] for k in ("bound","achieved","frac","avg_launch_us","concurrent_launches","traffic_gbs","traffic_frac","valu_util","launches_timed") if k in r})
PY
