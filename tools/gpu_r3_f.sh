cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sbm.py -q -m gpu > gpurun_out/pytest_f.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_f.log | tail -10
timeout 300 python bench.py --workload ch_sbm_1024_f32 --steps 5 --warmup 2 > gpurun_out/bench_ch_sbm_1024_f32.json 2> gpurun_out/bench_ch_sbm_1024_f32.err
timeout 900 bash tools/pmc_traffic.sh round/pmc_ch_sbm_1024_f32 --workload ch_sbm_1024_f32
python tools/pmc_to_json.py ch_sbm_1024_f32 r03tmp "sbm_tiled_kernel" gpurun_out/round/pmc_ch_sbm_1024_f32/pmc_fetch gpurun_out/round/pmc_ch_sbm_1024_f32/pmc_write gpurun_out/round/pmc_ch_sbm_1024_f32/pmc_valu 2>&1 | tail -25
cp profiles/pmc_r03tmp.json gpurun_out/ 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_ch_sbm_1024_f32.json").read().strip().splitlines()[-1])
print("sbm value", d["value"], "ms/step", d["ms_per_step"], "spot", d.get("parity_spot_rel_err"), d.get("parity_spot_max_abs_err"), d.get("parity_spot_ok"), d["roofline"]["avg_launch_us"])
PY
tail -3 gpurun_out/bench_ch_sbm_1024_f32.err
