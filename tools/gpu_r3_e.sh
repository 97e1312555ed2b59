cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
rm -f gpurun_out/bench_*.json
timeout 600 python -m pytest tests/test_gpu_sbm.py tests/test_gpu_small.py tests/test_gpu_parity.py tests/test_gpu_groups.py -q -m gpu > gpurun_out/pytest_e.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_e.log | tail -30
for w in ch_rk4_64_f32_small ch_rk4_128_f32_small ac_rk4_64_f32_small ch_sbm_1024_f32; do
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
done
timeout 300 python bench.py --workload ch_sbm_1024_f32 --steps 5 --warmup 2 --kernel-path 1 --no-parity-spot > gpurun_out/bench_ch_sbm_generic.json 2> gpurun_out/bench_sbm_generic.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_head.json 2> gpurun_out/bench_head.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bench_*.json")):
    for line in open(f):
        line=line.strip()
        if not line.startswith("{"): continue
        d=json.loads(line)
        print(f.split("/")[-1], "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "kernel", d["config"].get("kernel"), "spot", d.get("parity_spot_rel_err"), d.get("parity_spot_max_abs_err"), d.get("parity_spot_ok"), "api", d.get("api_value"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "roofline", d["roofline"]["bound"], round(d["roofline"]["frac"],3))
PY
tail -n 3 gpurun_out/bench_*.err | tail -20
