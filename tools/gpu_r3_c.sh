cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
rm -f gpurun_out/bench_*.json
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu.log | tail -30
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_head.json 2> gpurun_out/bench_head.err
for mode in "" "--decomp-halo 4"; do
  timeout 300 python bench.py --workload ch_rk4_4096_decomp --steps 5 --warmup 2 $mode >> gpurun_out/bench_decomp_loopback.json 2>> gpurun_out/bench_decomp.err
  timeout 300 python bench.py --workload ch_rk4_4096_decomp --steps 5 --warmup 2 --virtual-ranks 4 $mode >> gpurun_out/bench_decomp_v4.json 2>> gpurun_out/bench_decomp.err
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --workload ch_rk4_4096_decomp --decomp-grid 2048 --steps 5 --warmup 2 $mode >> gpurun_out/bench_decomp_rccl1.json 2>> gpurun_out/bench_decomp.err
done
timeout 300 python bench.py --workload ch_rk4_4096_decomp --decomp-grid 2048 --steps 5 --warmup 2 >> gpurun_out/bench_decomp_loopback2048.json 2>> gpurun_out/bench_decomp.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bench_*.json")):
    for line in open(f):
        line=line.strip()
        if not line.startswith("{"): continue
        d=json.loads(line)
        print(f.split("/")[-1], d["config"].get("workload"), d["config"].get("halo"), "value", round(d["value"],1), "us/substep", d.get("us_per_substep"), "spot", d.get("parity_spot_rel_err"), d.get("parity_spot_max_abs_err"), d.get("parity_spot_ok"), "api", d.get("api_value"))
PY
