cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu.log | tail -30
export TMPDIR=/tmp
R=$PWD
cd /tmp
for w in "--decomp-grid 2048" "--decomp-grid 2048 --decomp-halo 4"; do
  tag=$(echo $w | tr -d ' -')
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o t -- python3 $R/bench.py --workload ch_rk4_4096_decomp $w --steps 3 --warmup 1 --no-parity-spot > $R/gpurun_out/prof_$tag.log 2>&1
  f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $w"; head -8 $f | cut -c1-220
done
