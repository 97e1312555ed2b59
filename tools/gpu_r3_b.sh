cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_small.py tests/test_gpu_decomp.py tests/test_gpu_env.py -q -m gpu -x > gpurun_out/pytest_b.log 2>&1
tail -30 gpurun_out/pytest_b.log
timeout 600 python tools/small_grid_bench.py ch > gpurun_out/small_grid_ch.txt 2>&1
cat gpurun_out/small_grid_ch.txt
timeout 300 python tools/small_grid_bench.py ac > gpurun_out/small_grid_ac.txt 2>&1
cat gpurun_out/small_grid_ac.txt | head -30
