#!/bin/bash
# Everything the round's evidence comes from, in one gpurun call:
#   gpurun --timeout 3000 -- 'timeout 2900 bash tools/round_end.sh 2>&1 | tail -12'
# GPU test suite -> tools/profile_round.sh (bench lines, kernel traces, PMC traffic) -> tools/pmc_busy.sh (pipe
# occupancy of the headline kernel) -> tools/valubench + its PMC calibration.  Afterwards, here: tools/install_profiles.sh <tag>.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -2
timeout 2400 bash tools/profile_round.sh
[ -x tools/valubench.bin ] && timeout 200 ./tools/valubench.bin > gpurun_out/valubench.txt 2>&1
[ -x tools/lds_issue_bench.bin ] && timeout 200 ./tools/lds_issue_bench.bin > gpurun_out/lds_issue_bench.txt 2>&1
timeout 300 python tools/adaptive_bench.py > gpurun_out/adaptive_bench.txt 2>&1
timeout 300 python tools/single_env_latency.py > gpurun_out/single_env_latency.txt 2>&1
timeout 300 python tools/small_grid_bench.py ch > gpurun_out/small_grid_ch.txt 2>&1
# the peer-mapped exchange with one PROCESS per rank, all on this GPU (2 and 4 ranks of config 5's tile size)
(timeout 300 python tools/peer_mapped_bench.py 2 1 2048 200; timeout 300 python tools/peer_mapped_bench.py 2 2 2048 200) > gpurun_out/peer_mapped_bench.txt 2>&1
timeout 600 python tools/adaptive_coop_bench.py 1.0 "CH periodic" f32 > gpurun_out/adaptive_coop_larger_grids.txt 2>&1
tail -4 gpurun_out/busy_summary.txt | cut -c1-400
