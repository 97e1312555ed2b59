#!/bin/bash
# End-of-round evidence run on the GPU box (via gpurun): full PMC profile of the headline workload, kernel traces and
# PMC traffic of the secondary workloads, and the bench lines.  Output under gpurun_out/round/.  TAG (default r03)
# names profiles/pmc_<TAG>.json.  Every step runs under `timeout`: a hung tool must not eat the round's GPU budget.
TAG=${TAG:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/round
timeout 1500 bash tools/profile_gpu.sh round/ch_rk4_1024_f32 > /dev/null 2>&1
SECONDARY="ac_rk4_512_f32 ch_imex_1024_f32 gpe_strang_512_c64 ch_rk4_1024_f64 ch_rk4_64_f32_small ch_rk4_128_f32_small ac_rk4_64_f32_small ch_sbm_1024_f32 ch3d_rk4_128_f32 ch_sbm_100_tsit5 ch_rk4_128_f32_1env"
for w in $SECONDARY; do
  timeout 400 bash tools/trace_only.sh round/$w --workload $w > gpurun_out/round/${w}_trace_summary.txt 2>&1
done
for w in $SECONDARY; do
  timeout 900 bash tools/pmc_traffic.sh round/pmc_$w --workload $w > /dev/null 2>&1
done
# the decomposed field: one 2048^2 tile (the 2x2 share of config 5), loop-back exchange: what does the kernel cost?
timeout 400 bash tools/trace_only.sh round/decomp_tile2048 --workload ch_rk4_4096_decomp --decomp-grid 2048 > gpurun_out/round/decomp_tile2048_trace_summary.txt 2>&1
# ... and the counters OF THAT WORKLOAD (its own kernel at its own tile size: bench.py's decomposed roofline reads them
# under the key ch_rk4_decomp_tile<nx>x<ny>; round 3 scaled another workload's counters instead)
timeout 900 bash tools/pmc_traffic.sh round/pmc_decomp_tile2048 --workload ch_rk4_4096_decomp --decomp-grid 2048 > /dev/null 2>&1
timeout 900 bash tools/pmc_traffic.sh round/pmc_decomp_tile4096 --workload ch_rk4_4096_decomp > /dev/null 2>&1
cd $ROOT
# this run's counters -> profiles/pmc_<TAG>.json, which the roofline blocks of the bench lines below read
timeout 900 bash tools/pmc_busy.sh busy > gpurun_out/busy_summary.txt 2>&1
timeout 120 bash tools/make_pmc_json.sh $TAG
cp profiles/pmc_$TAG.json gpurun_out/round/pmc_$TAG.json
timeout 600 python bench.py > gpurun_out/round/bench.json 2> gpurun_out/round/bench.err
for w in $SECONDARY gpe_strang_512_c64_spots ch_rk4_1024_f32_cubic ch_rk4_4096_decomp ch_sbm_100_tsit5_f64 ch_sbm_100_tsit5_theta ad_64_tsit5 ch_rk4_96_f32_1env ch_rk4_128_f32_1env; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/round/bench_$w.json
done
timeout 300 python bench.py --workload ch_rk4_4096_decomp --virtual-ranks 4 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/round/bench_decomp_4_virtual_ranks_one_gpu.json
timeout 300 python bench.py --workload ch_rk4_4096_decomp --decomp-grid 2048 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/round/bench_decomp_tile2048_loopback.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 \
  --workload ch_rk4_4096_decomp --decomp-grid 2048 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/round/bench_decomp_tile2048_native_rccl_1rank.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 \
  --workload ch_rk4_4096_decomp --decomp-grid 2048 --decomp-halo 4 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/round/bench_decomp_tile2048_native_rccl_1rank_halo4.json
# the one-process-many-devices mode (VectorPDEEnv(devices=[...])), on the one device a gpurun box has
timeout 300 python bench.py --gpus 1 --single-process --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/round/bench_single_process_1gpu.json
# the reference's notebook-sized adaptive solves: multi-workgroup in-kernel controller vs the host-driven loop
timeout 400 python tools/adaptive_coop_bench.py 0.2 > gpurun_out/adaptive_coop_bench.txt 2>&1
# the driver's multi-GPU launch line, on the one rank a gpurun box has: RCCL barrier + per-rank clocks of bench.py
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 1 \
  --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/round/bench_headline_torchrun_1rank.json
tail -c 600 gpurun_out/round/bench.json
