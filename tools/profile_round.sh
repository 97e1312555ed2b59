#!/bin/bash
# End-of-round evidence run on the GPU box (via gpurun): full PMC profile of the headline workload,
# kernel traces of the secondary workloads, and the default bench line.  Output under gpurun_out/round/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/profile_gpu.sh round/ch_rk4_1024_f32 > /dev/null 2>&1
for w in ac_rk4_512_f32 ch_imex_1024_f32 gpe_strang_512_c64 ch_rk4_1024_f64; do
  bash tools/trace_only.sh round/$w --workload $w > gpurun_out/round/${w}_trace_summary.txt 2>&1
done
for w in ac_rk4_512_f32 ch_imex_1024_f32 gpe_strang_512_c64 ch_rk4_1024_f64; do
  bash tools/pmc_traffic.sh round/pmc_$w --workload $w > /dev/null 2>&1
done
cd $ROOT
# this run's counters -> profiles/pmc_r02.json, which the roofline blocks of the bench lines below read
timeout 300 bash tools/pmc_busy.sh busy > gpurun_out/busy_summary.txt 2>&1
bash tools/make_pmc_json.sh r02
cp profiles/pmc_r02.json gpurun_out/round/pmc_r02.json
python bench.py > gpurun_out/round/bench.json 2> gpurun_out/round/bench.err
for w in ac_rk4_512_f32 ch_imex_1024_f32 gpe_strang_512_c64 gpe_strang_512_c64_spots ch_rk4_1024_f64 ch_rk4_4096_decomp; do
  python bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/round/bench_$w.json
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 \
  --workload ch_rk4_4096_decomp --decomp-grid 2048 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/round/bench_decomp_tile2048_native_rccl_1rank.json
# the driver's multi-GPU launch line, on the one rank a gpurun box has: RCCL barrier + max-over-ranks path of bench.py
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 1 \
  --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/round/bench_headline_torchrun_1rank.json
tail -c 600 gpurun_out/round/bench.json
