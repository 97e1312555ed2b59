#!/bin/bash
# copy the evidence of the last tools/round_end.sh run (gpurun_out/) into profiles/ under a tag:
#   tools/install_profiles.sh r02
set -e
TAG=$1
ROOT=$(cd $(dirname $0)/.. && pwd); cd $ROOT
R=gpurun_out/round
cp $R/bench.json profiles/${TAG}_bench.json
cp $R/ch_rk4_1024_f32/summary.txt profiles/${TAG}_ch_rk4_1024_f32_summary.txt
cp $(ls -t $R/ch_rk4_1024_f32/trace/*/*kernel_stats.csv | head -1) profiles/${TAG}_ch_rk4_1024_f32_kernel_stats.csv
for w in ac_rk4_512_f32 ch_imex_1024_f32 gpe_strang_512_c64 ch_rk4_1024_f64 ch_rk4_64_f32_small ch_rk4_128_f32_small ac_rk4_64_f32_small ch_sbm_1024_f32 ch3d_rk4_128_f32 ch_sbm_100_tsit5 ch_rk4_128_f32_1env decomp_tile2048; do
  [ -f $R/${w}_trace_summary.txt ] && cp $R/${w}_trace_summary.txt profiles/${TAG}_${w}_trace_summary.txt
done
[ -f gpurun_out/small_grid_ch.txt ] && cp gpurun_out/small_grid_ch.txt profiles/${TAG}_small_grid_ch.txt
for f in $R/bench_*.json; do cp $f profiles/${TAG}_$(basename $f); done
cp gpurun_out/busy_summary.txt profiles/${TAG}_pmc_busy.txt
[ -f gpurun_out/valubench.txt ] && cp gpurun_out/valubench.txt profiles/${TAG}_valubench_raw.txt
[ -f gpurun_out/lds_issue_bench.txt ] && cp gpurun_out/lds_issue_bench.txt profiles/${TAG}_lds_issue_bench.txt
[ -f gpurun_out/adaptive_bench.txt ] && cp gpurun_out/adaptive_bench.txt profiles/${TAG}_adaptive_in_kernel.txt
[ -f gpurun_out/adaptive_coop_bench.txt ] && grep -v "coop prof" gpurun_out/adaptive_coop_bench.txt > profiles/${TAG}_adaptive_multi_workgroup.txt
[ -f gpurun_out/ab_adaptive_r04.txt ] && cp gpurun_out/ab_adaptive_r04.txt profiles/${TAG}_adaptive_variants_ab.txt
[ -f gpurun_out/membench_64MiB.txt ] && cat gpurun_out/membench_64MiB.txt > profiles/${TAG}_membench_cache_resident_64MiB.txt
[ -f gpurun_out/peer_mapped_bench.txt ] && cp gpurun_out/peer_mapped_bench.txt profiles/${TAG}_peer_mapped_processes_one_gpu.txt
[ -f gpurun_out/adaptive_coop_larger_grids.txt ] && cp gpurun_out/adaptive_coop_larger_grids.txt profiles/${TAG}_adaptive_larger_grids.txt
[ -f gpurun_out/single_env_latency.txt ] && cp gpurun_out/single_env_latency.txt profiles/${TAG}_single_env_latency.txt
bash tools/make_pmc_json.sh ${TAG}
for f in stencil.hip strang_fused.hip; do python tools/kernel_resources.py $f > /tmp/kres_$f.txt; done
(echo "# hipcc -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py), gfx950, product flags; lds = static LDS only"; \
 echo "## csrc/stencil.hip"; cat /tmp/kres_stencil.hip.txt; echo "## csrc/strang_fused.hip"; cat /tmp/kres_strang_fused.hip.txt) > profiles/${TAG}_kernel_resources.txt
echo "installed profiles/${TAG}_* and profiles/pmc_${TAG}.json"
