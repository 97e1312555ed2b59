#!/bin/bash
# profiles/pmc_<tag>.json from the PMC passes under gpurun_out/ (tools/pmc_busy.sh busy, tools/profile_gpu.sh
# round/ch_rk4_1024_f32, tools/pmc_traffic.sh round/pmc_<workload>).  Runs on the GPU box before the bench lines
# are taken (so that their roofline blocks carry this run's counters) and again by tools/install_profiles.sh.
TAG=$1
ROOT=$(cd $(dirname $0)/.. && pwd); cd $ROOT
R=gpurun_out/round
rm -f profiles/pmc_${TAG}.json
python tools/pmc_to_json.py ch_rk4_1024_f32 ${TAG} "ch_rk4_quad_kernel|stage_pair_kernel" gpurun_out/busy/pmc_busy gpurun_out/busy/pmc_busy2 \
  $R/ch_rk4_1024_f32/pmc_fetch $R/ch_rk4_1024_f32/pmc_write $R/ch_rk4_1024_f32/pmc_sq $R/ch_rk4_1024_f32/pmc_l2 > /dev/null 2>&1
# secondary workloads: fabric traffic + VALU issue per launch (tools/pmc_traffic.sh), every kernel of the substep
python tools/pmc_to_json.py ac_rk4_512_f32 ${TAG} "ac_rk4_quad_kernel" $R/pmc_ac_rk4_512_f32/pmc_fetch $R/pmc_ac_rk4_512_f32/pmc_write $R/pmc_ac_rk4_512_f32/pmc_valu > /dev/null 2>&1
python tools/pmc_to_json.py ch_imex_1024_f32 ${TAG} "stage_pair_kernel|imex_row_|strang_col" $R/pmc_ch_imex_1024_f32/pmc_fetch $R/pmc_ch_imex_1024_f32/pmc_write $R/pmc_ch_imex_1024_f32/pmc_valu > /dev/null 2>&1
python tools/pmc_to_json.py gpe_strang_512_c64 ${TAG} "strang_row_reg_kernel|strang_col_reg_kernel" $R/pmc_gpe_strang_512_c64/pmc_fetch $R/pmc_gpe_strang_512_c64/pmc_write $R/pmc_gpe_strang_512_c64/pmc_valu > /dev/null 2>&1
python tools/pmc_to_json.py ch_rk4_1024_f64 ${TAG} "stage_pair_kernel" $R/pmc_ch_rk4_1024_f64/pmc_fetch $R/pmc_ch_rk4_1024_f64/pmc_write $R/pmc_ch_rk4_1024_f64/pmc_valu > /dev/null 2>&1
python tools/pmc_to_json.py ch_rk4_64_f32_small ${TAG} "small_persist_kernel" $R/pmc_ch_rk4_64_f32_small/pmc_fetch $R/pmc_ch_rk4_64_f32_small/pmc_write $R/pmc_ch_rk4_64_f32_small/pmc_valu > /dev/null 2>&1
python tools/pmc_to_json.py ch_rk4_128_f32_small ${TAG} "small_persist_kernel" $R/pmc_ch_rk4_128_f32_small/pmc_fetch $R/pmc_ch_rk4_128_f32_small/pmc_write $R/pmc_ch_rk4_128_f32_small/pmc_valu > /dev/null 2>&1
python tools/pmc_to_json.py ac_rk4_64_f32_small ${TAG} "small_persist_kernel" $R/pmc_ac_rk4_64_f32_small/pmc_fetch $R/pmc_ac_rk4_64_f32_small/pmc_write $R/pmc_ac_rk4_64_f32_small/pmc_valu > /dev/null 2>&1
python tools/pmc_to_json.py ch_sbm_1024_f32 ${TAG} "sbm_tiled_kernel" $R/pmc_ch_sbm_1024_f32/pmc_fetch $R/pmc_ch_sbm_1024_f32/pmc_write $R/pmc_ch_sbm_1024_f32/pmc_valu > /dev/null 2>&1
python tools/pmc_to_json.py ch3d_rk4_128_f32 ${TAG} "ch3d_mu_kernel|ch3d_stage_kernel" $R/pmc_ch3d_rk4_128_f32/pmc_fetch $R/pmc_ch3d_rk4_128_f32/pmc_write $R/pmc_ch3d_rk4_128_f32/pmc_valu > /dev/null 2>&1
python tools/pmc_to_json.py ch_sbm_100_tsit5 ${TAG} "tsit5_coop_kernel" $R/pmc_ch_sbm_100_tsit5/pmc_fetch $R/pmc_ch_sbm_100_tsit5/pmc_write $R/pmc_ch_sbm_100_tsit5/pmc_valu > /dev/null 2>&1
python tools/pmc_to_json.py ch_rk4_128_f32_1env ${TAG} "tsit5_coop_kernel" $R/pmc_ch_rk4_128_f32_1env/pmc_fetch $R/pmc_ch_rk4_128_f32_1env/pmc_write $R/pmc_ch_rk4_128_f32_1env/pmc_valu > /dev/null 2>&1
# the decomposed field's own kernel(s) at the tile sizes a gpurun box can run (one rank: 2048^2 = the 2 x 2 share of config 5; 4096^2 whole)
python tools/pmc_to_json.py ch_rk4_decomp_tile2048x2048 ${TAG} "ch_rk4_quad_kernel|stage_pair_kernel" $R/pmc_decomp_tile2048/pmc_fetch $R/pmc_decomp_tile2048/pmc_write $R/pmc_decomp_tile2048/pmc_valu > /dev/null 2>&1
python tools/pmc_to_json.py ch_rk4_decomp_tile4096x4096 ${TAG} "ch_rk4_quad_kernel|stage_pair_kernel" $R/pmc_decomp_tile4096/pmc_fetch $R/pmc_decomp_tile4096/pmc_write $R/pmc_decomp_tile4096/pmc_valu > /dev/null 2>&1
