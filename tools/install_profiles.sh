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
for w in ac_rk4_512_f32 ch_imex_1024_f32 gpe_strang_512_c64 ch_rk4_1024_f64; do
  cp $R/${w}_trace_summary.txt profiles/${TAG}_${w}_trace_summary.txt
done
for f in $R/bench_*.json; do cp $f profiles/${TAG}_$(basename $f); done
cp gpurun_out/busy_summary.txt profiles/${TAG}_pmc_busy.txt
[ -f gpurun_out/valubench.txt ] && cp gpurun_out/valubench.txt profiles/${TAG}_valubench_raw.txt
rm -f profiles/pmc_${TAG}.json
python tools/pmc_to_json.py ch_rk4_1024_f32 ${TAG} stage_pair_kernel gpurun_out/busy/pmc_busy gpurun_out/busy/pmc_busy2 \
  $R/ch_rk4_1024_f32/pmc_fetch $R/ch_rk4_1024_f32/pmc_write $R/ch_rk4_1024_f32/pmc_sq $R/ch_rk4_1024_f32/pmc_l2 > /dev/null
# secondary workloads: fabric traffic + VALU issue per launch (tools/pmc_traffic.sh), every kernel of the substep
python tools/pmc_to_json.py ac_rk4_512_f32 ${TAG} "ac_rk4_quad_kernel" $R/pmc_ac_rk4_512_f32/pmc_fetch $R/pmc_ac_rk4_512_f32/pmc_write $R/pmc_ac_rk4_512_f32/pmc_valu > /dev/null
python tools/pmc_to_json.py ch_imex_1024_f32 ${TAG} "stage_pair_kernel|imex_row_|strang_col" $R/pmc_ch_imex_1024_f32/pmc_fetch $R/pmc_ch_imex_1024_f32/pmc_write $R/pmc_ch_imex_1024_f32/pmc_valu > /dev/null
python tools/pmc_to_json.py gpe_strang_512_c64 ${TAG} "strang_row_reg_kernel|strang_col_reg_kernel" $R/pmc_gpe_strang_512_c64/pmc_fetch $R/pmc_gpe_strang_512_c64/pmc_write $R/pmc_gpe_strang_512_c64/pmc_valu > /dev/null
python tools/pmc_to_json.py ch_rk4_1024_f64 ${TAG} "stage_pair_kernel" $R/pmc_ch_rk4_1024_f64/pmc_fetch $R/pmc_ch_rk4_1024_f64/pmc_write $R/pmc_ch_rk4_1024_f64/pmc_valu > /dev/null
for f in stencil.hip strang_fused.hip; do python tools/kernel_resources.py $f > /tmp/kres_$f.txt; done
(echo "# hipcc -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py), gfx950, product flags; lds = static LDS only"; \
 echo "## csrc/stencil.hip"; cat /tmp/kres_stencil.hip.txt; echo "## csrc/strang_fused.hip"; cat /tmp/kres_strang_fused.hip.txt) > profiles/${TAG}_kernel_resources.txt
echo "installed profiles/${TAG}_* and profiles/pmc_${TAG}.json"
