#!/bin/bash
# copy the evidence of the last tools/profile_round.sh + tools/pmc_busy.sh run (gpurun_out/) into profiles/
# under a new version tag, dropping the previous one:  tools/install_profiles.sh <old> <new>   e.g. v9 v10
set -e
OLD=$1; NEW=$2
ROOT=$(cd $(dirname $0)/.. && pwd); cd $ROOT
R=gpurun_out/round
git rm -q --cached profiles/r01_${OLD}_* 2>/dev/null || true
rm -f profiles/r01_${OLD}_*
cp $R/bench.json profiles/r01_${NEW}_bench.json
cp $R/ch_rk4_1024_f32/summary.txt profiles/r01_${NEW}_ch_rk4_1024_f32_summary.txt
cp $(ls -t $R/ch_rk4_1024_f32/trace/*/*kernel_stats.csv | head -1) profiles/r01_${NEW}_ch_rk4_1024_f32_kernel_stats.csv
for w in ac_rk4_512_f32 ch_imex_1024_f32 gpe_strang_512_c64 ch_rk4_1024_f64; do
  cp $R/${w}_trace_summary.txt profiles/r01_${NEW}_${w}_trace_summary.txt
  cp $R/bench_$w.json profiles/r01_${NEW}_bench_$w.json
done
for f in bench.py DESIGN.md README.md profiles/README.md profiles/traffic.json; do
  sed -i "s/r01_${OLD}_/r01_${NEW}_/g" $f
done
echo "installed profiles/r01_${NEW}_*  (pmc_busy.txt is written by hand from gpurun_out/busy*)"
