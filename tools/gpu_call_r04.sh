#!/bin/bash
# one GPU call of round 4's experiments (outputs under gpurun_out/)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/exp2.txt
: > $O
echo "== adaptive variants" >> $O
bash tools/ab_adaptive.sh "variants/lib_unroll1.so variants/lib_h1noring.so variants/lib_h1ring.so variants/lib_u1ring.so" "ch_sbm_100_tsit5" >> $O 2>&1
echo "== tick profiles" >> $O
for lib in cprof_base cprof_u2ring; do
  echo $lib >> $O
  PDEOPT_LIB=$PWD/variants/lib_$lib.so timeout 120 python bench.py --workload ch_sbm_100_tsit5 --steps 1 --warmup 1 --no-cpu-baseline --no-parity-spot 2>&1 | grep "coop prof" | tail -1 >> $O
done
echo "== decomposed: 4 virtual ranks, gathered copies (product build) vs in-place reads" >> $O
for r in 1 2; do
  for lib in pde_opt_amd/libpdeopt_hip.so variants/lib_localdirect.so; do
    PDEOPT_LIB=$PWD/$lib timeout 300 python bench.py --workload ch_rk4_4096_decomp --virtual-ranks 4 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step']*10,2), 'us/substep', d.get('parity_spot_ok'), d['config'].get('kernel'))" >> $O 2>&1
  done
done
echo "== decomposed 2048^2 tile loop-back: 32 x 128 tiles vs 64 x 64 tiles" >> $O
for r in 1 2; do
  for tr in 32 64; do
    timeout 300 python bench.py --workload ch_rk4_4096_decomp --decomp-grid 2048 --tile-rows $tr --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile rows $tr', round(d['ms_per_step']*10,2), 'us/substep', d.get('parity_spot_ok'), d['config'].get('kernel'))" >> $O 2>&1
  done
done
echo "== decomposition tests on the in-place build" >> $O
PDEOPT_LIB=$PWD/variants/lib_localdirect.so timeout 900 python -m pytest tests/test_gpu_decomp.py -q -m gpu -x 2>&1 | tail -3 >> $O
cat $O | cut -c1-220
