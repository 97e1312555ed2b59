#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes for bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-spot --no-api "$@" > $OUT/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-spot --no-api "$@" > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-spot --no-api "$@" > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-spot --no-api "$@" > $OUT/pmc_sq.log 2>&1
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-spot --no-api "$@" > $OUT/pmc_l2.log 2>&1
cd $ROOT
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only the small files (gpurun_out merge limit)
find $OUT -name "*.csv" -size +3M -delete
