#!/bin/bash
# fabric traffic + VALU issue of one secondary workload: three separate rocprofv3 --pmc passes (counters never
# share a pass with a trace).  usage: tools/pmc_traffic.sh <tag> --workload <w>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-parity-spot --no-api"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS "$@" > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS "$@" > $OUT/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_valu -- python3 $ROOT/bench.py $ARGS "$@" > $OUT/pmc_valu.log 2>&1
cd $ROOT
find $OUT -name "*.csv" -size +3M -delete
ls $OUT/*/*/*counter_collection.csv 2>/dev/null | wc -l
