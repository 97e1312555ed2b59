#!/bin/bash
# kernel-trace stats only.  usage: tools/trace_only.sh <tag> [bench args...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-spot --no-api "$@" > $OUT/trace.log 2>&1
cd $ROOT
python3 tools/summarize_prof.py $OUT | head -24
find $OUT -name "*.csv" -size +3M -delete
