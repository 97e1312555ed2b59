#!/bin/bash
# usage: tools/sweep.sh "<common bench args>" "<arg name>" v1 v2 ...   (runs bench.py per value, prints one line each)
COMMON=$1; NAME=$2; shift 2
for v in "$@"; do
  python bench.py --no-cpu-baseline --no-parity-spot --no-api $COMMON $NAME $v 2>&1 | tail -1 | python -c "
import json,sys
line=sys.stdin.read()
try:
    d=json.loads(line)
    print('$NAME=$v', d['config']['kernel'], round(d['value'],1), 'env-steps/s', round(d['achieved_gbs_whole_job']), 'GB/s algorithmic', round(d['ms_per_step'],2), 'ms/step')
except Exception as e:
    print('$NAME=$v FAILED', line[-300:])
"
done
