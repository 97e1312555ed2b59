#!/bin/bash
# keep asking for a GPU slot until the call is accepted (exit 3 = none free, nothing charged): tools/gpu_retry.sh <timeout-s> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
