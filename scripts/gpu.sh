#!/bin/bash
# gpurun with patience: exit code 3 = no box / slot free right now (nothing charged) -> wait and ask again, up to 12 times.
# usage: bash scripts/gpu.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 150
done
exit 3
