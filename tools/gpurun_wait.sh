#!/bin/bash
# gpurun with patience: exit code 3 = no slot free (nothing ran, nothing charged) -> wait and ask again.
# usage: tools/gpurun_wait.sh <timeout-seconds> '<command>'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
