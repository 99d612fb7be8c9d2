#!/bin/bash
# gpurun with retries while no GPU slot is free (exit code 3 = nothing charged): tools/gpurun_retry.sh TIMEOUT 'command'
T="$1"; shift
for i in $(seq 12); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 75
done
exit 3
