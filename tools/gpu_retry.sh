#!/bin/bash
# Retries a gpurun call while the pool has no free slot (exit code 3: nothing charged). Usage: tools/gpu_retry.sh <timeout-s> '<command>'
# Runs from the repository root of the BUILD container (not on the GPU box).
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
