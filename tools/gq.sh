#!/bin/bash
# queueing wrapper around gpurun: retries ONLY when no box/slot was free (exit code 3: nothing ran, nothing charged)
# usage: tools/gq.sh <timeout_s> '<command>'
T=$1; shift
for k in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
