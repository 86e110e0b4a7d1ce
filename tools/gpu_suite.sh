#!/bin/bash
# GPU box: the -m gpu suite, optionally under the guard allocator (MSEVB_GUARD=1: buffers end on an unmapped page, 2: start on one).
# usage: tools/gpu_suite.sh <tag> [guard-mode] [pytest args...]     output: gpurun_out/suite/<tag>.{out,err}
tag=$1; shift
guard=${1:-0}; shift
mkdir -p gpurun_out/suite
if [ "$guard" != "0" ]; then
  # one worker process that is replaced when a fault kills it (each test still runs once); -s: the runtime's fault message reaches the log
  MSEVB_GUARD=$guard MSEVB_GUARD_LOG=1 timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 1 -s -p no:cacheprovider "$@" > gpurun_out/suite/$tag.out 2> gpurun_out/suite/$tag.err
else
  timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider "$@" > gpurun_out/suite/$tag.out 2> gpurun_out/suite/$tag.err
fi
rc=$?
tail -n 25 gpurun_out/suite/$tag.out
if grep -a -q "Memory access fault" gpurun_out/suite/$tag.err gpurun_out/suite/$tag.out; then
  grep -a "Memory access fault" gpurun_out/suite/$tag.err gpurun_out/suite/$tag.out | sort | uniq -c
  exit 97
fi
exit $rc
