#!/bin/bash
# kernel trace only (fast): gpurun_out/prof_$1/kt
set -e
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline $BENCH_ARGS > $OUT/kt.log 2>&1
