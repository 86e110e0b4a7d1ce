#!/bin/bash
# Runs on the GPU box: kernel trace of the bench with every kernel ALONE on the GPU (--opt overlap=0 --opt graphs=0): stand-alone durations.
# usage: tools/prof_serial.sh TAG [extra bench args]
set -e
TAG=${1:-x}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --opt overlap=0 --opt graphs=0 "$@" > $OUT/kt.log 2>&1
python3 $R/tools/summarize_prof.py $OUT > $OUT/serial_summary.txt 2>&1 || true
#find $OUT -name '*kernel_trace.csv' -size +20M -delete || true
python3 $R/tools/pair_modes.py $OUT > $OUT/pair_modes.txt 2>&1 || true
head -45 $OUT/serial_summary.txt; cat $OUT/pair_modes.txt
rm -rf $OUT/kt
