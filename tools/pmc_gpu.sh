#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass per counter group (arguments: TAG "C1 C2 .." "C3 C4 .." ...).
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for grp in "$@"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$n -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline $BENCH_ARGS > $OUT/g$n.log 2>&1
done
python3 $R/tools/summarize_pmc.py $OUT $PMC_ONLY > $OUT/summary.txt 2>&1 || true
find $OUT -name "*.db" -delete || true      # gpurun merges at most 64 MiB back: keep the summary, drop the databases
tail -60 $OUT/summary.txt
