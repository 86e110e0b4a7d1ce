#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + two separate PMC passes of the bench command.
# Output under gpurun_out/prof_$1 ; copy the summaries you want to keep into profiles/.
set -e
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f -- $CMD > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w -- $CMD > $OUT/write.log 2>&1
python3 $R/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
# keep the merge small: drop the raw traces
find $OUT -name '*kernel_trace.csv' -size +20M -delete || true
ls -la $OUT $OUT/* | head -40
