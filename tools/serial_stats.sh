#!/bin/bash
# stand-alone kernel durations of the bench workload (one queue, no graph) on the GPU box:  tools/serial_stats.sh TAG [bench args...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ss_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ss_$TAG/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-thermal --steps 30 --warmup 5 --opt overlap=0 --opt graphs=0 "$@" > /dev/null 2>&1
python3 $R/tools/summarize_prof.py $R/gpurun_out/ss_$TAG > $R/gpurun_out/serial_$TAG.txt 2>&1
rm -rf $R/gpurun_out/ss_$TAG
