#!/bin/bash
# configs[4]-size profile set on the GPU box (985 k atoms, 16 diabats, 128^3):  tools/profile_config_c.sh TAG  ->  gpurun_out/prof_TAG/
set -e
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-thermal --repeats 1 --n-side 69 --diabats 16 --pme-grid 128"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $B --steps 10 --warmup 3 > $OUT/kt.log 2>&1
python3 $R/tools/summarize_prof.py $OUT > $OUT/rocprof_summary.txt 2>&1 || true
python3 $R/tools/timeline.py $OUT > $OUT/timeline.txt 2>&1 || true
rm -rf $OUT/kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $B --steps 10 --warmup 3 --opt overlap=0 --opt graphs=0 > $OUT/kts.log 2>&1
python3 $R/tools/summarize_prof.py $OUT > $OUT/serial_kernel_stats.txt 2>&1 || true
rm -rf $OUT/kt
$B --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || true
$B --steps 30 --warmup 5 --opt fft_own=0 > $OUT/bench_rocfft.json 2> $OUT/bench_rocfft.err || true
rm -f $OUT/*.log
ls -la $OUT
