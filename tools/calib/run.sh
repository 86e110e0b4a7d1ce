#!/bin/bash
# Runs on the GPU box: FETCH_SIZE and WRITE_SIZE (separate passes) of tools/calib/calib_fetch, counter / known bytes per kernel.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o p -- $R/tools/calib/calib_fetch > $OUT/$c.log 2>&1
done
python3 $R/tools/calib/summarize.py $OUT > $OUT/calibration.txt 2>&1 || true
find $OUT -name "*.db" -delete || true
cat $OUT/calibration.txt
