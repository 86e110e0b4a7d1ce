#!/bin/bash
# Runs on the GPU box (through gpurun): everything profiles/rNN_* is made of, as text (the rocprofv3 databases are dropped before gpurun
# merges gpurun_out/ back: 64 MiB limit).   usage: tools/profile_round.sh TAG
set -e
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-thermal --repeats 1"
# 1. kernel trace of the overlapped (default) schedule + timelines of a step without / with a list rebuild
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $B --steps 30 --warmup 5 > $OUT/kt.log 2>&1
python3 $R/tools/summarize_prof.py $OUT > $OUT/rocprof_summary.txt 2>&1 || true
python3 $R/tools/timeline.py $OUT > $OUT/timeline.txt 2>&1 || true
python3 $R/tools/timeline.py $OUT rebuild > $OUT/timeline_rebuild.txt 2>&1 || true
rm -rf $OUT/kt
# 2. the same kernels alone on the GPU (one queue, no graph): stand-alone durations
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $B --steps 30 --warmup 5 --opt overlap=0 --opt graphs=0 > $OUT/kts.log 2>&1
python3 $R/tools/summarize_prof.py $OUT > $OUT/serial_kernel_stats.txt 2>&1 || true
python3 $R/tools/pair_modes.py $OUT > $OUT/pair_modes.txt 2>&1 || true      # pair kernel: launches that read their list / that rewrite it
rm -rf $OUT/kt
if [ -n "$QUICK" ]; then rm -f $OUT/*.log; ls -la $OUT; exit 0; fi      # QUICK=1: the two kernel traces only
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md: they do not fit one pass), default schedule
# (with counters the profiler runs ONE kernel at a time, in the order they become ready: a kernel that waits on the device for a mark another queue's
#  kernel sets would wait until its 2 s limit - evb_publish_spin=0 orders those kernels with events instead; same kernels, same bytes)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f -- $B --steps 10 --warmup 2 --opt evb_publish_spin=0 > $OUT/fetch.log 2>&1 && rcf=0 || rcf=$?; echo "fetch pass rc=$rcf" >> $OUT/pmc_exit_codes.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w -- $B --steps 10 --warmup 2 --opt evb_publish_spin=0 > $OUT/write.log 2>&1 && rcw=0 || rcw=$?; echo "write pass rc=$rcw" >> $OUT/pmc_exit_codes.txt
python3 $R/tools/summarize_prof.py $OUT > $OUT/traffic_summary.txt 2>&1 || true
# (a failed counter pass leaves NO traffic.json behind - a stale one from an earlier round must not pass for this build's)
if [ "$rcf" = "0" ] && [ "$rcw" = "0" ]; then python3 $R/tools/make_traffic_json.py $OUT $OUT/traffic.json > /dev/null 2> $OUT/traffic_json.err || echo "make_traffic_json failed" >> $OUT/pmc_exit_codes.txt; else echo "traffic.json not made: a counter pass failed" >> $OUT/pmc_exit_codes.txt; fi
rm -rf $OUT/fetch $OUT/write
# 4. SQ / TCP counters of the dominant kernels, stand-alone
PMC_ONLY="k_pair k_pme_spread k_pme_gather" BENCH_ARGS="--no-thermal --opt overlap=0 --opt graphs=0" bash $R/tools/pmc_gpu.sh ${TAG}_sq \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAVES" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum" > /dev/null 2>&1 || true
cp $R/gpurun_out/pmc_${TAG}_sq/summary.txt $OUT/pmc_sq_summary.txt 2>/dev/null || true
# 5. GPU-side stamps of an unprofiled run, and the bench line itself (with the CPU baseline and the thermalised-box line)
MSEVB_STAMPS=1 $B --steps 200 --warmup 20 > /dev/null 2> $OUT/stamps.err || true
grep "msevb gpu stamps" $OUT/stamps.err > $OUT/gpu_stamps.txt || true
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || true
rm -f $OUT/*.log $OUT/stamps.err
ls -la $OUT
