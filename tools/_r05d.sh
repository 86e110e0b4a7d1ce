R=$PWD; OUT=$R/gpurun_out/r05d; mkdir -p $OUT
for m in "device" "host" "device step_graph=1"; do echo "== $m"; timeout -k 10 500 python3 tools/virtual_rank_times.py 200 B 1,2,4,8 $m 2> $OUT/vr.err | tee "$OUT/vr_B_${m// /_}.txt" | grep "^R="; tail -2 $OUT/vr.err; done
