R=$PWD; OUT=$R/gpurun_out/r05h; mkdir -p $OUT
{ for m in "device" "host" "device step_graph=1"; do echo "== mode: $m"; timeout -k 10 500 python3 tools/virtual_rank_times.py 200 B 1,2,4,8 $m 2>/dev/null; done; } > $OUT/virtual_rank_times_B.txt
{ for m in "device" "host" "device step_graph=1"; do echo "== mode: $m"; timeout -k 10 900 python3 tools/virtual_rank_times.py 30 C 1,2,4,8 $m 2>/dev/null; done; } > $OUT/virtual_rank_times_C.txt
grep "^R=\|^==" $OUT/virtual_rank_times_B.txt $OUT/virtual_rank_times_C.txt
bash tools/profile_config_c.sh r05c > $OUT/prof_c.log 2>&1; tail -3 $OUT/prof_c.log
