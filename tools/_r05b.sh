mkdir -p gpurun_out/r05b
MSEVB_API_TIMING=1 MSEVB_HOST_TIMING=1 python3 bench.py --no-cpu-baseline --no-thermal --repeats 2 > gpurun_out/r05b/api_bench.json 2> gpurun_out/r05b/api.err
grep -a "msevb" gpurun_out/r05b/api.err | tail -8
bash tools/ab_bench.sh -r 2 base v1 env64 | tee gpurun_out/r05b/ab.txt
for n in base v1 e32 e64 e96; do echo -n "$n: "; MSEVB_LIB=$PWD/ab_tmp/lib_$n.so python3 tools/time_stage.py pair --opt prune_freeze=1 --reps 50 2>&1 | tail -1; done | tee gpurun_out/r05b/stage.txt
