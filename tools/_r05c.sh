R=$PWD; OUT=$R/gpurun_out/r05c; mkdir -p $OUT
for g in 1 0 1 0; do MSEVB_GRAPH_DEBUG=1 MSEVB_API_TIMING=1 MSEVB_HOST_TIMING=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-thermal --repeats 3 --opt step_graph=$g > $OUT/bench_g$g.json 2> $OUT/bench_g$g.err; python3 -c "
import json;d=json.load(open('$OUT/bench_g$g.json'));print('step_graph=$g', d['value'], d['blocks_ms_per_step'], d['phases_ms_per_step'].get('real space (pair loop)'), d['step_graph'])"; grep -a "msevb api\|host us" $OUT/bench_g$g.err | tail -2; grep -a "captured" $OUT/bench_g$g.err | tail -1; done
MSEVB_GRAPH_DEBUG=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step_graph or schedule_options" -p no:cacheprovider > $OUT/tests.out 2>&1; echo "pytest rc=$?"; tail -5 $OUT/tests.out
