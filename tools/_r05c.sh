R=$PWD; OUT=$R/gpurun_out/r05c; mkdir -p $OUT
for o in evb_fork_once=1 evb_fork_once=0 evb_fork_once=1 evb_fork_once=0 evb_fork_once=1 evb_fork_once=0; do timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-long-run --repeats 3 --no-phase-timing --opt $o 2> $OUT/b.err | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$o', d['value'], d['blocks_ms_per_step'], 'thermal', d['thermalised_box']['value'])"; done
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "schedule_options or md_trajectory" -p no:cacheprovider > $OUT/tests.out 2>&1; echo "pytest rc=$?"; tail -3 $OUT/tests.out
