R=$PWD; OUT=$R/gpurun_out/r05c; mkdir -p $OUT
for o in overlap=1 overlap=1; do timeout -k 10 200 python3 bench.py --no-cpu-baseline --repeats 3 --no-phase-timing --opt $o 2> $OUT/b.err | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$o', d['value'], d['blocks_ms_per_step'], 'thermal', d['thermalised_box']['value'], 'long', d['long_run'])"; done
