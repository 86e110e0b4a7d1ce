R=$PWD; OUT=$R/gpurun_out/r05c; mkdir -p $OUT
for o in evb_grid_first=0 evb_grid_first=1 evb_grid_first=0 evb_grid_first=1 step_graph=1 "step_graph=1 --opt evb_grid_first=1"; do timeout -k 10 200 python3 bench.py --no-cpu-baseline --repeats 3 --no-phase-timing --opt $o 2> $OUT/b.err | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$o', d['value'], d['blocks_ms_per_step'], 'thermal', d['thermalised_box']['value'], d['step_graph']['graph_launches'])"; done
MSEVB_STAMPS=2 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-thermal --repeats 2 --no-phase-timing --opt evb_grid_first=1 2>&1 >/dev/null | grep -a "gpu stamps" | tail -1
