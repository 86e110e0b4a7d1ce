R=$PWD; OUT=$R/gpurun_out/r05c; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "slabs" -p no:cacheprovider > $OUT/tests.out 2>&1; echo "pytest rc=$?"; tail -12 $OUT/tests.out
for o in spread_slabs=1 spread_slabs=0 spread_slabs=1 spread_slabs=0; do timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-long-run --repeats 3 --opt $o 2> $OUT/b.err | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());p=d['phases_ms_per_step'];print('$o', d['value'], d['blocks_ms_per_step'], 'pair', p.get('real space (pair loop)'), 'spread', p.get('Q_grid (spread)'), 'thermal', d['thermalised_box']['value'])"; done
for o in spread_slabs=1 spread_slabs=0; do MSEVB_STAMPS=2 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-thermal --repeats 2 --no-phase-timing --opt $o 2>&1 >/dev/null | grep -a "gpu stamps" | tail -1; done
python3 tools/time_stage.py pme --opt spread_slabs=1 --opt overlap=0 --opt graphs=0 | tail -1; python3 tools/time_stage.py pme --opt spread_slabs=0 --opt overlap=0 --opt graphs=0 | tail -1
