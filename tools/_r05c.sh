R=$PWD; OUT=$R/gpurun_out/r05c; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "evb_against_reference or translated" -p no:cacheprovider > $OUT/tests.out 2>&1; echo "pytest rc=$?"; tail -3 $OUT/tests.out
bash tools/ab_bench.sh -r 3 -a "--no-phase-timing" base new
for n in base new; do MSEVB_LIB=$R/ab_tmp/lib_$n.so MSEVB_STAMPS=2 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-thermal --repeats 2 --no-phase-timing 2>&1 >/dev/null | grep -a "gpu stamps" | tail -1; done
