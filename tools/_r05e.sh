OUT=gpurun_out/r05e; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gathered_hamiltonian or sharding or rccl_communicator or sharded_principal" -p no:cacheprovider > $OUT/tests.out 2>&1; echo "pytest rc=$?"; tail -30 $OUT/tests.out
