#!/bin/bash
# GPU box: bench.py (short form) on variant builds of the library, alternating, all on THIS box (boxes of the pool differ by +- 4 %).
#   tools/ab_bench.sh [-r ROUNDS] [-a "extra bench args"] NAME ...      NAME = ab_tmp/lib_NAME.so (tools/variant.sh); "default" = the library in the tree
rounds=2; extra=""
while getopts "r:a:" o; do case $o in r) rounds=$OPTARG;; a) extra=$OPTARG;; esac; done
shift $((OPTIND - 1))
root=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq 1 $rounds); do
  for n in "$@"; do
    lib=$root/ab_tmp/lib_$n.so; [ "$n" = default ] && lib=$root/reactive_pb_nn_md_amd/libmsevb_hip.so
    MSEVB_LIB=$lib timeout -k 10 200 python3 $root/bench.py --no-cpu-baseline --no-thermal --repeats 3 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
p=d['phases_ms_per_step']
print('%-12s %8.1f steps/s  blocks %s  pair %.4f  spread %.4f  evb-real %.4f  blend %.4f' % ('$n', d['value'], d['blocks_ms_per_step'], p.get('real space (pair loop)',0), p.get('Q_grid (spread)',0), p.get('ms-evb diabat real space',0), p.get('ms-evb Hellmann-Feynman blend',0)))
" || echo "$n FAILED"
  done
done
