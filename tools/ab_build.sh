#!/bin/bash
# Dev container: ab_tmp/libA.so = the library of the last commit, ab_tmp/libB.so = the working tree's.  Then on the GPU box:
#   for v in A B A B; do cp ab_tmp/lib$v.so reactive_pb_nn_md_amd/libmsevb_hip.so; python bench.py ...; done
# (boxes of the pool differ by +-3 %: only runs on the SAME box compare)
set -e
cd "$(dirname "$0")/.."
mkdir -p ab_tmp
git stash -q
make -C reactive_pb_nn_md_amd/csrc > /dev/null
cp reactive_pb_nn_md_amd/libmsevb_hip.so ab_tmp/libA.so
git stash pop -q
make -C reactive_pb_nn_md_amd/csrc > /dev/null
cp reactive_pb_nn_md_amd/libmsevb_hip.so ab_tmp/libB.so
ls -la ab_tmp
