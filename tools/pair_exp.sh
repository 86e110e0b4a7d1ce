#!/bin/bash
# Dev container: diagnostic builds of the pair kernel with parts of its loop removed (PF_EXP bits, pair_flat.hip) -> ab_tmp/libE<n>.so.
# On the GPU box:  for n in 0 1 2 ...; do cp ab_tmp/libE$n.so reactive_pb_nn_md_amd/libmsevb_hip.so; python tools/time_stage.py pair --opt prune_freeze=1; done
set -e
cd "$(dirname "$0")/.."
mkdir -p ab_tmp
for n in "$@"; do
  touch reactive_pb_nn_md_amd/csrc/pair_flat.hip
  make -s -C reactive_pb_nn_md_amd/csrc CXXFLAGS_EXTRA="-DPF_EXP=$n" > /dev/null
  cp reactive_pb_nn_md_amd/libmsevb_hip.so ab_tmp/libE$n.so
done
touch reactive_pb_nn_md_amd/csrc/pair_flat.hip
make -s -C reactive_pb_nn_md_amd/csrc > /dev/null
ls -la ab_tmp
