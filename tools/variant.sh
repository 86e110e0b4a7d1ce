#!/bin/bash
# Dev container: a variant build of the library in a scratch copy of the sources -> ab_tmp/lib_<name>.so (travels to the GPU box; git-ignored).
#   tools/variant.sh NAME "-DPF_VAR=1 -DPF_EXP=64"      on the GPU box: tools/ab_bench.sh NAME1 NAME2 ... (same box, alternating)
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
b=/tmp/vb/$name
rm -rf $b; mkdir -p $b/reactive_pb_nn_md_amd $root/ab_tmp
cp -r $root/include $b/include
cp -r $root/reactive_pb_nn_md_amd/csrc $b/reactive_pb_nn_md_amd/csrc
rm -f $b/reactive_pb_nn_md_amd/csrc/*.o
make -s -j4 -C $b/reactive_pb_nn_md_amd/csrc CXXFLAGS_EXTRA="$flags" ../libmsevb_hip.so > /dev/null
cp $b/reactive_pb_nn_md_amd/libmsevb_hip.so $root/ab_tmp/lib_$name.so
echo "built ab_tmp/lib_$name.so ($flags)"
