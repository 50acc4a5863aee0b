#!/bin/bash
# Builds a variant of the library with extra compiler flags for context.hip (e.g. "-DDR_PAD_VALU=40") into tools/_exp/lib_NAME.so,
# reusing the product's host objects (run `python -m dogeray_amd.build` first).  Run with DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_NAME.so.
# usage: tools/exp_variant.sh NAME "-DDR_PAD_VALU=40"
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/_exp /tmp/exp_$NAME
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -enable-post-misched=0 $FLAGS --offload-arch=gfx950 -c $R/dogeray_amd/csrc/context.hip -o /tmp/exp_$NAME/context.o
HOSTOBJ=$(ls $R/dogeray_amd/_build/*.cpp.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/_exp/lib_$NAME.so $HOSTOBJ /tmp/exp_$NAME/context.o -pthread
echo built tools/_exp/lib_$NAME.so
