#!/bin/bash
# Builds a variant of the library with extra compiler flags for the device code (e.g. "-DDOGERAY_EXPERIMENTAL", "-DDR_PAD_VALU=40") into
# tools/_exp/lib_NAME.so, reusing the product's host objects (run `python -m dogeray_amd.build` first).
# Run with DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_NAME.so.
# usage: tools/exp_variant.sh NAME "-DDOGERAY_EXPERIMENTAL"      (SCHED="" tools/exp_variant.sh NAME "": with the post-RA machine scheduler the product build switches off)
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/_exp /tmp/exp_$NAME
OBJS=""
for f in kernels_render kernels_aux; do
  /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize ${SCHED--mllvm -enable-post-misched=0 -mllvm -amdgpu-use-amdgpu-trackers=1} $FLAGS --offload-arch=gfx950 -c $R/dogeray_amd/csrc/$f.hip -o /tmp/exp_$NAME/$f.o &
  OBJS="$OBJS /tmp/exp_$NAME/$f.o"
done
wait
HOSTOBJ=$(ls $R/dogeray_amd/_build/*.cpp.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/_exp/lib_$NAME.so $HOSTOBJ $OBJS -pthread -ldl
echo built tools/_exp/lib_$NAME.so
