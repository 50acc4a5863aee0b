#!/bin/bash
# Sensitivity experiment: builds a copy of the library whose node step carries extra, useless work --
#   -DDR_PAD_VALU=n  n VALU instructions      -DDR_PAD_SALU=n  n SALU instructions
#   -DDR_PAD_VMEM=n  n more 16-byte fetches per lane, 64 B on from the node (-DDR_PAD_UNIFORM=1: one address for all lanes)
# into tools/_exp/lib_NAME.so; run it with DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_NAME.so python3 bench.py ...
# The product sources are not touched (the patch is applied to a copy under /tmp/exp).
# usage: tools/exp_pad_variant.sh NAME "-DDR_PAD_VALU=8"
set -e
NAME=$1; FLAGS=$2
D=/tmp/exp/$NAME; rm -rf $D; mkdir -p $D /root/repo/tools/_exp
mkdir -p $D/pkg $D/include; cp -r /root/repo/dogeray_amd/csrc $D/pkg/csrc; cp /root/repo/include/dogeray_amd.h $D/include/
# patch: pads in trav_step_park
python3 - "$D" <<'PY'
import sys
d=sys.argv[1]
p=d+'/pkg/csrc/device_core.hpp'
s=open(p).read()
old='''  if (leaf) { pk.C = ld_unit_raw(walk, off + 32); pk.D = ld_unit_raw(walk, off + 48); }
  float mn[3], mx[3] = {B.x, B.y, B.z};'''
old2='''  if (leaf) { pk.C = ld_unit_raw(walk, off + 32); pk.D = ld_unit_raw(walk, off + 48); }\n  float mn[3] = {A.x, A.y, A.z}, mx[3] = {B.x, B.y, B.z};'''
assert old2 in s
s=s.replace(old2,'''  if (leaf) { pk.C = ld_unit_raw(walk, off + 32); pk.D = ld_unit_raw(walk, off + 48); }
#if DR_PAD_VMEM
  u32x4 padm[DR_PAD_VMEM];
  _Pragma("unroll") for (int k = 0; k < DR_PAD_VMEM; k++) padm[k] = ld_unit_raw(walk, DR_PAD_UNIFORM ? 4096u * (unsigned)(k + 1) : off + 64u * (unsigned)(k + 1));
#endif
  float mn[3] = {A.x, A.y, A.z}, mx[3] = {B.x, B.y, B.z};''',1)
old3='''  pk.parked = h && leaf;
  tr.node = (h && !leaf) ? w0 : next_miss;
}'''
assert old3 in s
s=s.replace(old3,'''  pk.parked = h && leaf;
  tr.node = (h && !leaf) ? w0 : next_miss;
#if DR_PAD_VALU
  _Pragma("unroll") for (int k = 0; k < DR_PAD_VALU; k++) asm volatile("v_or_b32 %0, 0, %0" : "+v"(tr.best_slot));
#endif
#if DR_PAD_SALU
  { int sx = 0; _Pragma("unroll") for (int k = 0; k < DR_PAD_SALU; k++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx) :: "scc"); asm volatile("" :: "s"(sx)); }
#endif
#if DR_PAD_VMEM
  _Pragma("unroll") for (int k = 0; k < DR_PAD_VMEM; k++) asm volatile("" :: "v"(padm[k]));
#endif
}''',1)
s=s.replace('#pragma once','#pragma once\n#ifndef DR_PAD_VALU\n#define DR_PAD_VALU 0\n#endif\n#ifndef DR_PAD_SALU\n#define DR_PAD_SALU 0\n#endif\n#ifndef DR_PAD_VMEM\n#define DR_PAD_VMEM 0\n#endif\n#ifndef DR_PAD_UNIFORM\n#define DR_PAD_UNIFORM 0\n#endif\n',1)
open(p,'w').write(s)
PY
cd $D
C="-std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math"
for f in rts_reader bvh_builder linearise capi_host; do /opt/rocm/bin/hipcc $C -x c++ -c pkg/csrc/$f.cpp -o $f.o & done
/opt/rocm/bin/hipcc $C -fno-slp-vectorize $FLAGS --offload-arch=gfx950 -c pkg/csrc/context.hip -o context.o
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/tools/_exp/lib_$NAME.so rts_reader.o bvh_builder.o linearise.o capi_host.o context.o -pthread
echo built $NAME
