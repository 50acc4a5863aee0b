#!/bin/bash
# VGPRs / spills / LDS of every kernel of one translation unit (compiles with -save-temps into /tmp/kres):
#   tools/kernel_resources.sh [unit=kernels_render] [grep pattern] [extra flags]
R=$(cd "$(dirname "$0")/.." && pwd)
U=${1:-kernels_render}
rm -rf /tmp/kres && mkdir -p /tmp/kres && cd /tmp/kres
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -enable-post-misched=0 -mllvm -amdgpu-use-amdgpu-trackers=1 $3 --offload-arch=gfx950 -c $R/dogeray_amd/csrc/$U.hip -o u.o -save-temps 2>/dev/null
python3 - "$2" "$U" <<'PY'
import re, sys
t = open('/tmp/kres/%s-hip-amdgcn-amd-amdhsa-gfx950.s' % sys.argv[2]).read()
pat = sys.argv[1] if len(sys.argv) > 1 else ''
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size:', t, re.S):
    b = m.group(0)
    g = lambda k: re.search(r'\.%s:\s*(\S+)' % k, b).group(1)
    name = g('name')
    if pat and not re.search(pat, name): continue
    print("%-90s vgpr %s spill %s sgpr %s lds %s scratch %s" % (name[:90], g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
PY
