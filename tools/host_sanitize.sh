#!/bin/bash
# Host C++ (reader, BVH builder, lineariser, wide-walk builder, .rtsb) under AddressSanitizer + UBSan on fuzzed, malformed
# and golden scenes, plus BVH arrays with corrupted links (as a hand-made .rtsb or a caller's arrays could carry) (GPU sanitizers are not available on the pool; the device code is covered by the parity tests instead).
# usage: tools/host_sanitize.sh      (from the repo root; exits non-zero on any finding)
set -e
D=${TMPDIR:-/tmp}/dogeray_asan; rm -rf $D; mkdir -p $D/sc
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -ffp-contract=off \
    -I dogeray_amd/csrc -I include -o $D/drv tools/host_sanitize_driver.cpp \
    dogeray_amd/csrc/linearise.cpp dogeray_amd/csrc/wide_builder.cpp dogeray_amd/csrc/rts_reader.cpp dogeray_amd/csrc/bvh_builder.cpp dogeray_amd/csrc/capi_host.cpp -pthread
python3 - "$D" <<'PY'
import sys, os, numpy as np
sys.path.insert(0, 'tests')
from scene_fuzz import random_scene
d = sys.argv[1] + '/sc'
rng = np.random.default_rng(7)
for k in range(30):
    random_scene(rng, int(rng.integers(2, 300)), '%s/f%d.rts' % (d, k), textures=["a.ppm"])
open(d + '/bad1.rts', 'w').write("1,2,x,2\n3,4,5,2\n")
open(d + '/bad2.rts', 'w').write("*,1,2\n1,2,3\n")
open(d + '/one.rts', 'w').write("1,2,3,2,0.5,0.5,0.5,0,0,1,1,1,0\n")
open(d + '/empty.rts', 'w').write("")
PY
ASAN_OPTIONS=detect_leaks=1 $D/drv tests/golden/textures $D/sc/*.rts tests/golden/scenes/*.rts | grep -v ": ok" || true
echo "host_sanitize: clean"
