#!/bin/bash
# tools/r2_quick.sh <tag> [DOGERAY_OPTIONS variants...]: parity smoke (wide-walk subset), default bench with diag, then A/B variants
TAG=$1; shift
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "closest_hit or scene_frames or fuzzed or ties or cube_ladder or tuning" > gpurun_out/${TAG}_parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/${TAG}_parity.log
tail -3 gpurun_out/${TAG}_parity.log
STEPS=32 WARM=4 tools/ab.sh "feedback=1 $*" 2>&1 | tee gpurun_out/${TAG}_ab.txt
