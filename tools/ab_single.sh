#!/bin/bash
# Interleaved A/B of library variants for SHORT launches (the work-sharing build): one frame per launch, the pipelined present loop, and the slowest
# rank's 20-frame stripe launch of an 8-GPU run:   tools/ab_single.sh "default home0 ..." [rounds]
VARS=${1:-default}; ROUNDS=${2:-2}
S=/tmp/dogeray_bench/heightfield_709_1920x1080.rts
for r in $(seq 1 $ROUNDS); do for v in $VARS; do
  if [ "$v" = default ]; then unset DOGERAY_AMD_LIB; else export DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_$v.so; fi
  echo "== $v"; python3 tools/exp_single.py "" 2>&1 | grep -v amdgpu; FRAMES=48 python3 tools/exp_pipeline.py "" 2>&1 | grep -v amdgpu
  [ -f $S ] && python3 tools/exp_stripes.py $S 20 8 2>&1 | grep world
done; done
