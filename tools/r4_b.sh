#!/bin/bash
# round 4: the suite with the hand-off chain, the batched rate against round 3's library, single frames per option set with the chain's timeline
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4b_gpu_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r4b_gpu_tests.log; tail -5 gpurun_out/r4b_gpu_tests.log
[ $rc -ne 0 ] && exit 1
STEPS=20 REPEATS=5 tools/ab_libs.sh "default r3" 3 > gpurun_out/r4b_ab_batched.txt 2>&1; tail -3 gpurun_out/r4b_ab_batched.txt
timeout -k 10 400 python3 tools/exp_handoff.py "handoff=0" "handoff=1" "handoff=1,short_one_queue=0" "handoff=1,handoff_mid=1,handoff_mid_wait=12" "handoff=1,handoff_mid=1,handoff_mid_wait=30" "handoff=1,handoff_mid=2,handoff_mid_wait=20" > gpurun_out/r4b_handoff.txt 2>&1; cat gpurun_out/r4b_handoff.txt | grep -v amdgpu.ids
