#!/bin/bash
# round 4, first GPU call: the suite on the cleaned tree, A/B against round 3's library, the single-frame timeline before the hand-off
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4a_gpu_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r4a_gpu_tests.log; tail -3 gpurun_out/r4a_gpu_tests.log
[ $rc -ne 0 ] && exit 1
STEPS=20 REPEATS=5 tools/ab_libs.sh "default r3" 3 > gpurun_out/r4a_ab_cleanup.txt 2>&1; tail -3 gpurun_out/r4a_ab_cleanup.txt
(timeout -k 10 200 python3 tools/exp_timeline.py 1 && FRAMES=16 timeout -k 10 200 python3 tools/exp_single.py "") > gpurun_out/r4a_timeline_before.txt 2>&1; tail -5 gpurun_out/r4a_timeline_before.txt
