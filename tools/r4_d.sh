#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4d_gpu_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r4d_gpu_tests.log; tail -5 gpurun_out/r4d_gpu_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python3 tools/exp_handoff.py "handoff=0" "handoff=1,handoff_wait=0" "handoff=1,handoff_wait=4" "handoff=1,handoff_wait=8" "handoff=1,handoff_wait=16" "handoff=1,handoff_wait=30" "handoff=1,handoff_wait=8,short_one_queue=0" > gpurun_out/r4d_handoff.txt 2>&1; cat gpurun_out/r4d_handoff.txt | grep -v amdgpu.ids
(FRAMES=48 timeout -k 10 300 python3 tools/exp_pipeline.py "handoff=0" "handoff=1,handoff_wait=0" "handoff=1,handoff_wait=8" "handoff=1,handoff_wait=16") 2>&1 | grep -v amdgpu.ids > gpurun_out/r4d_pipeline.txt; cat gpurun_out/r4d_pipeline.txt
