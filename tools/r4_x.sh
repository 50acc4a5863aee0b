#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4x_gpu_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4x_gpu_tests.log
[ $rc -eq 0 ] && STEPS=20 REPEATS=10 bash tools/ab_libs.sh "default p255" 5 --config C3 2>&1 | grep -v amdgpu.ids > gpurun_out/r4x_ab_c3.txt; tail -3 gpurun_out/r4x_ab_c3.txt
