#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
V=${1:-xasm}
DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_$V.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scene_frames or cube_ladder or fuzz or progressive or rng" > gpurun_out/r4w_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4w_tests.log
[ $rc -eq 0 ] && STEPS=20 REPEATS=10 bash tools/ab_libs.sh "default $V" 6 2>&1 | grep -v amdgpu.ids > gpurun_out/r4w_ab_$V.txt; tail -3 gpurun_out/r4w_ab_$V.txt
