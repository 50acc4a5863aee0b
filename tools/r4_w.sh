#!/bin/bash
# parity subset on the product library, then an interleaved A/B against variant libraries: tools/r4_w.sh VARIANT...
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scene_frames or cube_ladder or fuzz or progressive or rng or triangle or closest" > gpurun_out/r4w_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4w_tests.log
[ $rc -eq 0 ] && STEPS=20 REPEATS=10 bash tools/ab_libs.sh "default $*" ${ROUNDS:-6} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4w_ab.txt; grep mean gpurun_out/r4w_ab.txt
