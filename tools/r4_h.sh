#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4h_gpu_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r4h_gpu_tests.log; tail -3 gpurun_out/r4h_gpu_tests.log
[ $rc -ne 0 ] && exit 1
DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_trifirst.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scene_frames or fuzz or closest_hit or cube_ladder or work_sharing" > gpurun_out/r4h_trifirst_tests.log 2>&1; echo "trifirst tests rc=$?"; tail -2 gpurun_out/r4h_trifirst_tests.log
timeout -k 10 200 python3 tools/phase_budget.py 32 2>&1 | grep -v amdgpu.ids > gpurun_out/r4h_phase_budget.txt; cat gpurun_out/r4h_phase_budget.txt
STEPS=20 REPEATS=5 tools/ab_libs.sh "default xorplain trifirst r3" 3 > gpurun_out/r4h_ab.txt 2>&1; tail -5 gpurun_out/r4h_ab.txt
