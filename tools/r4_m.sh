#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_sgpr.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scene_frames or stripes_partition or progressive" > gpurun_out/r4m_sgpr_tests.log 2>&1; echo "sgpr tests rc=$?"; tail -2 gpurun_out/r4m_sgpr_tests.log
STEPS=20 REPEATS=5 tools/ab_libs.sh "default sgpr" 5 > gpurun_out/r4m_ab.txt 2>&1; tail -3 gpurun_out/r4m_ab.txt
