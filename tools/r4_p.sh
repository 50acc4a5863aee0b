#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4p_gpu_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r4p_gpu_tests.log; tail -4 gpurun_out/r4p_gpu_tests.log
