#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=20 REPEATS=10 bash tools/ab_libs.sh "default $*" ${ROUNDS:-3} 2>&1 | grep -v amdgpu.ids > gpurun_out/r4y_ab.txt; grep mean gpurun_out/r4y_ab.txt
