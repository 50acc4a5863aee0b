#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 200 python3 tools/phase_budget.py 32 2>&1 | grep -v amdgpu.ids > gpurun_out/r4i_phase_budget.txt; cat gpurun_out/r4i_phase_budget.txt
STEPS=20 REPEATS=5 tools/ab_libs.sh "default xorplain trifirst" 6 > gpurun_out/r4i_ab.txt 2>&1; tail -4 gpurun_out/r4i_ab.txt
