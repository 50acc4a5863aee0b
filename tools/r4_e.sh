#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rc=0

timeout -k 10 500 python3 tools/exp_handoff.py "handoff=0" "handoff=1,handoff_wait=0" "handoff=1,handoff_wait=8" "handoff=1,handoff_wait=8,duo_exp=1" "handoff=1,handoff_wait=0,duo_exp=1" "handoff=1,handoff_wait=20,duo_exp=1" "handoff=0,duo_exp=2" "handoff=0,duo_exp=2,split_waves=20" > gpurun_out/r4e_handoff.txt 2>&1; cat gpurun_out/r4e_handoff.txt | grep -v amdgpu.ids
