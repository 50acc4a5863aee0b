#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
FRAMES=24 timeout -k 10 600 python3 tools/exp_single.py "" "split_waves=8" "split_waves=16" "split_waves=20" "split_steps=304" "split_steps=512" "split_parts=2" "split_parts=8" "coop_rounds=3" "coop_steps=1" "coop_steps=3" "feedback_every=4" "feedback_every=16" "heavy_factor=2" "" 2>&1 | grep -v amdgpu.ids > gpurun_out/r4v_single_sweep.txt; cat gpurun_out/r4v_single_sweep.txt
