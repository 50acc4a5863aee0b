#!/bin/bash
# pipelined single frames and the stripes projection, chain on / off
mkdir -p gpurun_out
export TMPDIR=/tmp
python3 -c "
import bench
bench.ensure_scene('/tmp/dogeray_bench', 709, 1920, 1080)" > /dev/null 2>&1
(FRAMES=48 timeout -k 10 300 python3 tools/exp_pipeline.py "handoff=0" "handoff=1" "handoff=1,pipe_streams=3" "handoff=1,handoff_mid=1,handoff_mid_wait=20") 2>&1 | grep -v amdgpu.ids > gpurun_out/r4c_pipeline.txt; cat gpurun_out/r4c_pipeline.txt
S=/tmp/dogeray_bench/heightfield_709_1920x1080.rts
(EXP_OPTIONS="handoff=0" timeout -k 10 200 python3 tools/exp_stripes.py $S 20 1,8; EXP_OPTIONS="handoff=1" timeout -k 10 200 python3 tools/exp_stripes.py $S 20 1,8) 2>&1 | grep -v amdgpu.ids > gpurun_out/r4c_stripes.txt; cat gpurun_out/r4c_stripes.txt
