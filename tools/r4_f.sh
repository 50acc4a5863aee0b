#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python3 -c "
import bench
bench.ensure_scene('/tmp/dogeray_bench', 709, 1920, 1080)" > /dev/null 2>&1
S=/tmp/dogeray_bench/heightfield_709_1920x1080.rts
(for o in "handoff=0" "handoff=1,handoff_wait=0" "handoff=1,handoff_wait=8" "handoff=1,handoff_wait=8,short_one_queue=0" "handoff=1,handoff_wait=24"; do echo "== $o"; EXP_OPTIONS="$o" timeout -k 10 200 python3 tools/exp_stripes.py $S 20 8; done) 2>&1 | grep -v amdgpu.ids > gpurun_out/r4f_stripes.txt; cat gpurun_out/r4f_stripes.txt
(FRAMES=48 timeout -k 10 300 python3 tools/exp_pipeline.py "handoff=0" "handoff=1,handoff_wait=0" "handoff=1,handoff_wait=8" "handoff=0,pipe_lean=1") 2>&1 | grep -v amdgpu.ids > gpurun_out/r4f_pipeline.txt; cat gpurun_out/r4f_pipeline.txt
