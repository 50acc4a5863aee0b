#!/bin/bash
# the stripes projection on C4 again, with the own-bounds tree (tools/exp_stripes.py: every rank's stripe in turn on ONE MI355X)
mkdir -p gpurun_out
export TMPDIR=/tmp
python3 -c "
import sys; sys.path.insert(0,'.')
import bench; print(bench.ensure_scene('/tmp/dogeray_bench', 709, 1920, 1080))" > /dev/null
S4=/tmp/dogeray_bench/heightfield_709_1920x1080.rts
(timeout -k 10 300 python3 tools/exp_stripes.py $S4 20 1,2,4,8 && timeout -k 10 300 python3 tools/exp_stripes.py $S4 128 1,8) 2>&1 | grep -v amdgpu.ids > gpurun_out/r4ad_stripes_C4.txt; cat gpurun_out/r4ad_stripes_C4.txt
