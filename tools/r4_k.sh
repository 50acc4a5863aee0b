#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python3 -c "
import bench
bench.ensure_scene('/tmp/dogeray_bench', 709, 1920, 1080)" > /dev/null 2>&1
S=/tmp/dogeray_bench/heightfield_709_1920x1080.rts
(for o in "" "short_one_queue=0" "coop_steps=1" "coop_rounds=4" "coop_tiles_per_wave=8" "coop_tiles_per_wave=16" "heavy_factor=2" "heavy_factor=-1" "feedback=0"; do echo "== $o"; EXP_OPTIONS="$o" timeout -k 10 200 python3 tools/exp_stripes.py $S 20 8; done) 2>&1 | grep -v amdgpu.ids > gpurun_out/r4k_stripes_opts.txt; cat gpurun_out/r4k_stripes_opts.txt
