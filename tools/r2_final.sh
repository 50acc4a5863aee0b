#!/bin/bash
# GPU box: everything the round's profiles/ files are made from, in one call (about 6 minutes):
#   default bench line, rocprofv3 trace + PMC passes, other configs, stripes projection, single-frame timeline
mkdir -p gpurun_out
python3 bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || { tail -5 gpurun_out/final_bench.err; exit 1; }
tools/gpu_profile.sh final > gpurun_out/final_profile.log 2>&1 || { tail -5 gpurun_out/final_profile.log; exit 1; }
tools/r2_configs.sh > gpurun_out/final_configs.txt 2>&1
S=/tmp/dogeray_bench/heightfield_709_1920x1080.rts
for K in 20 32 128; do python3 tools/exp_stripes.py $S $K 2>&1 | grep world; done > gpurun_out/final_stripes.txt
python3 tools/exp_timeline.py 1 2>&1 | grep -v "^\[bench\]\|amdgpu.ids" > gpurun_out/final_timeline.txt
python3 tools/exp_single.py "" "split_parts=1" "split_parts=1,coop_steps=0" "feedback_every=1" "short_one_queue=0" 2>&1 | grep -v "^\[bench\]\|amdgpu.ids" > gpurun_out/final_single.txt
echo done
