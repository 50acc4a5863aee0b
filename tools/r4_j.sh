#!/bin/bash
# round 4: the default bench line with the per-class counters; C5 bench line + stripes projection on C5 (BASELINE config 5: 3840x2160, "8 x MI355X with RCCL tile gather")
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4j_gpu_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r4j_gpu_tests.log; tail -3 gpurun_out/r4j_gpu_tests.log
[ $rc -ne 0 ] && exit 1
(FRAMES=64 timeout -k 10 300 python3 tools/exp_pipeline.py "pipe_group=1" "pipe_group=2" "pipe_group=4" "pipe_group=8" "pipe_group=16" "pipe_group=8,pipe_streams=3") 2>&1 | grep -v amdgpu.ids > gpurun_out/r4j_pipeline_groups.txt; cat gpurun_out/r4j_pipeline_groups.txt
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4j_bench.json 2> gpurun_out/r4j_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r4j_bench.err
python3 -c "
import json; j=json.load(open('gpurun_out/r4j_bench.json')); r=j['roofline']; print(j['value'], j['ms_per_step'], r['frac'], {k:v for k,v in r['valu'].items() if 'issue' in k or 'sq_' in k}); print(r['valu'].get('classes_per_launch')); print(j.get('single_frame_ms'), j.get('single_frame_pipelined_ms'))"
timeout -k 10 400 python3 bench.py --config C5 --steps 16 --warmup 4 --repeats 5 --no-traffic --no-cpu-baseline > gpurun_out/r4j_bench_C5.json 2> gpurun_out/r4j_bench_C5.err; echo "C5 rc=$?"
python3 -c "
import json; j=json.load(open('gpurun_out/r4j_bench_C5.json')); print('C5', j['value'], j['ms_per_step'], j.get('single_frame_ms'), j.get('single_frame_pipelined_ms'))"
S=/tmp/dogeray_bench/city_200_3840x2160.rts
(timeout -k 10 400 python3 tools/exp_stripes.py $S 20 1,2,4,8; timeout -k 10 400 python3 tools/exp_stripes.py $S 128 1,8) 2>&1 | grep -v amdgpu.ids > gpurun_out/r4j_stripes_C5.txt; cat gpurun_out/r4j_stripes_C5.txt
S4=/tmp/dogeray_bench/heightfield_709_1920x1080.rts
(timeout -k 10 300 python3 tools/exp_stripes.py $S4 20 1,2,4,8; timeout -k 10 300 python3 tools/exp_stripes.py $S4 128 1,8) 2>&1 | grep -v amdgpu.ids > gpurun_out/r4j_stripes_C4.txt; cat gpurun_out/r4j_stripes_C4.txt
