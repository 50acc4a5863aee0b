#!/bin/bash
# round 4: the suite (group failure injection included), then the 8-rank one-device rehearsal of dr_group: JSON, kernel trace, reserve_cus 0 / 2 / 4
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4g_gpu_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r4g_gpu_tests.log; tail -5 gpurun_out/r4g_gpu_tests.log
[ $rc -ne 0 ] && exit 1
export DOGERAY_GROUP_DEVICES=0,0,0,0,0,0,0,0
for r in 0 2 4; do
  DOGERAY_RESERVE_CUS=$r timeout -k 10 300 python3 bench.py --gpus 8 --steps 20 --warmup 5 --repeats 5 --gather-every 5 > gpurun_out/r4g_rehearsal_reserve$r.json 2> gpurun_out/r4g_rehearsal_reserve$r.err; echo "reserve $r rc=$?"
  python3 -c "
import json; j=json.load(open('gpurun_out/r4g_rehearsal_reserve$r.json')); print('reserve_cus $r: region %.3f ms, slowest rank kernels %.3f ms, overhead %.1f %%, identical %s, ms/frame %.4f' % (j['region_ms'], max(j['kernel_ms_per_rank']), 100*j['region_overhead_frac'], j['assembled_frame_identical_to_one_context'], j['ms_per_step']))"
done
DOGERAY_RESERVE_CUS=0 timeout -k 10 300 python3 bench.py --gpus 8 --steps 20 --warmup 5 --repeats 5 > gpurun_out/r4g_rehearsal_onebatch.json 2> /dev/null
python3 -c "
import json; j=json.load(open('gpurun_out/r4g_rehearsal_onebatch.json')); print('one gather per region: region %.3f ms, slowest rank kernels %.3f ms, overhead %.1f %%, identical %s' % (j['region_ms'], max(j['kernel_ms_per_rank']), 100*j['region_overhead_frac'], j['assembled_frame_identical_to_one_context']))"
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r4g_trace -o trace --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 8 --steps 20 --warmup 5 --repeats 2 --gather-every 5 > /dev/null 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT && python3 tools/trace_overlap.py gpurun_out/r4g_trace > gpurun_out/r4g_overlap.txt 2>&1; cat gpurun_out/r4g_overlap.txt | tail -30
rm -rf gpurun_out/r4g_trace
