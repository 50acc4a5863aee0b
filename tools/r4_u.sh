#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4u_gpu_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r4u_gpu_tests.log; tail -3 gpurun_out/r4u_gpu_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python3 tools/phase_budget.py 32 2>&1 | grep -v amdgpu.ids
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4u_trace -o trace --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-extras > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
grep "render_persistent" gpurun_out/r4u_trace/*/*kernel_stats.csv | cut -c1-200; rm -rf gpurun_out/r4u_trace
