#!/bin/bash
# what the driver runs at round end: smoke(), then bench.py with its default flags (and time it)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4q_smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r4q_smoke.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "pipelined or group" > gpurun_out/r4q_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r4q_tests.log
SECONDS=0
timeout -k 10 900 python3 bench.py > gpurun_out/r4q_bench_default.json 2> gpurun_out/r4q_bench_default.err; echo "bench (no flags) rc=$? in $SECONDS s"
python3 -c "
import json; j=json.load(open('gpurun_out/r4q_bench_default.json')); print(j['value'], j['ms_per_step'], j['steps'], j['warmup'], j['roofline']['frac'], j['single_frame_ms'], j['single_frame_pipelined_ms'], j['cpu_baseline']['value'])"
