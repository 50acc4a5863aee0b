#!/bin/bash
# Everything profiles/r3_r_* are made from, in one call on the GPU box: tools/r3_final.sh
# (full -m gpu suite, the default bench line with its PMC passes, the rocprofv3 kernel-trace summary of the same command, the other configs, the stripes projection)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3r_gpu_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3r_gpu_tests.log; tail -3 gpurun_out/r3r_gpu_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r3r_bench.json 2> gpurun_out/r3r_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r3r_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3r_trace -o trace --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-extras > gpurun_out/r3r_bench_traced.json 2> gpurun_out/r3r_trace.err; echo "trace rc=$?"
for c in C2 C3 C5; do
  timeout -k 10 300 python3 bench.py --config $c --steps 16 --warmup 4 --repeats 5 --no-traffic --no-cpu-baseline > gpurun_out/r3r_bench_$c.json 2> gpurun_out/r3r_bench_$c.err; echo "$c rc=$?"
done
S=$(ls /tmp/dogeray_bench/heightfield_709_1920x1080.rts 2>/dev/null)
if [ -n "$S" ]; then
  (timeout -k 10 200 python3 tools/exp_stripes.py $S 20 1,2,4,8; timeout -k 10 200 python3 tools/exp_stripes.py $S 32 1,8) > gpurun_out/r3r_stripes.txt 2>&1; tail -6 gpurun_out/r3r_stripes.txt
fi
