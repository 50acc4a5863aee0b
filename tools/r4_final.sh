#!/bin/bash
# Everything profiles/r4_z_* are made from, in one call on the GPU box: tools/r4_final.sh
# (full -m gpu suite, the default bench line with its PMC passes, the rocprofv3 kernel-trace summary of the same command, the other configs, a fuzz campaign)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4z_gpu_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r4z_gpu_tests.log; tail -3 gpurun_out/r4z_gpu_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4z_bench.json 2> gpurun_out/r4z_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r4z_bench.err
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4z_trace -o trace --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-traffic --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/r4z_bench_traced.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4z_trace.err; echo "trace rc=$?"; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r4z && rm -rf gpurun_out/prof_r4z/trace && mv gpurun_out/r4z_trace gpurun_out/prof_r4z/trace && python3 tools/summarise_prof.py gpurun_out/prof_r4z > gpurun_out/r4z_rocprofv3_summary.txt 2>&1; head -12 gpurun_out/r4z_rocprofv3_summary.txt; rm -rf gpurun_out/prof_r4z
for c in C2 C3 C5; do
  timeout -k 10 400 python3 bench.py --config $c --steps 16 --warmup 4 --repeats 5 --no-traffic --no-cpu-baseline > gpurun_out/r4z_bench_$c.json 2> gpurun_out/r4z_bench_$c.err; echo "$c rc=$?"
done
timeout -k 10 900 python3 tools/fuzz_campaign.py ${FUZZ_SCENES:-800} 40004 > gpurun_out/r4z_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r4z_fuzz.txt
FUZZ_SIZES=8,9,16,24,33,40 timeout -k 10 400 python3 tools/fuzz_campaign.py ${FUZZ_TINY:-200} 40005 > gpurun_out/r4z_fuzz_tiny.txt 2>&1; echo "fuzz tiny rc=$?"; tail -1 gpurun_out/r4z_fuzz_tiny.txt
