#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + stats, then PMC passes, of the default bench command.
# Usage: tools/gpu_profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r1}; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 32 --warmup 64 --repeats 5 --no-cpu-baseline --no-traffic --no-extras $@"
python3 bench.py $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/pmc1 -o pmc --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc1.err || { tail -5 $OUT/pmc1.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc2 -o pmc --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc2.err || { tail -5 $OUT/pmc2.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc3.err || { tail -5 $OUT/pmc3.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc4.err || { tail -5 $OUT/pmc4.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc5 -o pmc --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc5.err || { tail -5 $OUT/pmc5.err; echo pmc5 failed; }
find $OUT -name '*.csv' | head -50
