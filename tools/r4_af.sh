#!/bin/bash
# the -m gpu suite, then short launches (one frame per launch, the present pipeline with groups of 8 and of 1, an 8-rank stripe launch) for the product and variant libraries
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4af_gpu_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4af_gpu_tests.log
[ $rc -ne 0 ] && exit 1
python3 -c "
import sys; sys.path.insert(0,'.')
import bench; bench.ensure_scene('/tmp/dogeray_bench', 709, 1920, 1080)" > /dev/null
bash tools/ab_single.sh "default $*" 2 2>&1 | grep -v amdgpu.ids > gpurun_out/r4af_short.txt; cat gpurun_out/r4af_short.txt
