#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1000 python3 tools/fuzz_campaign.py ${FUZZ_SCENES:-6000} ${FUZZ_SEED:-50001} > gpurun_out/r4o_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r4o_fuzz.txt
