#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r4ae_single.txt
for r in 1 2; do for v in default "$@"; do
  if [ "$v" = default ]; then unset DOGERAY_AMD_LIB; else export DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_$v.so; fi
  echo "$v $(FRAMES=24 python3 tools/exp_single.py "" 2>&1 | grep -v amdgpu | tail -1)" >> gpurun_out/r4ae_single.txt
done; done
sort gpurun_out/r4ae_single.txt
