#!/bin/bash
# like ab_libs.sh, one round, with the counting build's diagnostics (cycles, clock)
for v in $1; do
  if [ "$v" = default ]; then unset DOGERAY_AMD_LIB; else export DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_$v.so; fi
  echo "== $v"; STEPS=32 WARM=8 tools/ab.sh "feedback=1"
done
