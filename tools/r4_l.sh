#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for r in 1 2 3; do
  for o in "wide_tree=1" "wide_tree=2"; do
    DOGERAY_OPTIONS="$o" python3 bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-traffic --no-extras --repeats 5 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$o', 'C4', round(j['kernel_ms_per_frame'],4), round(j['value'],1))"
  done
done 2>&1 | tee gpurun_out/r4l_tree_ab.txt
for o in "wide_tree=1" "wide_tree=2"; do
  for c in C2 C3 C5; do
    DOGERAY_OPTIONS="$o" python3 bench.py --config $c --steps 16 --warmup 4 --no-cpu-baseline --no-traffic --no-extras --repeats 5 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$o', '$c', round(j['kernel_ms_per_frame'],4), round(j['value'],1))"
  done
done 2>&1 | tee -a gpurun_out/r4l_tree_ab.txt
