#!/bin/bash
# the -m gpu suite, then the bench at the driver's flags with the tree over the triangles' own bounds (default) and over the reference's leaf boxes (wide_tree=1), interleaved
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4ab_gpu_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4ab_gpu_tests.log
[ $rc -ne 0 ] && exit 1
rm -f gpurun_out/r4ab_bench.txt
for r in 1 2 3; do
  for o in "wide_tree=2" "wide_tree=1"; do
    DOGERAY_OPTIONS=$o timeout -k 10 300 python3 bench.py --config ${CONFIG:-C4} --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$o', round(j['kernel_ms_per_frame'],4), round(j['value'],1), 'records/ray %.2f' % j['per_ray']['kernel']['V'])" >> gpurun_out/r4ab_bench.txt || exit 1
  done
done
cat gpurun_out/r4ab_bench.txt
