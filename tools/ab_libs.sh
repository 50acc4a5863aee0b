#!/bin/bash
# Interleaved A/B of library variants on one GPU box: tools/ab_libs.sh "default nofold ..." [rounds] [bench args]
# ("default" = the product library, any other name = tools/_exp/lib_NAME.so); prints kernel ms/frame of every run and the mean.
VARS=$1; ROUNDS=${2:-3}; shift; shift
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    if [ "$v" = default ]; then unset DOGERAY_AMD_LIB; else export DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_$v.so; fi
    python3 bench.py --steps ${STEPS:-32} --warmup 8 --no-cpu-baseline --no-traffic --no-extras --repeats ${REPEATS:-5} "$@" 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', round(j['kernel_ms_per_frame'],4), round(j['value'],1), 'clock %.0f MHz' % j['timed_waves']['shader_clock_mhz'], 'wave-cycles/frame %.4g' % j['timed_waves']['wave_cycles_per_frame'])"
  done
done | tee /tmp/ab_libs.txt
python3 - <<'PY'
from collections import defaultdict
d=defaultdict(list)
for l in open('/tmp/ab_libs.txt'):
    n,ms=l.split()[:2]; d[n].append(float(ms))
for n,v in d.items(): print("mean %-12s %.4f ms/frame over %d runs (min %.4f max %.4f)" % (n, sum(v)/len(v), len(v), min(v), max(v)))
PY
