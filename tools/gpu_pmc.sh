#!/bin/bash
# tools/gpu_pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ... ; env DOGERAY_VARIANT respected
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp -d $OUT/p$i -o pmc --output-format csv -- python3 bench.py --steps 32 --warmup 32 --repeats 1 --no-cpu-baseline --no-traffic --no-extras > /dev/null 2> $OUT/p$i.err || { tail -3 $OUT/p$i.err; echo "pass $i failed: $grp"; }
done
python3 - <<PY
import csv, glob, os
from collections import defaultdict
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "render_" not in k or "<true" in k: continue
        for cn, vals in sorted(cs.items()):
            print("%-28s %-40s n=%d mean=%.6g" % (k[10:38], cn, len(vals), sum(vals)/len(vals)))
PY
