#!/bin/bash
# memory-path counters of the timed render kernel, one rocprofv3 pass per group (stops at the first pass that fails)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for grp in "$@"; do
  i=$((i+1)); OUT=/tmp/pmcz_$i; rm -rf $OUT
  timeout -k 10 240 rocprofv3 --pmc $grp -d $OUT -o pmc --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-traffic --no-extras > /dev/null 2> /tmp/pmcz_err.txt || { tail -3 /tmp/pmcz_err.txt; echo "pass failed: $grp"; exit 1; }
  python3 - "$OUT" <<'PY' | tee -a $R/gpurun_out/r4z_mem_counters.txt
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_persistent" in r["Kernel_Name"] and "<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print("%-40s launches=%d  mean per launch = %.6g" % (k, len(v), sum(v) / len(v)))
PY
done
