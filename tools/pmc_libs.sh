#!/bin/bash
# PMC counters of the timed render kernel for library variants on one box: tools/pmc_libs.sh "default nofold" "CTR1 CTR2 ..." ["more counters" ...]
VARS=$1; shift
export TMPDIR=/tmp
for v in $VARS; do
  if [ "$v" = default ]; then unset DOGERAY_AMD_LIB; else export DOGERAY_AMD_LIB=$PWD/tools/_exp/lib_$v.so; fi
  i=0
  for grp in "$@"; do
    i=$((i+1)); OUT=/tmp/pmc_$v_$i; rm -rf $OUT
    timeout -k 10 150 rocprofv3 --pmc $grp -d $OUT -o pmc --output-format csv -- python3 bench.py --steps 32 --warmup 32 --no-cpu-baseline --no-traffic > /dev/null 2> /tmp/pmc_err.txt || { tail -3 /tmp/pmc_err.txt; echo "pass failed: $grp"; }
    python3 - "$OUT" "$v" <<'PY'
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_" in r["Kernel_Name"] and "<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], "  ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(agg.items())))
PY
  done
done
