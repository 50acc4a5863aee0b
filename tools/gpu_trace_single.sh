#!/bin/bash
# GPU box: kernel trace of one-frame launches (tools/exp_single.py): per-kernel durations and the gaps between consecutive kernels
OUT=$PWD/gpurun_out/trace_single; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 tools/exp_single.py "$@" > $OUT/out.txt 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-60:]          # the last launches: the timed single frames
prev_end = None
for r in rows[-12:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s %9.1f us   gap before %7.1f us" % (r["Kernel_Name"][:60], (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
PY
