#!/usr/bin/env python3
"""Condenses a tools/gpu_profile.sh output directory into one text summary (for profiles/)."""
import csv, glob, json, os, sys
from collections import defaultdict
d = sys.argv[1]
out = []
for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    out.append("== kernel stats (%s)" % os.path.relpath(f, d))
    out.append(open(f).read().strip())
for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    per = defaultdict(list)
    for r in rows:
        per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    out.append("== kernel trace: per-kernel launches, avg ms, min, max; VGPR/SGPR/LDS of first launch")
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        r0 = next(r for r in rows if r["Kernel_Name"] == k)
        out.append("%-70s n=%d avg=%.4f min=%.4f max=%.4f  vgpr=%s agpr=%s sgpr=%s lds=%s scratch=%s wg=%s grid=%s" % (
            k[:70], len(v), sum(v) / len(v), min(v), max(v), r0.get("VGPR_Count"), r0.get("Accum_VGPR_Count"), r0.get("SGPR_Count"),
            r0.get("LDS_Block_Size"), r0.get("Scratch_Size"), r0.get("Workgroup_Size"), r0.get("Grid_Size")))
for p in sorted(glob.glob(os.path.join(d, "pmc*"))):
    if not os.path.isdir(p):
        continue
    for f in sorted(glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(lambda: defaultdict(list))
        for r in rows:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.append("== PMC %s (mean per launch)" % os.path.relpath(f, d))
        for k, cs in agg.items():
            if "render_" not in k:
                continue
            out.append("  " + k[:90])
            for cn, vals in sorted(cs.items()):
                out.append("    %-36s n=%d mean=%.6g" % (cn, len(vals), sum(vals) / len(vals)))
for name in ("bench_plain.json", "bench_trace.json"):
    f = os.path.join(d, name)
    if os.path.exists(f) and os.path.getsize(f):
        j = json.loads(open(f).read().strip().splitlines()[-1])
        out.append("== %s: value=%.1f %s ms_per_step=%.4f kernel_ms_per_frame=%.4f launch_ms=%.3f frames_per_launch=%.0f" % (
            name, j["value"], j["unit"], j["ms_per_step"], j["kernel_ms_per_frame"], j["roofline"]["launch_ms"], j["roofline"]["frames_per_launch"]))
print("\n".join(out))
