"""Reads a rocprofv3 --kernel-trace CSV of a dr_group run (one-device rehearsal or real) and says whether the gather ran BESIDE the rendering:
for every stripe_copy_kernel (pack on a rank's render stream, unpack on rank 0's communication stream) the render kernels that were running when it
started and ended.   python tools/trace_overlap.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
if not rows:
    sys.exit("no kernel trace found")
t0 = rows[0][0]
render = [(a, b, q) for a, b, n, q in rows if "render_persistent_kernel" in n and "<false" in n]
copies = [(a, b, q) for a, b, n, q in rows if "stripe_copy_kernel" in n]
print("%d render launches, %d stripe copies (pack + unpack); time zero = first kernel" % (len(render), len(copies)))
# only the timed part is interesting: take the last third of the trace
cut = t0 + (rows[-1][1] - t0) * 2 // 3
beside = 0; waited = 0; shown = 0
for a, b, q in copies:
    if a < cut: continue
    running_at_start = [(ra, rb, rq) for ra, rb, rq in render if ra <= a < rb]
    started_after = [(ra, rb, rq) for ra, rb, rq in render if a <= ra < b]
    if running_at_start or started_after: beside += 1
    else: waited += 1
    if shown < 24:
        shown += 1
        print("  copy on queue %s: %9.1f .. %9.1f us (%.1f us)   render kernels running when it started: %d (ending %s)   started while it ran: %d" % (
            q, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, len(running_at_start),
            ", ".join("%.1f" % ((rb - t0) / 1e3) for _, rb, _ in running_at_start[:4]) or "-", len(started_after)))
print("stripe copies in the last third of the trace: %d ran beside a render kernel, %d did not" % (beside, waited))
