"""Experiment: where one launch of the persistent kernel spends its time (run on the GPU box):
   python tools/exp_timeline.py [frames_per_launch=1] ["opt=v,opt=v"]
Uses the wave log (begin / queue empty / end of every wave) and the per-pixel step counts of the bench scene."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dogeray_amd as dr
W, H = 1920, 1080
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opts = sys.argv[2] if len(sys.argv) > 2 else ""
if os.environ.get("SCENE"):          # another scene at its own size
    sc = dr.Scene.load(os.environ["SCENE"], os.environ.get("TEX", "")); sc.build_bvh(); s = sc.settings()
    W, H = s.width, s.height
else:
    path = bench.ensure_scene(os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"), 709, W, H)
    sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
for kv in opts.split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
st = dr.pack_settings13(s, 1, spp=1)
ctx.set_option("batch_frames", batch)
ctx.set_option("wave_log", 1)
ctx.accum_reset(W, H)
ctx.render_accumulate(st, W, H, s.background, 1, 1000003, 8 * batch)
ctx.stats_reset()
ctx.render_accumulate(st, W, H, s.background, 1, 1000003, batch)      # ONE launch
o = ctx.stats()
log = ctx.wave_log().astype(np.int64)
t0 = log[:, 0].min()
b, e, end, it = (log[:, 0] - t0) / 100.0, np.where(log[:, 1] > 0, log[:, 1] - t0, 0) / 100.0, (log[:, 2] - t0) / 100.0, log[:, 3]      # microseconds
print("launch: %.1f us by HIP events, %d waves; last wave ends at %.1f us" % (o["kernel_ms"] * 1e3, len(log), end.max()))
q = lambda a, p: np.percentile(a, p)
print("wave begin   : median %.1f  p99 %.1f  max %.1f us" % (q(b, 50), q(b, 99), b.max()))
print("queue empty  : min %.1f  median %.1f  max %.1f us   (first wave to find the queue empty .. last)" % (e.min(), q(e, 50), e.max()))
print("wave end     : p10 %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f us" % (q(end, 10), q(end, 50), q(end, 90), q(end, 99), end.max()))
tail = end - e
print("drain (end - queue empty) per wave: median %.1f  p90 %.1f  p99 %.1f  max %.1f us;  iterations in it: median %d  max %d;  us/iteration: %.3f" % (
    q(tail, 50), q(tail, 90), q(tail, 99), tail.max(), q(it, 50), it.max(), tail.sum() / max(1, it.sum())))
edges = np.arange(0, end.max() + 50, 50)
alive = [(int(((b <= t) & (end > t)).sum()), int(((e <= t) & (end > t)).sum())) for t in edges]
print("waves alive / of them draining, every 50 us: " + "  ".join("%d:%d/%d" % (t, a, d) for t, (a, d) in zip(edges, alive)))
late = np.argsort(-end)[:12]
print("the waves that end last:  end us | drain us | iterations")
for w in late:
    print("   wave %5d: %7.1f | %6.1f | %4d" % (w, end[w], tail[w], it[w]))
pc = ctx.pixel_cost(W, H)
if pc.size:
    print("node steps per pixel: mean %.1f  p50 %d  p99 %d  p99.9 %d  p99.99 %d  max %d;  pixels over 300 steps: %d, over 600: %d" % (
        pc.mean(), q(pc, 50), q(pc, 99), q(pc, 99.9), q(pc, 99.99), pc.max(), (pc > 300).sum(), (pc > 600).sum()))
