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
print("the waves that end last:  end us | drain us | iterations | phases | hand-overs | walking lanes per iteration | us in phases")
for w in late:
    print("   wave %5d: %7.1f | %6.1f | %4d | %4d | %4d | %5.2f | %6.1f" % (w, end[w], tail[w], it[w], log[w, 4], log[w, 5], log[w, 6] / max(1, it[w]), log[w, 7] / 100.0))
if log[:, 14].sum():
    life = end - b
    print("whole life of the same waves: us | loop iterations | us per iteration")
    for w in late:
        print("   wave %5d: %7.1f | %5d | %.3f" % (w, life[w], log[w, 14], life[w] / max(1, log[w, 14])))
    print("   all waves: mean %.1f us, %.1f iterations, %.3f us per iteration" % (life.mean(), log[:, 14].mean(), life.sum() / max(1, log[:, 14].sum())))
if log[:, 11].sum():
    print("sharing block, the same waves: iterations it ran | givers | idle lanes | walking owners | owners waiting for helpers | lanes waiting for a phase   (per iteration)")
    for w in late:
        k = max(1, log[w, 11])
        print("   wave %5d: %5d | %5.2f | %5.2f | %5.2f | %5.2f | %5.2f" % (w, log[w, 11], log[w, 8] / k, log[w, 9] / k, log[w, 10] / k, log[w, 12] / k, log[w, 13] / k))
    k = max(1, log[:, 11].sum())
    print("   all waves: %d | %.2f | %.2f | %.2f | %.2f | %.2f" % (log[:, 11].sum(), log[:, 8].sum() / k, log[:, 9].sum() / k, log[:, 10].sum() / k, log[:, 12].sum() / k, log[:, 13].sum() / k))
if log[:, 4].sum():
    print("all waves, drain: iterations %d  phases %d  hand-overs %d  walking lanes per iteration %.2f  share of drain time in phases %.2f" % (
        it.sum(), log[:, 4].sum(), log[:, 5].sum(), log[:, 6].sum() / max(1, it.sum()), log[:, 7].sum() / 100.0 / tail.sum()))
pc = ctx.pixel_cost(W, H)
if pc.size:
    print("node steps per pixel: mean %.1f  p50 %d  p99 %d  p99.9 %d  p99.99 %d  max %d;  pixels over 300 steps: %d, over 600: %d" % (
        pc.mean(), q(pc, 50), q(pc, 99), q(pc, 99.9), q(pc, 99.99), pc.max(), (pc > 300).sum(), (pc > 600).sum()))
pt = ctx.pixel_times(W, H)
if pt is not None and pc.size:
    ps, pe = pt[0] / 100.0, pt[1] / 100.0
    life = pe - ps
    print("pixel lifetimes (experiment build): us per node step, by steps of the pixel:")
    for lo, hi in ((1, 10), (10, 50), (50, 150), (150, 300), (300, 450), (450, 10000)):
        m = (pc >= lo) & (pc < hi)
        if m.any(): print("   %4d..%-5d steps: %8d pixels  median lifetime %7.1f us  = %.2f us/step;  started median %.0f p90 %.0f max %.0f us;  finished median %.0f p99 %.0f max %.0f us" % (
            lo, hi, m.sum(), q(life[m], 50), np.median(life[m] / pc[m]), q(ps[m], 50), q(ps[m], 90), ps[m].max(), q(pe[m], 50), q(pe[m], 99), pe[m].max()))
    for T in (900, 1000, 1100):
        m = pe > T
        if m.any(): print("   pixels finished after %d us: %d;  their steps median %d (min %d max %d), started median %.0f (min %.0f max %.0f) us, us/step median %.2f" % (
            T, m.sum(), q(pc[m], 50), pc[m].min(), pc[m].max(), q(ps[m], 50), ps[m].min(), ps[m].max(), np.median(life[m] / np.maximum(1, pc[m]))))
    last = np.argsort(-pe.ravel())[:10]
    for i in last:
        x, y = np.unravel_index(i, pe.shape)
        print("   pixel (%4d,%4d): steps %4d  started %7.1f  finished %7.1f us  (%.2f us/step)" % (x, y, pc[x, y], ps[x, y], pe[x, y], life[x, y] / max(1, pc[x, y])))
    for spec in os.environ.get("PIXELS", "1103,623;1099,620;1099,623;1110,621;1037,623" if not os.environ.get("SCENE") else "0,0").split(";"):
        x, y = map(int, spec.split(","))
        print("   watched pixel (%4d,%4d): steps %4d  started %7.1f  finished %7.1f us  (%.2f us/step)" % (x, y, pc[x, y], ps[x, y], pe[x, y], life[x, y] / max(1, pc[x, y])))
