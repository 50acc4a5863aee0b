"""Experiment (experiment build, -DDR_WAVE_LOG_DETAIL=1): one frame per launch with one tile queue per XCD -- per XCD: when its waves left
their own band, how many tiles they took from it and from other bands, when they ended.   python tools/exp_regions.py ["opt=v,..."]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dogeray_amd as dr
W, H = 1920, 1080
path = bench.ensure_scene(os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"), 709, W, H)
sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
for kv in (sys.argv[1] if len(sys.argv) > 1 else "short_one_queue=0").split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
st = dr.pack_settings13(s, 1, spp=1)
ctx.set_option("batch_frames", 1); ctx.set_option("wave_log", 1)
ctx.accum_reset(W, H)
ctx.render_accumulate(st, W, H, s.background, 1, 1000003, 8)
ctx.stats_reset()
ctx.render_accumulate(st, W, H, s.background, 1, 1000003, 1)
o = ctx.stats()
log = ctx.wave_log().astype(np.int64)
t0 = log[:, 0].min()
end = (log[:, 2] - t0) / 100.0; left = np.where(log[:, 6] > 0, log[:, 6] - t0, 0) / 100.0; empty = np.where(log[:, 1] > 0, log[:, 1] - t0, 0) / 100.0
print("launch %.1f us; %d waves" % (o["kernel_ms"] * 1e3, len(log)))
for x in range(8):
    m = log[:, 15] == x
    if not m.any(): continue
    print("XCD %d: %4d waves; left own band median %.0f (min %.0f max %.0f) us; tiles per wave from own band %.1f, from others %.1f; queue empty median %.0f; wave end median %.0f p99 %.0f max %.0f us" % (
        x, m.sum(), np.median(left[m]), left[m].min(), left[m].max(), log[m, 4].mean(), log[m, 5].mean(), np.median(empty[m]), np.median(end[m]), np.percentile(end[m], 99), end[m].max()))
