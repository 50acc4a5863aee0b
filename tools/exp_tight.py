"""Experiment (one configuration per process: the tree is built at upload): the bench scene with the environment's DOGERAY_X_TIGHT / DOGERAY_X_MU,
20 frames per launch; prints kernel ms per frame, records and primitive tests per ray (counting build) and how many pixels of the accumulated image
differ from /tmp/exp_tight_base.npy (written by the run without the variables).   python tools/exp_tight.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dogeray_amd as dr
W, H = 1920, 1080
path = bench.ensure_scene(os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"), 709, W, H)
sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
st = dr.pack_settings13(s, 1, spp=1)
n = 20
ctx.set_option("batch_frames", n)
ctx.accum_reset(W, H)
ctx.render_accumulate(st, W, H, s.background, 1, 1000003, n)
img = ctx.accum_read().copy()
best = None
for rep in range(5):
    ctx.stats_reset()
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 1, 1000003, n)
    o = ctx.stats()
    ms = o["kernel_ms"] / max(1, o["frames"])
    best = ms if best is None or ms < best else best
ctx.enable_counters(True)
ctx.stats_reset(); ctx.accum_reset(W, H)
ctx.render_accumulate(st, W, H, s.background, 1, 1000003, 2)
o = ctx.stats()
tag = "tight=%s mu=%s" % (os.environ.get("DOGERAY_X_TIGHT", "-"), os.environ.get("DOGERAY_X_MU", "-"))
base = "/tmp/exp_tight_base.npy"
if "DOGERAY_X_TIGHT" not in os.environ and "DOGERAY_X_MU" not in os.environ:
    np.save(base, img); diff = 0
else:
    diff = int((np.load(base) != img).any(axis=-1).sum()) if os.path.exists(base) else -1
print("%-28s %.4f ms/frame   records/ray %.2f   primitive tests/ray %.2f   pixels differing from the baseline %d of %d" % (tag, best, o["node_visits"] / max(1, o["rays"]), o["prim_tests"] / max(1, o["rays"]), diff, W * H))
