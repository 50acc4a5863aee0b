"""Experiment: how well does one frame's per-tile cost predict where the NEXT frame's longest pixels are?
   python tools/exp_cost_predict.py   (GPU box; bench scene, one frame per launch)"""
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
ctx.set_option("batch_frames", 1)
maps = []
for k in range(6):
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 1 + 1000003 * k, 1000003, 1)
    maps.append(ctx.pixel_cost(W, H).astype(np.int64))
def tiles(m, f):
    t = m[:W // 8 * 8, :H // 8 * 8].reshape(W // 8, 8, H // 8, 8).transpose(0, 2, 1, 3).reshape(W // 8, H // 8, 64)
    return f(t)
nt = (W // 8) * (H // 8)
for thr in (200, 300, 400):
    tgt = maps[5] > thr
    tt = tiles(tgt.astype(np.int64), lambda t: t.sum(axis=2)).ravel()
    print("frame 5: %d pixels over %d steps in %d of %d tiles" % (tgt.sum(), thr, (tt > 0).sum(), nt))
    for name, score in (("sum of frame 4", tiles(maps[4], lambda t: t.sum(axis=2))), ("max of frame 4", tiles(maps[4], lambda t: t.max(axis=2))),
                        ("max over frames 0-4", np.max([tiles(m, lambda t: t.max(axis=2)) for m in maps[:5]], axis=0)),
                        ("sum over frames 0-4", np.sum([tiles(m, lambda t: t.sum(axis=2)) for m in maps[:5]], axis=0)),
                        ("pixels over 150, frames 0-4", np.sum([tiles((m > 150).astype(np.int64), lambda t: t.sum(axis=2)) for m in maps[:5]], axis=0))):
        order = np.argsort(-score.ravel(), kind="stable")
        got = np.cumsum(tt[order]) / max(1, tt.sum())
        print("   tiles ranked by %-28s: first 5%% of tiles hold %.2f of them, 10%% %.2f, 20%% %.2f, 33%% %.2f, 50%% %.2f" % (
            name, got[nt // 20], got[nt // 10], got[nt // 5], got[nt // 3], got[nt // 2]))
