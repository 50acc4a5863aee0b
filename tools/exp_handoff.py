"""Experiment (GPU box): one frame per launch on the bench scene, the hand-off chain against the single work-sharing kernel, with the chain's timeline:
   python tools/exp_handoff.py "handoff=0" "handoff=1" "handoff=1,handoff_mid=1,handoff_mid_wait=24" ...
Per option set: kernel ms per frame (HIP events around the whole chain, best of 3 x FRAMES launches), and from the wave log of one more launch, per stage:
waves that took part, first begin / median end / last end in us since the launch began, paths handed on."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dogeray_amd as dr
W, H = 1920, 1080
if os.environ.get("SCENE"):
    sc = dr.Scene.load(os.environ["SCENE"], os.environ.get("TEX", "")); sc.build_bvh(); s = sc.settings(); W, H = s.width, s.height
else:
    path = bench.ensure_scene(os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"), 709, W, H)
    sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
st = dr.pack_settings13(s, 1, spp=1)
n = int(os.environ.get("FRAMES", "16"))
batch = int(os.environ.get("BATCH", "1"))
STAGE = 16384
for opts in (sys.argv[1:] or ["handoff=0", "handoff=1"]):
    sets = [kv.split("=") for kv in opts.split(",") if kv]
    old = {k: ctx.get_option(k) for k, _ in sets}
    for k, v in sets: ctx.set_option(k, int(v))
    ctx.set_option("batch_frames", batch)
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 1, 1000003, n * batch)
    best = 1e9
    for rep in range(3):
        ctx.stats_reset()
        ctx.render_accumulate(st, W, H, s.background, 1, 1000003, n * batch)
        o = ctx.stats()
        best = min(best, o["kernel_ms"] / max(1, o["launches"]))
    ctx.set_option("wave_log", 1)
    ctx.render_accumulate(st, W, H, s.background, 1, 1000003, batch)
    ctx.stats_reset()
    ctx.render_accumulate(st, W, H, s.background, 1, 1000003, batch)
    logged = ctx.stats()["kernel_ms"]
    log = ctx.wave_log().astype(np.int64)
    ctx.set_option("wave_log", 0)
    print("%-60s %.4f ms/launch of %d frame(s)   (the logged launch: %.4f)" % (opts, best, batch, logged))
    live = log[:, 0] > 0
    if live.any():
        t0 = log[live, 0].min()
        for sidx in range((len(log) + STAGE - 1) // STAGE):
            part = log[sidx * STAGE:(sidx + 1) * STAGE]
            m = part[:, 0] > 0
            if not m.any(): continue
            b, e, end = (part[m, 0] - t0) / 100.0, np.where(part[m, 1] > 0, part[m, 1] - t0, 0) / 100.0, (part[m, 2] - t0) / 100.0
            q = lambda a, p: float(np.percentile(a, p))
            alive = "  ".join("%d:%d" % (t, int(((b <= t) & (end > t)).sum())) for t in np.arange(np.floor(b.min() / 50) * 50, end.max() + 50, 50))
            print("   stage %d: %5d waves  begin %.1f..%.1f  queue empty (first..median..last) %.1f..%.1f..%.1f  end p10 %.1f median %.1f p90 %.1f p99 %.1f last %.1f us" % (
                sidx, int(m.sum()), b.min(), b.max(), e[e > 0].min() if (e > 0).any() else 0, q(e[e > 0], 50) if (e > 0).any() else 0, e.max(), q(end, 10), q(end, 50), q(end, 90), q(end, 99), end.max()))
            print("            waves alive every 50 us: " + alive)
    for k, v in old.items(): ctx.set_option(k, v)
