"""Experiment: one frame per launch on the bench scene, per option set (run on the GPU box):
   python tools/exp_single.py "coop_steps=4" "coop_lanes=16,coop_steps=4" ...   ('' = defaults)
Prints kernel ms per frame (HIP events) and the mean wave lifetime as a share of it (clock stamps of every wave)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dogeray_amd as dr
W, H = 1920, 1080
if os.environ.get("SCENE"):          # SCENE=/tmp/cfg/bunny.rts [TEX=/tmp/cfg/tex]: another scene at its own size (tools/r2_configs.sh generates them)
    path = os.environ["SCENE"]
    sc = dr.Scene.load(path, os.environ.get("TEX", "")); sc.build_bvh(); s = sc.settings()
    W, H = s.width, s.height
else:
    path = bench.ensure_scene(os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"), 709, W, H)
    sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
st = dr.pack_settings13(s, 1, spp=1)
n = int(os.environ.get("FRAMES", "16"))
for opts in (sys.argv[1:] or [""]):
    ctx2 = ctx
    sets = [kv.split("=") for kv in opts.split(",") if kv]
    old = {k: ctx.get_option(k) for k, _ in sets}
    for k, v in sets: ctx.set_option(k, int(v))
    if "batch_frames" not in old: ctx.set_option("batch_frames", 1)
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 1, 1000003, n)
    best = None
    move = float(os.environ.get("MOVE", "0"))       # MOVE=0.05: the camera moves by that much between frames (every frame is a new view)
    for rep in range(3):
        ctx.stats_reset()
        if move:
            for k in range(n):
                st2 = st.copy(); st2[0] += move * (k + 1 + n * rep)
                ctx.accum_reset(W, H)
                ctx.render_accumulate(st2, W, H, s.background, 1 + k, 1000003, 1)
        else:
            ctx.render_accumulate(st, W, H, s.background, 1, 1000003, n)
        o = ctx.stats()
        ms = o["kernel_ms"] / max(1, o["launches"])
        if best is None or ms < best[0]: best = (ms, o)
    ms, o = best
    d = o["diag"]
    clk = 1e8 * d[0] / max(1, d[7])
    life = d[0] / max(1, o["launches"]) / 5120 / clk * 1e3      # (5120 waves: right for frames of 5120+ tiles)
    print("%-40s %.4f ms/launch (%d launches, %d frames)  mean wave lifetime %.4f ms = %.0f%%  clock %.0f MHz" % (opts or "(defaults)", ms, o["launches"], o["frames"], life, 100 * life / ms, clk / 1e6))
    for k, v in old.items(): ctx.set_option(k, v)
