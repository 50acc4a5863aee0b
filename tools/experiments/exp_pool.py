"""Experiment: the pool kernel on the bench scene, per option set, with its own diagnostics (run on the GPU box):
   python tools/exp_pool.py "" "pool_min_fill=32" "pool=0" ...    ('' = defaults; FRAMES=32 frames per launch, REPS=3)
Prints kernel ms per frame (HIP events) and, from the pool_diag build of the same launch shape: batches and mean batch fill per stage,
shader cycles per batch inside the step, and the share of a wave's life spent choosing / stepping / pushing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dogeray_amd as dr
W, H = 1920, 1080
if os.environ.get("SCENE"):
    path = os.environ["SCENE"]
    sc = dr.Scene.load(path, os.environ.get("TEX", "")); sc.build_bvh(); s = sc.settings()
    W, H = s.width, s.height
else:
    path = bench.ensure_scene(os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"), 709, W, H)
    sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
st = dr.pack_settings13(s, 1, spp=1)
n = int(os.environ.get("FRAMES", "32"))
reps = int(os.environ.get("REPS", "3"))
for opts in (sys.argv[1:] or [""]):
    sets = [kv.split("=") for kv in opts.split(",") if kv]
    old = {k: ctx.get_option(k) for k, _ in sets}
    for k, v in sets: ctx.set_option(k, int(v))
    ctx.set_option("batch_frames", min(n, 256))
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 1, 1000003, n)      # warm-up, tile order
    best = None
    for rep in range(reps):
        ctx.stats_reset()
        ctx.render_accumulate(st, W, H, s.background, 1, 1000003, n)
        o = ctx.stats()
        ms = o["kernel_ms"] / max(1, o["frames"])
        if best is None or ms < best: best = ms
    line = "%-44s %.4f ms/frame" % (opts or "(defaults)", best)
    if ctx.get_option("pool"):
        ctx.set_option("pool_diag", 1)
        ctx.stats_reset()
        ctx.render_accumulate(st, W, H, s.background, 1, 1000003, n)
        o = ctx.stats()
        d = ctx.kernel_diag(16)
        ctx.set_option("pool_diag", 0)
        life = o["diag"][0]
        b, l = d[0:3], d[3:6]
        if sum(b):
            line += "  (diag build %.4f)\n" % (o["kernel_ms"] / max(1, o["frames"]))
            for k, name in enumerate(("node", "leaf", "shade")):
                line += "      %-5s batches/frame %9.0f  fill %5.1f  cycles/batch in step %7.0f  share of wave life %4.1f%%\n" % (
                    name, b[k] / n, l[k] / max(1, b[k]), d[7 + k] / max(1, b[k]), 100.0 * d[7 + k] / max(1, life))
            line += "      choosing + compacting %4.1f%% of wave life (%.0f cycles/batch);  node step: state in registers after %.0f cycles, record after %.0f more, compute + write back %.0f" % (
                100.0 * d[6] / max(1, life), d[6] / max(1, sum(b)), d[10] / max(1, b[0]), d[11] / max(1, b[0]), (d[7] - d[10] - d[11]) / max(1, b[0]))
    print(line, flush=True)
    for k, v in old.items(): ctx.set_option(k, v)
