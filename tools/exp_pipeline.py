"""Experiment: one frame per launch, sequential against pipelined (dr_pipeline_*), per option set (run on the GPU box):
   python tools/exp_pipeline.py "" "feedback_every=1000" "short_one_queue=0" ...
Prints wall ms per frame for n one-frame launches in a row and for the same frames through the pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dogeray_amd as dr
W, H = 1920, 1080
path = bench.ensure_scene(os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"), 709, W, H)
sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
st = dr.pack_settings13(s, 1, spp=1)
n = int(os.environ.get("FRAMES", "48"))
for opts in (sys.argv[1:] or [""]):
    sets = [kv.split("=") for kv in opts.split(",") if kv]
    old = {k: ctx.get_option(k) for k, _ in sets}
    for k, v in sets: ctx.set_option(k, int(v))
    ctx.set_option("batch_frames", 1)
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 1, 1000003, 8)
    t0 = time.perf_counter(); ctx.render_accumulate(st, W, H, s.background, 1, 1000003, n); seq = (time.perf_counter() - t0) / n * 1e3
    ctx.render_accumulate_pipelined(st, W, H, s.background, 1, 1000003, 8)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); ctx.render_accumulate_pipelined(st, W, H, s.background, 1, 1000003, n); best = min(best, (time.perf_counter() - t0) / n * 1e3)
    print("%-50s sequential %.4f ms/frame   pipelined %.4f ms/frame" % (opts or "(defaults)", seq, best), flush=True)
    for k, v in old.items(): ctx.set_option(k, v)
