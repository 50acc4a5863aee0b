"""Experiment (GPU box): the walk as a kernel of its own.  The rays of real frames of the bench scene (logged by the counting build) through the trace-only probe
(dr_context_probe_trace), per variant, against the megakernel's rate on the same frames:   python tools/exp_trace_rate.py [frames=4] [variants=0,1,2,3,4,5,6,7]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dogeray_amd as dr
W, H = 1920, 1080
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3,4,5,6,7").split(",")]
cfg = os.environ.get("CONFIG", "C4")
if cfg == "C4":
    path = bench.ensure_scene(os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"), 709, W, H); tex = ""
else:
    class A: pass
    a = A(); a.config = cfg; a.cache = os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"); a.verts = 709; a.width = W; a.height = H
    path, tex, W, H, _ = bench.select_workload(a)
sc = dr.Scene.load(path, tex); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
st = dr.pack_settings13(s, 1, spp=1)
ctx.set_option("batch_frames", 32)
ctx.accum_reset(W, H)
ctx.render_accumulate(st, W, H, s.background, 1, 1000003, 32)
ctx.stats_reset(); ctx.render_accumulate(st, W, H, s.background, 1, 1000003, 32)
o = ctx.stats()
ms_frame = o["kernel_ms"] / 32
ctx.enable_counters(True); ctx.stats_reset(); ctx.render_accumulate(st, W, H, s.background, 1, 1000003, 2); rays_frame = ctx.stats()["rays"] / 2; ctx.enable_counters(False)
print("%s: megakernel %.4f ms/frame, %.0f rays/frame = %.2f Grays/s (walk + shade + refill in one kernel)" % (cfg, ms_frame, rays_frame, rays_frame / ms_frame / 1e6))
names = {0: "one ray per lane, waves wait for their slowest", 1: "persistent, 6 waves/SIMD, refill at 1 free lane, leaf step at 20", 2: "persistent, 6, refill at 8, leaf 20", 3: "persistent, 6, refill at 16, leaf 20",
         4: "persistent, 8 waves/SIMD, refill at 8, leaf 20", 5: "persistent, 8, refill at 8, leaf 28", 6: "persistent, 6, refill at 8, leaf 28", 7: "persistent, 8, refill at 4, leaf 32"}
for v in variants:
    r, n, bad = ctx.probe_trace(st, W, H, s.background, 1, frames, v)
    print("  variant %d (%s): %d rays per frame, %.2f Grays/s = %.4f ms per frame's rays (%.0f %% of the megakernel's frame), %d results differ" % (v, names.get(v, "?"), n, r / 1e9, n / r * 1e3, 100 * (n / r * 1e3) / ms_frame, bad), flush=True)
