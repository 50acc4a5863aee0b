"""Experiment: Mrays/s, V/ray and ms/frame of an arbitrary scene (run on the GPU box):
   python tools/exp_scene_rate.py <scene.rts> [texture_dir] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dogeray_amd as dr
path = sys.argv[1]; tex = sys.argv[2] if len(sys.argv) > 2 else ""; frames = int(sys.argv[3]) if len(sys.argv) > 3 else 32
sc = dr.Scene.load(path, tex); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
ctx.set_traversal(int(os.environ.get('DOGERAY_TRAV', '2')))
st = dr.pack_settings13(s, 1, spp=1)
W, H = s.width, s.height
ctx.accum_reset(W, H)
ctx.render_accumulate(st, W, H, s.background, 1, 1000003, frames)      # warm-up: establishes the tile order
ctx.stats_reset()
t0 = time.perf_counter(); ctx.render_accumulate(st, W, H, s.background, 1 + 1000003 * frames, 1000003, frames); dt = time.perf_counter() - t0
timed = ctx.stats()
ctx.enable_counters(True); ctx.stats_reset()
ctx.render_accumulate(st, W, H, s.background, 1 + 1000003 * frames, 1000003, frames)
c = ctx.stats()
ctx.enable_counters(False); ctx.set_option("batch_frames", 1); ctx.stats_reset()
ctx.render_accumulate(st, W, H, s.background, 1 + 1000003 * frames, 1000003, min(frames, 8))
one = ctx.stats()
clk = 1e8 * one["diag"][0] / max(1, one["diag"][7])
print("   single frame per launch: %.3f ms/frame; mean wave lifetime %.3f ms of it (%d waves at %.0f MHz)" % (one["kernel_ms"] / max(1, one["frames"]), one["diag"][0] / max(1, one["frames"]) / 5120 / clk * 1e3, 5120, clk / 1e6))
print("%s: %d objects %dx%d  %.3f ms/frame  %.1f Mrays/s  rays/frame %.3g  V/ray %.1f L/ray %.2f S/ray %.2f  node-loop lane use %.2f" % (
    os.path.basename(path), sc.num_objects, W, H, timed["kernel_ms"] / frames, c["rays"] / dt / 1e6, c["rays"] / frames,
    c["node_visits"] / c["rays"], c["prim_tests"] / c["rays"], c["shades"] / c["rays"], c["node_visits"] / max(1, c["trav_slots"])))
