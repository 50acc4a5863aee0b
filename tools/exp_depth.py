"""Experiment: ray rate by path depth -- how much of the frame time is the (coherent) primary rays and how much the scattered
   ones:  python tools/exp_depth.py <scene.rts> [frames=32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dogeray_amd as dr
path = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
W, H = s.width, s.height
for depth in (1, 2, 3, 10):
    st = dr.pack_settings13(s, 1, spp=1, depth=depth)
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 1, 1000003, frames)
    ctx.render_accumulate(st, W, H, s.background, 1, 1000003, frames)
    ctx.stats_reset()
    ctx.render_accumulate(st, W, H, s.background, 1 + 1000003 * frames, 1000003, frames)
    ms = ctx.stats()["kernel_ms"] / frames
    ctx.enable_counters(True); ctx.stats_reset()
    ctx.render_accumulate(st, W, H, s.background, 1 + 1000003 * frames, 1000003, frames)
    c = ctx.stats(); ctx.enable_counters(False)
    print("max depth %2d: %.3f ms/frame, %.2f M rays/frame, %.0f Mrays/s, V/ray %.1f, node-loop lane use %.2f" % (
        depth, ms, c["rays"] / frames / 1e6, c["rays"] / frames / ms / 1e3, c["node_visits"] / c["rays"], c["node_visits"] / max(1, c["trav_slots"])), flush=True)
