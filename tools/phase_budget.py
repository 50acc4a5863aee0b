"""The shade / refill phase's dynamic budget on the bench scene (GPU box): the counting build's phase counters (dr_stats_phase_counts) per frame --
phases, lanes shaded / started, turns of the rejection loop (histogram), candidates per turn -- to go with the static prices of tools/isa_cost.py.
   python tools/phase_budget.py [frames=32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dogeray_amd as dr
W, H = 1920, 1080
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
path = bench.ensure_scene(os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"), 709, W, H)
sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
st = dr.pack_settings13(s, 1, spp=1)
ctx.set_option("batch_frames", frames)
ctx.accum_reset(W, H)
ctx.render_accumulate(st, W, H, s.background, 1, 1000003, frames)
ctx.enable_counters(True); ctx.stats_reset()
ctx.render_accumulate(st, W, H, s.background, 1, 1000003, frames)
o = ctx.stats(); pc = ctx.phase_counts(32)
ctx.enable_counters(False)
d = o["diag"]
f = float(frames)
phases = d[3] / f
print("per frame (counting build, %d frames per launch): rays %.0f, loop iterations %.0f, phases %.0f, wave-level node steps %.0f, leaf steps %.0f" % (frames, o["rays"] / f, d[2] / f, phases, d[4] / f, d[5] / f))
print("   lanes shaded per phase %.2f;  lanes that drew a point: in the sphere (scatter) %.2f, in the disk (new path) %.2f per phase;  retired lanes %.2f" % (
    d[6] / max(1, d[3]), pc[18] / max(1, d[3]), pc[19] / max(1, d[3]), pc[20] / max(1, d[3])))
turns, cands = pc[17], pc[16]
print("   rejection loop: %.2f turns per phase (the unluckiest lane's), %.2f candidates per phase = %.2f lanes drawing per turn; per drawing lane %.2f candidates" % (
    turns / max(1, d[3]), cands / max(1, d[3]), cands / max(1, turns), cands / max(1, pc[18] + pc[19])))
tot = sum(pc[:16])
print("   turns per phase, share of phases:  " + "  ".join("%d:%.3f" % (k, pc[k] / max(1, tot)) for k in range(16)))
print("   cycles inside phases / wave cycles (counting build): %.3f;  shader cycles per phase %.0f" % (d[1] / max(1, d[0]), d[1] / max(1, d[3])))
