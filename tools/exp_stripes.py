"""Experiment: what one GPU of an N-GPU run has to do.  Renders the bench scene with the stripe of every rank of
   world = 1, 2, 4, 8 in turn on ONE GPU and reports the slowest rank's kernel time per K-frame launch, i.e. the compute
   part of bench.py --gpus N (the gather is not in it):
   python tools/exp_stripes.py <scene.rts> [frames=32] [batches...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dogeray_amd as dr
path = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
st = dr.pack_settings13(s, 1, spp=1)
W, H = s.width, s.height
import os
for kv in os.environ.get("EXP_OPTIONS", "").split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
worlds = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
base = None
for world in worlds:
    worst, total = 0.0, 0.0
    per = []
    for rank in range(world):
        ctx.set_stripe(world, rank)
        ctx.set_option("batch_frames", min(256, 32 * world))
        ctx.accum_reset(W, H)
        ctx.render_accumulate(st, W, H, s.background, 1, 1000003, frames)      # warm-up: establishes the tile order
        ctx.render_accumulate(st, W, H, s.background, 1, 1000003, frames)
        ctx.stats_reset()
        ctx.render_accumulate(st, W, H, s.background, 1 + 1000003 * frames, 1000003, frames)
        ms = ctx.stats()["kernel_ms"]
        per.append(ms); worst = max(worst, ms); total += ms
    if base is None:
        base = worst
    print("world %d: %d frames, slowest rank %.3f ms (%.3f ms/frame), mean rank %.3f ms, compute-only speed-up %.2fx  [%s]" % (
        world, frames, worst, worst / frames, total / world, base / worst, " ".join("%.2f" % p for p in per)), flush=True)
