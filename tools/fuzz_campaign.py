"""GPU box: a longer parity campaign than the test suite runs -- random scenes (tests/scene_fuzz.py), random frame sizes (also not multiples
of 8), random scheduling options of the work-sharing / split-tile / six-wave paths, single frames (twice: the second launch has the first
one's tile order) and batched accumulation, every frame against the oracle.   python tools/fuzz_campaign.py [scenes=40] [seed=1]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dogeray_amd as dr
from oracle import orc
from scene_fuzz import random_scene

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
d = tempfile.mkdtemp(prefix="dogeray_fuzz_")
tex = os.path.join(d, "tex"); os.makedirs(tex)
gen = os.path.join(ROOT, "tools", "scenegen")
for name, w, h, k in (("synth_albedo.ppm", 128, 128, 0), ("synth_rough.ppm", 64, 64, 1), ("synth_env.ppm", 256, 128, 2), ("a.ppm", 32, 32, 0)):
    subprocess.check_call([gen, "ppm", os.path.join(tex, name), str(w), str(h), str(k)])
names = ["synth_albedo.ppm", "synth_rough.ppm", "synth_env.ppm", "a.ppm"]
ctx = dr.Context(0)
defaults = {k: ctx.get_option(k) for k in ("coop_steps", "coop_rounds", "split_parts", "split_steps", "split_waves", "occupancy", "batch_frames", "feedback_every", "coop_tiles_per_wave", "park_min", "unroll")}
pool = {"coop_steps": [1, 2, 8], "coop_rounds": [1, 2, 5], "split_parts": [1, 2, 4, 8], "split_steps": [16, 32, 400], "split_waves": [5, 12, 100, 1000], "occupancy": [4, 5, 6],
        "batch_frames": [1, 2, 3, 32], "feedback_every": [1, 8], "coop_tiles_per_wave": [0, 32, 100000], "park_min": [0, 8, 16], "unroll": [1, 2]}
bad = frames = 0
for k in range(n_scenes):
    sizes = [int(v) for v in os.environ["FUZZ_SIZES"].split(",")] if os.environ.get("FUZZ_SIZES") else None      # e.g. FUZZ_SIZES=8,9,16,24,33: tiny frames
    W = int(rng.choice(sizes or [64, 96, 100, 131, 200, 320])); H = int(rng.choice(sizes or [40, 64, 75, 128, 192]))
    nobj = int(rng.integers(2, 1500))
    path = random_scene(rng, nobj, os.path.join(d, "f%d.rts" % k), W=W, H=H, textures=names)
    ps = dr.Scene.load(path, tex); ps.build_bvh()
    osc = orc.Scene(path, tex); osc.build_bvh()
    ctx.upload(ps)
    s = ps.settings()
    opts = {name: int(rng.choice(v)) for name, v in pool.items() if rng.random() < 0.6}
    for name, v in defaults.items(): ctx.set_option(name, v)
    for name, v in opts.items(): ctx.set_option(name, v)
    st = dr.pack_settings13(s, 1)
    seed = 100 + k
    ref, _ = osc.render(st, W, H, s.background, seed, nthreads=8)
    ok = True
    for rep in range(3):                                   # the same frame three times: no order, order, order + refreshed costs
        g = ctx.render_frame(st, W, H, s.background, seed); frames += 1
        ok &= bool(np.array_equal(g, ref))
    n = int(rng.integers(2, 6))
    total = ref.astype(np.int64).copy()
    for f in range(1, n):
        total += osc.render(st, W, H, s.background, seed + 1000003 * f, nthreads=8)[0]
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, seed, 1000003, n); frames += n
    ok &= bool(np.array_equal(ctx.accum_read().astype(np.int64), total))
    if not ok:
        bad += 1
        print("MISMATCH scene %d (%d objects, %dx%d) options %r" % (k, nobj, W, H, opts), flush=True)
    elif k % 10 == 0:
        print("scene %d ok (%d objects, %dx%d, %r)" % (k, nobj, W, H, opts), flush=True)
print("fuzz campaign: %d scenes, %d frames, %d mismatching scenes" % (n_scenes, frames, bad))
sys.exit(1 if bad else 0)
