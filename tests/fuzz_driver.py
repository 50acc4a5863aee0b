"""Parity campaign shared by tools/fuzz_campaign.py (thousands of scenes, run by hand on the GPU box) and the -m gpu suite (a 200-scene
slice, so the driver fuzzes the work-sharing / split-tile / six-wave / pipelined paths too): random scenes (scene_fuzz.py), random frame
sizes (also not multiples of 8), random scheduling options, single frames (three times: no tile order, order, order + refreshed
costs), batched accumulation and pipelined single frames, every result against the oracle."""
import os

import numpy as np

OPTION_POOL = {"coop_steps": [1, 2, 8], "coop_rounds": [1, 2, 5], "split_parts": [1, 2, 4, 8], "split_steps": [16, 32, 400], "split_waves": [5, 12, 100, 1000],
               "occupancy": [4, 5, 6], "batch_frames": [1, 2, 3, 32], "feedback_every": [1, 8], "coop_tiles_per_wave": [0, 32, 100000], "schedule": [0, 0, 1, 2],
               "pipe_streams": [2, 3, 4], "pipe_lean": [0, 1], "pipe_group": [1, 2, 3, 8, 16]}


def run_campaign(dr, orc, ctx, n_scenes, seed, workdir, texdir, texture_names, sizes=None, log=print):
    """Returns (scenes, frames, list of mismatch descriptions)."""
    from scene_fuzz import random_scene
    rng = np.random.default_rng(seed)
    defaults = {k: ctx.get_option(k) for k in OPTION_POOL}
    bad, frames = [], 0
    try:
        for k in range(n_scenes):
            W = int(rng.choice(sizes or [64, 96, 100, 131, 200, 320])); H = int(rng.choice(sizes or [40, 64, 75, 128, 192]))
            nobj = int(rng.integers(2, 1500))
            scale = float(rng.choice([1.0, 1.0, 0.2, 0.05, 0.02]))      # (small scenes: triangles the size of the reference's padding, entered with their own bounds)
            path = random_scene(rng, nobj, os.path.join(workdir, "f%d.rts" % k), W=W, H=H, textures=texture_names, scale=scale)
            ps = dr.Scene.load(path, texdir); ps.build_bvh()
            osc = orc.Scene(path, texdir); osc.build_bvh()
            ctx.set_option("wide_tree", int(rng.choice([2, 2, 2, 1, 0])))      # (read at upload)
            ctx.upload(ps)
            s = ps.settings()
            opts = {name: int(rng.choice(v)) for name, v in OPTION_POOL.items() if rng.random() < 0.6}
            for name, v in defaults.items(): ctx.set_option(name, v)
            for name, v in opts.items(): ctx.set_option(name, v)
            st = dr.pack_settings13(s, 1)
            fseed = 100 + k
            ref, _ = osc.render(st, W, H, s.background, fseed, nthreads=8)
            ok = True
            what = []
            for rep in range(3):
                g = ctx.render_frame(st, W, H, s.background, fseed); frames += 1
                if not np.array_equal(g, ref): what.append("single frame, launch %d: %d pixels" % (rep, int(np.any(g != ref, axis=2).sum())))
            n = int(rng.integers(2, 6))
            total = ref.astype(np.int64).copy()
            for f in range(1, n):
                total += osc.render(st, W, H, s.background, fseed + 1000003 * f, nthreads=8)[0]
            ctx.accum_reset(W, H)
            ctx.render_accumulate(st, W, H, s.background, fseed, 1000003, n); frames += n
            got = ctx.accum_read().astype(np.int64)
            if not np.array_equal(got, total): what.append("batched accumulation of %d frames: %d pixels" % (n, int(np.any(got != total, axis=2).sum())))
            ctx.accum_reset(W, H)
            ctx.render_accumulate_pipelined(st, W, H, s.background, fseed, 1000003, n); frames += n
            got = ctx.accum_read().astype(np.int64)
            if not np.array_equal(got, total):
                where = np.argwhere(np.any(got != total, axis=2))
                what.append("pipelined accumulation of %d frames: %d pixels, first (x, y) %r, got %r want %r" % (n, len(where), where[:4].tolist(), got[tuple(where[0])].tolist(), total[tuple(where[0])].tolist()))
            ok = not what
            os.remove(path)
            if not ok:
                bad.append("scene %d (%d objects, %dx%d) options %r: %s" % (k, nobj, W, H, opts, "; ".join(what)))
                log("MISMATCH " + bad[-1])
            elif k % 25 == 0:
                log("scene %d ok (%d objects, %dx%d, %r)" % (k, nobj, W, H, opts))
    finally:
        for name, v in defaults.items(): ctx.set_option(name, v)
        ctx.set_option("wide_tree", 2)
    return n_scenes, frames, bad
