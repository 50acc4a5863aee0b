"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: integer / index work bit-exact; the float pipeline is specified operation by operation
(-ffp-contract=off on both sides, IEEE divide/sqrt, no libm transcendental on the device), so
frames are expected to be IDENTICAL int32 triples.  north_star's tolerance is 1e-4 relative per
channel; the asserts below are stricter (exact) and print the measured agreement.
"""
import os

import numpy as np
import pytest

from conftest import CUBE_SETTINGS, SCENES, frame_stats, with_settings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dr():
    import dogeray_amd
    assert dogeray_amd.device_count() >= 1, "no GPU visible: the HIP path cannot run (there is no fallback)"
    return dogeray_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import orc as o
    return o


@pytest.fixture(scope="module")
def ctx(dr, synth):          # synth first: the generated scenes exist before this process touches the GPU
    c = dr.Context(0)
    yield c
    c.close()


def rand_dirs(rng, n):
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[rng.random(n) < 0.05, 0] = 0.0          # axis-parallel rays: 1/0 = inf in the slab test
    d[rng.random(n) < 0.02, 1] = -0.0
    return d


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_rng_xorwow(ctx, orc):
    for seed in (0, 1, 12345, 2 ** 32 + 7, 2 ** 64 - 1, 1 + 1000003 * 5 + 255 + 255 * 256):
        g = ctx.kat_rng(seed, 257)
        o = orc.kat_rng(seed, 257)
        assert np.array_equal(g.view(np.uint64), o.view(np.uint64)), "seed %d" % seed
        assert g.min() > 0.0 and g.max() < 1.0


def test_aabb(ctx, orc):
    rng = np.random.default_rng(1)
    n = 200_000
    o = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)
    d = rand_dirs(rng, n)
    c = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)
    e = rng.uniform(0, 3, size=(n, 3)).astype(np.float32)
    mn, mx = c - e, c + e
    o[:1000] = mn[:1000]                       # origin on a box face: (min - o) = 0, times inf = NaN
    gh, gd = ctx.kat_aabb(o, d, mn, mx)
    oh, od = orc.kat_aabb(o, d, mn, mx)
    assert np.array_equal(gh, oh)
    assert np.array_equal(bits(gd), bits(od))
    assert 0.05 < gh.mean() < 0.95


def test_triangle(ctx, orc):
    rng = np.random.default_rng(2)
    n = 200_000
    v0 = rng.uniform(-2, 2, size=(n, 3)).astype(np.float32)
    v1 = (v0 + rng.uniform(-1, 1, size=(n, 3))).astype(np.float32)
    v2 = (v0 + rng.uniform(-1, 1, size=(n, 3))).astype(np.float32)
    o = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    target = (v0 + v1 + v2) / 3 + rng.normal(scale=0.4, size=(n, 3))
    d = (target - o).astype(np.float32)
    d[:2000] = (v1[:2000] - v0[:2000])         # rays in the triangle's plane: |a| < EPSILON branch
    g = ctx.kat_tri(o, d, v0, v1, v2)
    r = orc.kat_tri(o, d, v0, v1, v2)
    assert np.array_equal(bits(g), bits(r))
    assert 0.1 < (g > 0).mean() < 0.9


def test_node_plane_arithmetic(ctx):
    """wide_node_test reads a plane byte as the f16 denormal byte * 2^-24 inside v_fma_mix_f32 (one instruction per plane): the result must be
    fma(byte, a, b) bit for bit -- i.e. the instruction keeps f16 denormals and rounds once -- over the whole range of a = scale / direction"""
    rng = np.random.default_rng(11)
    n = 200_000
    w = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    w[:4] = [0, 0xffffffff, 0x00ff00ff, 0xff00ff00]
    a = (rng.uniform(0.5, 1, size=n) * np.exp2(rng.integers(-120, 97, size=n)) * rng.choice([-1, 1], size=n)).astype(np.float32)
    b = (rng.normal(size=n) * np.exp2(rng.integers(-30, 40, size=n))).astype(np.float32)
    b[::7] = 0
    t_mix, t_cvt = ctx.kat_node_planes(w, a, b)
    assert np.array_equal(bits(t_mix), bits(t_cvt))
    assert np.isfinite(t_cvt).mean() > 0.99 and len(np.unique(t_cvt)) > n


def test_sphere(ctx, orc):
    rng = np.random.default_rng(3)
    n = 100_000
    c = rng.uniform(-2, 2, size=(n, 3)).astype(np.float32)
    r = rng.uniform(0.1, 1.5, size=n).astype(np.float32)
    o = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    d = (c + rng.normal(scale=0.7, size=(n, 3)) - o).astype(np.float32)
    g = ctx.kat_sphere(o, d, c, r)
    ref = orc.kat_sphere(o, d, c, r)
    assert np.array_equal(bits(g), bits(ref))
    assert 0.1 < (g > 0).mean() < 0.95


def test_optics(ctx, orc):
    rng = np.random.default_rng(4)
    n = 100_000
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
    nr = rng.normal(size=(n, 3)).astype(np.float32)
    nr /= np.linalg.norm(nr, axis=1, keepdims=True).astype(np.float32)
    eta = rng.choice(np.array([1.5, 1 / 1.5, 1.33, 2.4, 1.0], dtype=np.float32), size=n)
    g = ctx.kat_optics(v, nr, eta)
    r = orc.kat_optics(v, nr, eta)
    for a, b in zip(g, r):
        assert np.array_equal(bits(a), bits(b))


def _load_both(dr, orc, path, texdir=""):
    ps = dr.Scene.load(path, texdir)
    ps.build_bvh()
    os_ = orc.Scene(path, texdir if texdir else None)
    os_.build_bvh()
    return ps, os_


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_closest_hit_queries(dr, orc, ctx, synth, mode):
    """hit() K:468-512: t bit-exact and the same object index, for every traversal (2 = the wide walk, the default)."""
    rng = np.random.default_rng(5)
    for path in (os.path.join(synth["dir"], "hf_small.rts"), os.path.join(synth["dir"], "city_small.rts"),
                 os.path.join(SCENES, "scene.rts"), os.path.join(SCENES, "lots.rts")):
        ps, os_ = _load_both(dr, orc, path)
        ctx.upload(ps)
        ctx.set_traversal(mode)
        n = 100_000
        o = rng.uniform(-12, 12, size=(n, 3)).astype(np.float32)
        tgt = rng.uniform(-6, 6, size=(n, 3)).astype(np.float32)
        d = (tgt - o).astype(np.float32)
        gt, gi = ctx.kat_hit(o, d)
        rt, ri = os_.kat_hit(o, d)
        assert np.array_equal(bits(gt), bits(rt)), path
        assert np.array_equal(gi, ri), path
        assert (gt > 0).mean() > 0.02, path
        if mode == 2:
            assert ctx.get_option("traversal") == 2 and ctx.get_option("wide_depth") >= 1
    ctx.set_traversal(dr.TRAVERSAL_WIDE)


def _render_pair(dr, orc, ctx, path, texdir, W, H, div, seed, spp=None, depth=None, mode=0, kernel=1):
    ps, os_ = _load_both(dr, orc, path, texdir)
    ctx.upload(ps)
    ctx.set_traversal(mode)
    ctx.set_option("kernel", kernel)      # 0: one wave per tile, 1: persistent (ordered traversal always runs the tile kernel)
    s = ps.settings()
    so = os_.settings()
    assert bytes(s) == bytes(so)
    st = dr.pack_settings13(s, div, spp=spp, depth=depth)
    ctx.enable_counters(True)
    ctx.stats_reset()
    g = ctx.render_frame(st, W, H, s.background, seed)
    stats = ctx.stats()
    ctx.enable_counters(False)
    g2 = ctx.render_frame(st, W, H, s.background, seed)     # the timed (non-counting) build of the kernel
    if not np.array_equal(g, g2):
        bad = np.argwhere(np.any(g != g2, axis=2))
        raise AssertionError("counting and non-counting kernels disagree at %d pixels of %s, first (x, y): %r; counting %r, non-counting %r"
                             % (len(bad), os.path.basename(path), bad[:8].tolist(), g[tuple(bad[0])].tolist(), g2[tuple(bad[0])].tolist()))
    r, rc = os_.render(st, W, H, s.background, seed, nthreads=4)
    ctx.set_traversal(dr.TRAVERSAL_WIDE)       # the defaults
    ctx.set_option("kernel", 1)
    return g, r, stats, rc


def _assert_frames(g, r, what):
    frac, maxdiff = frame_stats(g, r)
    print("%s: %.6f of pixels identical, max |diff| = %d" % (what, frac, maxdiff))
    assert frac == 1.0, "%s: only %.6f of pixels identical (max diff %d)" % (what, frac, maxdiff)


@pytest.mark.parametrize("kernel,mode", [(0, 0), (1, 0), (1, 2), (0, 2)])
def test_cube_ladder_frames(dr, orc, ctx, tmp_path, kernel, mode):
    """Config C1: samples/cube.rts 256x256 1 spp, every preview-ladder stage (K:2169-2211)."""
    path = with_settings(os.path.join(SCENES, "cube.rts"), str(tmp_path / "cube256.rts"), CUBE_SETTINGS)
    for k, (div, spp, depth) in enumerate([(8, None, None), (4, 1, 2), (2, 1, 2), (1, 1, 2), (1, None, None), (1, None, None)]):
        seed = 1 + 1000003 * k
        g, r, stats, rc = _render_pair(dr, orc, ctx, path, "", 256, 256, div, seed, spp, depth, kernel=kernel, mode=mode)
        _assert_frames(g, r, "cube stage %d kernel %d traversal %d" % (k, kernel, mode))
        if div > 1:   # unrendered margin is 0
            assert not g[256 // div:, :, :].any() and not g[:, 256 // div:, :].any()
        for a, b in (("rays", "rays"), ("node_visits", "V"), ("prim_tests", "L"), ("shades", "S"), ("texels", "T"), ("samples", "samples")):
            if mode == 0 or a not in ("node_visits", "prim_tests"):      # only the threaded walk visits in the reference's order
                assert stats[a] == rc[b], (k, a, stats[a], rc[b])


@pytest.mark.parametrize("mode,kernel", [(2, 1), (2, 0), (0, 1), (0, 0), (1, 0)])
def test_scene_frames(dr, orc, ctx, synth, mode, kernel):
    """Spheres, every material, textures, checker, env map, smooth normals, large meshes."""
    cases = [
        (os.path.join(SCENES, "scene.rts"), "", 320, 192),              # spheres + the unsupported type 1
        (os.path.join(synth["dir"], "matball.rts"), synth["tex"], 256, 256),
        (os.path.join(SCENES, "glass.rts"), "", 320, 192),
        (os.path.join(SCENES, "cow.rts"), synth["tex"], 320, 192),      # a.ppm + testtwo.ppm by name
        (os.path.join(SCENES, "rough.blend.rts"), synth["tex"], 320, 192),   # env map + roughness texture
        (os.path.join(synth["dir"], "hf_small.rts"), "", 320, 192),
        (os.path.join(synth["dir"], "bunny_small.rts"), "", 320, 192),
        (os.path.join(synth["dir"], "city_small.rts"), "", 320, 192),
    ]
    for path, tex, W, H in cases:
        for seed in (3, 1 + 1000003 * 7):
            g, r, stats, rc = _render_pair(dr, orc, ctx, path, tex, W, H, 1, seed, mode=mode, kernel=kernel)
            _assert_frames(g, r, "%s seed %d traversal %d kernel %d" % (os.path.basename(path), seed, mode, kernel))
            assert stats["rays"] == rc["rays"] and stats["shades"] == rc["S"] and stats["texels"] == rc["T"]
            if mode == 0:
                assert stats["node_visits"] == rc["V"] and stats["prim_tests"] == rc["L"]
            else:
                print("   records visited: traversal %d: %d vs reference order %d; primitive tests %d vs %d" % (mode, stats["node_visits"], rc["V"], stats["prim_tests"], rc["L"]))


@pytest.mark.parametrize("n,ratio", [(64, 1.5), (200, 1.1)])
def test_deepest_wide_tree(dr, orc, ctx, tmp_path, n, ratio):
    """a wide tree of exactly WIDE_MAX_DEPTH = 17 levels -- the per-lane LDS stack full to its last word -- through the persistent kernel (lean and
    work-sharing build) and the tile kernel, against the oracle"""
    from scene_fuzz import stadium_scene
    path = stadium_scene(str(tmp_path / "stadium.rts"), n, ratio, W=192, H=128)
    for kernel in (1, 0):
        for seed in (5, 1 + 1000003 * 3):
            g, r, stats, rc = _render_pair(dr, orc, ctx, path, "", 192, 128, 1, seed, mode=2, kernel=kernel)
            assert ctx.get_option("traversal") == 2 and ctx.get_option("wide_depth") == 17
            _assert_frames(g, r, "stadium %d x %.1f kernel %d" % (n, ratio, kernel))
            assert stats["rays"] == rc["rays"] and stats["shades"] == rc["S"]


def _scaled_rts(src, dst, k):
    """A copy of a scene of triangles with every position (vertices, camera, look-at, focus distance) times k"""
    out = []
    for line in open(src).read().split("\n"):
        c = line.split(",")
        if line.startswith("*"):
            for i in (1, 2, 3, 5, 6, 7, 8): c[i] = repr(float(c[i]) * k)
        elif len(c) > 15 and c[3].strip() == "2":
            for i in (0, 1, 2, 9, 10, 11, 13, 14, 15): c[i] = repr(float(c[i]) * k)
        out.append(",".join(c))
    open(dst, "w").write("\n".join(out))
    return dst


@pytest.mark.parametrize("scene", ["hf_small.rts", "hf_small.rts/10", "bunny_small.rts", "city_small.rts"])
def test_wide_tree_over_the_triangles_own_bounds(dr, orc, ctx, synth, scene, tmp_path):
    """wide_tree = 2 (default): small triangles enter the tree with their own bounds instead of the reference's padded leaf box and every ray carries the
    margin of DESIGN.md 4.10 -- the frames stay the oracle's (camera rays and scattered rays, persistent and tile kernel, counting and plain build), with fewer
    records per ray than the tree over the padded boxes wherever triangles qualified"""
    path = os.path.join(synth["dir"], scene.split("/")[0])
    if "/" in scene: path = _scaled_rts(path, str(tmp_path / "scaled.rts"), 1.0 / float(scene.split("/")[1]))      # (a tenth of the size: its triangles qualify)
    visits = {}
    try:
        for tree in (2, 1):
            ctx.set_option("wide_tree", tree)
            for kernel in (1, 0):
                g, r, stats, rc = _render_pair(dr, orc, ctx, path, synth["tex"], 320, 192, 1, 31 + tree, mode=2, kernel=kernel)
                _assert_frames(g, r, "%s wide_tree %d kernel %d" % (scene, tree, kernel))
                assert stats["rays"] == rc["rays"] and stats["shades"] == rc["S"]
            visits[tree] = (stats["node_visits"] / stats["rays"], ctx.get_option("wide_own_bounds"))
        print("   %s: records per ray %.2f over own bounds (%d triangles), %.2f over the leaf boxes" % (scene, visits[2][0], visits[2][1], visits[1][0]))
        assert visits[1][1] == 0
        if visits[2][1] > 0: assert visits[2][0] < visits[1][0]
        if scene == "hf_small.rts/10": assert visits[2][1] > 9000
    finally:
        ctx.set_option("wide_tree", 2)


@pytest.mark.parametrize("kernel,mode", [(0, 0), (1, 0), (1, 2)])
def test_spp_and_aperture(dr, orc, ctx, synth, tmp_path, kernel, mode):
    """spp > 1 inside one launch (per-sample reseed, K:1059-1065) and a wide lens (K:1071-1073)."""
    src = os.path.join(synth["dir"], "matball.rts")
    path = with_settings(src, str(tmp_path / "mb4.rts"),
                         "*,0,-2.5,7,0.6,0,-0.5,0,7,50,6,4,0.9,synth_env.ppm,192,128")
    g, r, stats, rc = _render_pair(dr, orc, ctx, path, synth["tex"], 192, 128, 1, 77, kernel=kernel, mode=mode)
    _assert_frames(g, r, "matball 4 spp")
    assert stats["samples"] == 192 * 128 * 4


def test_progressive_accumulation(dr, orc, ctx, tmp_path):
    """Present loop K:2154-2224: ladder, then accumulate; display = clamp(sum / (iter - pnum))."""
    path = with_settings(os.path.join(SCENES, "cube.rts"), str(tmp_path / "cube256.rts"), CUBE_SETTINGS)
    ps, os_ = _load_both(dr, orc, path)
    ctx.upload(ps)
    s = ps.settings()
    pr = dr.ProgressiveRenderer(ctx, s, seed_base=1, seed_stride=1000003)
    outr = None
    for k in range(7):
        td, divide_by = pr.step()
        if k < 4:
            div = (8, 4, 2, 1)[k]
            st = dr.pack_settings13(s, div, spp=(None if k == 0 else 1), depth=(None if k == 0 else 2))
            outr, _ = os_.render(st, 256, 256, s.background, 1 + 1000003 * k, nthreads=4)
            assert td == div and divide_by == 1
        else:
            f, _ = os_.render(dr.pack_settings13(s, 1), 256, 256, s.background, 1 + 1000003 * k, nthreads=4)
            outr = outr + f
            assert td == 1 and divide_by == k - 2
        acc = ctx.accum_read()
        assert np.array_equal(acc, outr), "iteration %d" % k
        img = pr.image(divide_by)
        quot = (np.abs(outr) // divide_by) * np.sign(outr)          # C integer division truncates toward zero
        want = np.clip(quot, 0, 255).astype(np.uint8)
        assert np.array_equal(img, want.transpose(1, 0, 2))


def test_pipelined_present_loop(dr, orc, ctx, synth, tmp_path):
    """dr_pipeline_submit / dr_pipeline_wait (frame k + 1 starts while frame k drains; per-frame buffers added in ticket order): every
    displayed image is exactly clamp(sum of the oracle's frames so far / count) (K:2213-2218, K:2287), the accumulator ends as their sum,
    and calls of the ordinary API afterwards see all of it.  Long enough for the tile order to be refreshed inside the pipeline."""
    path = with_settings(os.path.join(SCENES, "cube.rts"), str(tmp_path / "cube256.rts"), CUBE_SETTINGS)
    ps, os_ = _load_both(dr, orc, path)
    ctx.upload(ps)
    s = ps.settings()
    pr = dr.ProgressiveRenderer(ctx, s, seed_base=1, seed_stride=1000003)
    outr = None
    for k in range(4):
        pr.step()
        st = dr.pack_settings13(s, (8, 4, 2, 1)[k], spp=(None if k == 0 else 1), depth=(None if k == 0 else 2))
        outr, _ = os_.render(st, 256, 256, s.background, 1 + 1000003 * k, nthreads=4)
    n = 21
    sums = []
    for k in range(4, 4 + n):
        f, _ = os_.render(dr.pack_settings13(s, 1), 256, 256, s.background, 1 + 1000003 * k, nthreads=4)
        outr = outr + f
        sums.append(outr.copy())
    shown = []
    pr.run_pipelined(n, on_image=lambda it, dv, img: shown.append((it, dv, img.copy())))
    assert [it for it, _, _ in shown] == list(range(5, 5 + n))
    for j, (it, dv, img) in enumerate(shown):
        assert dv == it - 3
        quot = (np.abs(sums[j]) // dv) * np.sign(sums[j])
        want = np.clip(quot, 0, 255).astype(np.uint8).transpose(1, 0, 2)
        assert np.array_equal(img, want), "image shown after iteration %d" % it
    assert np.array_equal(ctx.accum_read(), sums[-1])
    # the presented image in place (dr_pipeline_image): the same bytes as the copy
    t = ctx.pipeline_submit(dr.pack_settings13(s, 1), 256, 256, s.background, 99, present_divide_by=5)
    view = ctx.pipeline_wait(t, want_image=True, in_place=True)
    assert view.shape == (256, 256, 3) and not view.flags.writeable
    assert np.array_equal(view, ctx.accum_present(5))
    # and on a larger frame with textures, without presents, against the batched accumulation
    ps2 = dr.Scene.load(os.path.join(synth["dir"], "matball.rts"), synth["tex"])
    ps2.build_bvh()
    ctx.upload(ps2)
    s2 = ps2.settings()
    st2 = dr.pack_settings13(s2, 1)
    ctx.accum_reset(256, 256)
    ctx.render_accumulate(st2, 256, 256, s2.background, 9, 1000003, 13)
    want = ctx.accum_read()
    ctx.accum_reset(256, 256)
    ctx.render_accumulate_pipelined(st2, 256, 256, s2.background, 9, 1000003, 13)
    assert np.array_equal(ctx.accum_read(), want)
    ctx.render_accumulate_pipelined(st2, 256, 256, s2.background, 9, 1000003, 13)      # on top of it, then an ordinary call behind the pipeline
    ctx.render_accumulate(st2, 256, 256, s2.background, 9, 1000003, 13)
    assert np.array_equal(ctx.accum_read().astype(np.int64), 3 * want.astype(np.int64))
    # frames of a group that is still open when an option takes the per-frame-storing builds away are launched one by one; a frame with another seed step or
    # another view starts a new group; every group size gives the same sums
    for group in (1, 3, 16):
        ctx.set_option("pipe_group", group)
        ctx.accum_reset(256, 256)
        ts = [ctx.pipeline_submit(st2, 256, 256, s2.background, 9 + k * 1000003) for k in range(5)]
        ctx.set_option("schedule", 1)
        ts += [ctx.pipeline_submit(st2, 256, 256, s2.background, 9 + 5 * 1000003), ctx.pipeline_submit(st2, 256, 256, s2.background, 9 + 7 * 1000003 - 1000003)]
        ctx.pipeline_wait(ts[-1])
        ctx.set_option("schedule", 0)
        ts = [ctx.pipeline_submit(st2, 256, 256, s2.background, 9 + k * 1000003) for k in range(7, 13)]
        ctx.pipeline_wait(ts[-1])
        assert np.array_equal(ctx.accum_read(), want), group
    ctx.set_option("pipe_group", 8)
    # a camera that moves between submits: every change of view closes the group (K:2341-2500: an interactive viewer); the sum is the sum of the frames
    st2b = st2.copy(); st2b[0] += 0.25
    plan = [(st2, 9), (st2, 9 + 1000003), (st2b, 77), (st2b, 77 + 1000003), (st2b, 77 + 2 * 1000003), (st2, 5), (st2b, 6)]
    ctx.accum_reset(256, 256)
    for stv, seed in plan:
        ctx.render_accumulate(stv, 256, 256, s2.background, seed, 0, 1)
    want_views = ctx.accum_read()
    ctx.accum_reset(256, 256)
    ts = [ctx.pipeline_submit(stv, 256, 256, s2.background, seed) for stv, seed in plan]
    ctx.pipeline_wait(ts[-1])
    assert np.array_equal(ctx.accum_read(), want_views)


def test_moving_camera_keeps_the_previous_views_tile_order(dr, orc, ctx, synth):
    """An interactive viewer changes the camera between frames (K:2341-2500: every frame is a new view).  The persistent kernel then starts
    from the previous view's tile order (same tile grid: any order is a valid order) instead of none -- frames identical to the oracle's,
    also when the preview divisor changes the tile grid with W and H unchanged (that must NOT reuse the order)."""
    path = os.path.join(synth["dir"], "hf_small.rts")
    ps, os_ = _load_both(dr, orc, path, "")
    ctx.upload(ps)
    s = ps.settings()
    W, H = 320, 192
    assert ctx.get_option("order_follows_camera") == 1
    for k, (dx, div, depth) in enumerate([(0.0, 1, None), (0.0, 1, None), (0.4, 1, None), (0.8, 1, 3), (0.8, 2, 3), (1.2, 2, None), (1.2, 1, None), (-2.0, 1, None)]):
        st = dr.pack_settings13(s, div, spp=1, depth=depth)
        st[0] += dx
        st[4] -= 0.5 * dx
        g = ctx.render_frame(st, W, H, s.background, 77 + k)
        r, _ = os_.render(st, W, H, s.background, 77 + k, nthreads=4)
        _assert_frames(g, r, "moving camera, frame %d" % k)


def test_wave_log_and_pixel_cost_of_a_single_frame(dr, ctx, synth):
    """dr_stats_wave_log / dr_stats_pixel_cost (the measurement aids behind DESIGN 5's single-frame anatomy): every wave of a one-frame launch
    logs begin <= queue empty <= end, every rendered pixel has a cost of at least one node step, and neither changes the frame."""
    ps = dr.Scene.load(os.path.join(synth["dir"], "hf_small.rts"))
    ps.build_bvh()
    ctx.upload(ps)
    s = ps.settings()
    st = dr.pack_settings13(s, 1, spp=1)
    W, H = 320, 192
    plain = ctx.render_frame(st, W, H, s.background, 5)
    ctx.set_option("wave_log", 1)
    try:
        again = ctx.render_frame(st, W, H, s.background, 5)
        assert np.array_equal(plain, again)
        log = ctx.wave_log()
        assert len(log) > 0 and len(log) % 4 == 0
        begin, empty, end = log[:, 0].astype(np.int64), log[:, 1].astype(np.int64), log[:, 2].astype(np.int64)
        assert np.all(end >= begin) and np.all((empty == 0) | ((empty >= begin) & (empty <= end)))
        assert np.all(empty > 0)                                  # every wave sees the queue run empty before it ends
        cost = ctx.pixel_cost(W, H)
        assert cost.shape == (W, H) and cost.min() >= 1 and cost.max() < 100000
    finally:
        ctx.set_option("wave_log", 0)


def test_trace_only_probe_agrees_with_the_plain_walk(dr, ctx, synth):
    """dr_context_probe_trace (measurement aid): the rays of one frame, logged by the counting build in wavefront order, walked by the trace-only kernel --
    persistent waves refilling from the ray list -- give the same (t bits, slot) as the one-ray-per-lane walk, ray for ray; the log holds every ray of the frame."""
    ps = dr.Scene.load(os.path.join(synth["dir"], "hf_small.rts"))
    ps.build_bvh()
    ctx.upload(ps)
    s = ps.settings()
    st = dr.pack_settings13(s, 1, spp=1)
    W, H = 320, 192
    ctx.enable_counters(True); ctx.stats_reset()
    ctx.render_frame(st, W, H, s.background, 5)
    rays = ctx.stats()["rays"]
    ctx.enable_counters(False)
    for variant in (0, 1, 3, 7):
        rate, n, bad = ctx.probe_trace(st, W, H, s.background, 5, frames=2, variant=variant)
        assert n == rays and bad == 0 and rate > 0, (variant, n, rays, bad)


def test_stripes_partition_the_frame(dr, ctx, synth):
    """Multi-GPU partition: block columns bx % R == r; the union over r is the 1-GPU frame."""
    ps = dr.Scene.load(os.path.join(synth["dir"], "hf_small.rts"))
    ps.build_bvh()
    ctx.upload(ps)
    s = ps.settings()
    st = dr.pack_settings13(s, 1)
    full = ctx.render_frame(st, 320, 192, s.background, 9)
    total = np.zeros_like(full)
    for R in (2, 3, 8):
        total[:] = 0
        for r in range(R):
            ctx.set_stripe(R, r)
            part = ctx.render_frame(st, 320, 192, s.background, 9)
            cols = np.arange(320) // 8 % R == r
            assert part[~cols].max() == 0 and part[~cols].min() == 0
            total += part
        assert np.array_equal(total, full), R
    ctx.set_stripe(1, 0)


def test_batched_accumulation_equals_frame_sum(dr, ctx, synth):
    """dr_render_accumulate(nframes) may render several frames per launch (one work queue, atomic adds):
    the accumulator must equal the sum of the frames rendered one at a time."""
    ps = dr.Scene.load(os.path.join(synth["dir"], "hf_small.rts"))
    ps.build_bvh()
    ctx.upload(ps)
    s = ps.settings()
    st = dr.pack_settings13(s, 1)
    W, H, n = 320, 192, 11
    total = np.zeros((W, H, 3), dtype=np.int64)
    for k in range(n):
        total += ctx.render_frame(st, W, H, s.background, 5 + 1000003 * k)
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 5, 1000003, n)
    assert np.array_equal(ctx.accum_read().astype(np.int64), total)
    # and again on top (accumulator keeps its contents)
    ctx.render_accumulate(st, W, H, s.background, 5 + 1000003 * n, 1000003, 3)
    for k in range(n, n + 3):
        total += ctx.render_frame(st, W, H, s.background, 5 + 1000003 * k)
    assert np.array_equal(ctx.accum_read().astype(np.int64), total)


def test_tuning_options_do_not_change_pixels(dr, ctx, synth):
    """Every scheduling knob (kernel, occupancy, thresholds, batching, tile order) leaves the frame bit-identical."""
    ps = dr.Scene.load(os.path.join(synth["dir"], "city_small.rts"))
    ps.build_bvh()
    ctx.upload(ps)
    s = ps.settings()
    st = dr.pack_settings13(s, 1)
    W, H = 320, 192
    base = None
    combos = [{"kernel": 0, "occupancy": 4}, {"kernel": 0, "occupancy": 6}, {"kernel": 1, "occupancy": 4, "schedule": 1},
              {"kernel": 1, "occupancy": 5, "schedule": 2}, {"kernel": 1, "occupancy": 4, "schedule": 0},
              {"kernel": 1, "occupancy": 4, "schedule": 2, "feedback": 0}, {"kernel": 1, "batch_frames": 3},
              {"kernel": 1, "schedule": 1, "occupancy": 5}, {"kernel": 1, "schedule": 0, "occupancy": 5},
              {"kernel": 1, "batch_frames": 1, "coop_steps": 1, "coop_lanes": 64}, {"kernel": 1, "batch_frames": 1, "coop_steps": 0},
              {"kernel": 1, "batch_frames": 2, "coop_steps": 16, "coop_lanes": 4},
              {"kernel": 1, "batch_frames": 1, "split_parts": 4, "split_steps": 32, "coop_rounds": 4, "split_waves": 50}, {"kernel": 1, "batch_frames": 3, "split_parts": 8, "split_steps": 16},
              {"kernel": 1, "batch_frames": 1, "split_parts": 8, "split_steps": 16, "split_waves": 400},
              {"kernel": 1, "batch_frames": 1, "split_parts": 1, "coop_rounds": 1},
              {"kernel": 1, "occupancy": 6, "batch_frames": 32, "coop_tiles_per_wave": 0}, {"kernel": 1, "occupancy": 6, "batch_frames": 1},      # six waves per SIMD (lean wide build)
              ]
    for opts in combos:
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.accum_reset(W, H)
        ctx.render_accumulate(st, W, H, s.background, 21, 1000003, 7)
        ctx.render_accumulate(st, W, H, s.background, 21 + 7 * 1000003, 1000003, 2)   # second call reuses the tile order
        acc = ctx.accum_read()
        if base is None:
            base = acc
        assert np.array_equal(acc, base), opts
    for k, v in {"kernel": 1, "occupancy": 6, "schedule": 0, "feedback": 1, "batch_frames": 32, "coop_steps": 2, "coop_lanes": 8, "coop_rounds": 2,
                 "split_parts": 4, "split_steps": 400, "split_waves": 12, "coop_tiles_per_wave": 32}.items():
        ctx.set_option(k, v)
    assert ctx.get_option("schedule") == 0 and ctx.get_option("batch_frames") == 32 and ctx.get_option("tree_depth") == 11   # 1 730 leaves
    with pytest.raises(dr.DogerayError):
        ctx.set_option("schedule", 7)
    with pytest.raises(dr.DogerayError):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(dr.DogerayError):
        ctx.get_option("no_such_option")


def test_accumulator_tensor_aliases_device_memory(dr, ctx, synth):
    """multigpu.accumulator_tensor wraps the library's accumulator for torch.distributed without a copy."""
    import torch
    from dogeray_amd import multigpu
    ps = dr.Scene.load(os.path.join(synth["dir"], "city_small.rts"))
    ps.build_bvh()
    ctx.upload(ps)
    s = ps.settings()
    st = dr.pack_settings13(s, 1)
    W, H = 320, 192
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 3, 1, 2)
    t = multigpu.accumulator_tensor(ctx, torch.device("cuda", 0))
    assert t.dtype == torch.int32 and t.numel() == W * H * 3
    host = ctx.accum_read()
    assert np.array_equal(t.cpu().numpy().reshape(W, H, 3), host)
    ctx.render_accumulate(st, W, H, s.background, 5, 1, 1)                 # the tensor sees later frames: same memory
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy().reshape(W, H, 3), ctx.accum_read())
    assert multigpu.gather_frame(t, W, H, 1, 0) is t
    # pack / unpack used by the gather, on the device: columns r::R of the [gx, 8*H*3] view
    cols = t[: (W // 8) * 8 * H * 3].view(W // 8, 8 * H * 3)
    back = torch.zeros_like(cols)
    for r in range(3):
        back[r::3] = cols[r::3].contiguous()
    assert torch.equal(back, cols)


def test_scene_from_arrays_renders_like_the_file(dr, ctx, synth):
    src = dr.Scene.load(os.path.join(synth["dir"], "matball.rts"), synth["tex"])
    src.build_bvh()
    s = src.settings()
    st = dr.pack_settings13(s, 1)
    ctx.upload(src)
    want = ctx.render_frame(st, 256, 256, s.background, 4)
    twin = dr.Scene.from_arrays(src.objects(), s, bvh=src.bvh()[0], textures=src.textures())
    ctx.upload(twin)
    assert np.array_equal(ctx.render_frame(st, 256, 256, s.background, 4), want)


@pytest.mark.parametrize("kernel,mode", [(1, 2), (0, 2), (1, 0), (0, 0), (0, 1)])
def test_fuzzed_scenes_render_like_the_oracle(dr, orc, ctx, synth, tmp_path, kernel, mode):
    """Random scenes with spheres, all materials, textures, equal-t duplicates (tie-break = first leaf the
    reference's walk reaches), zero-area and axis-aligned triangles, lens blur, spp > 1: frames identical."""
    from scene_fuzz import random_scene
    rng = np.random.default_rng(77)
    names = ["synth_albedo.ppm", "synth_rough.ppm", "synth_env.ppm", "a.ppm"]
    for k in range(12):
        n = int(rng.integers(2, 600))
        path = random_scene(rng, n, str(tmp_path / ("fuzz%d.rts" % k)), W=96, H=64, textures=names, scale=(1.0, 0.05, 1.0, 0.02)[k % 4])      # (small scenes: the wide tree enters their triangles with their own bounds)
        g, r, stats, rc = _render_pair(dr, orc, ctx, path, synth["tex"], 96, 64, 1, 1000 + k, mode=mode, kernel=kernel)
        _assert_frames(g, r, "fuzz %d (%d objects) kernel %d traversal %d" % (k, n, kernel, mode))
        assert stats["rays"] == rc["rays"] and stats["shades"] == rc["S"] and stats["texels"] == rc["T"]


def test_equal_t_ties_go_to_the_first_leaf_in_reference_order(dr, orc, ctx, tmp_path):
    """Two coincident triangles with different colours: hit() keeps the one its walk reaches first (K:488, strict <)."""
    tri = "%s,2,%s,0.5,0,%s,1,%s"      # emissive (mat 1): the pixel shows which triangle won
    a = ("-1.000000,-1.000000,0.000000", "1.000000,0.000000,0.000000", "1.000000,-1.000000,0.000000", "0.000000,1.000000,0.000000")
    b = ("-1.000000,-1.000000,0.000000", "0.000000,1.000000,0.000000", "1.000000,-1.000000,0.000000", "0.000000,1.000000,0.000000")
    far = ("-1.000000,-1.000000,-1.000000", "0.000000,0.000000,1.000000", "1.000000,-1.000000,-1.000000", "0.000000,1.000000,-1.000000")
    winners = []
    for order in ((a, b, far), (b, a, far), (far, b, a)):
        p = tmp_path / "tie.rts"
        p.write_text("*,0,0,3,0.0,0,0,0,3,40,4,1,1,no,64,64\n" + "\n".join(tri % t for t in order) + "\n")
        for kernel, mode in ((1, 2), (0, 2), (1, 0), (0, 0), (0, 1)):
            g, r, _, _ = _render_pair(dr, orc, ctx, str(p), "", 64, 64, 1, 5, mode=mode, kernel=kernel)
            _assert_frames(g, r, "tie kernel %d traversal %d" % (kernel, mode))
        lit = g[g.sum(axis=2) > 0]
        hit_colours = {tuple(int(v > 0) for v in px) for px in lit if px.max() == 255 and sorted(px)[1] == 0}
        winners.append(hit_colours & {(1, 0, 0), (0, 1, 0)})
    # the coincident red / green pair: exactly one of them is ever seen, and swapping their order in the file swaps it
    assert all(len(w) == 1 for w in winners[:2]) and winners[0] != winners[1]


@pytest.mark.parametrize("kernel,mode", [(0, 0), (1, 0), (1, 2)])
def test_degenerate_settings(dr, orc, ctx, tmp_path, kernel, mode):
    """spp 0, depth 0 and odd frame sizes (margins, K:2633): zeros where the reference computes nothing."""
    src = os.path.join(SCENES, "cube.rts")
    for line, W, H in (("*,7.358891,-6.925791,4.958309,0.01,0,0,0,3,45,0,1,1,no,100,70", 100, 70),     # depth 0
                       ("*,7.358891,-6.925791,4.958309,0.01,0,0,0,3,45,5,0,1,no,100,70", 100, 70),     # spp 0
                       ("*,7.358891,-6.925791,4.958309,0.01,0,0,0,3,45,3,2,1,no,37,23", 37, 23),       # 4 x 2 tiles + margins
                       ("*,7.358891,-6.925791,4.958309,0.01,0,0,0,3,45,3,1,1,no,7,200", 7, 200)):      # narrower than one tile: nothing rendered
        path = with_settings(src, str(tmp_path / "d.rts"), line)
        g, r, stats, rc = _render_pair(dr, orc, ctx, path, "", W, H, 1, 9, kernel=kernel, mode=mode)
        _assert_frames(g, r, "degenerate %dx%d kernel %d" % (W, H, kernel))
        assert stats["rays"] == rc["rays"]
        assert not g[(W // 8) * 8:].any() and not g[:, (H // 8) * 8:].any()


def test_gpu_reproduces_the_reference_image(dr, ctx, tmp_path):
    """The reference's own saved frame for samples/eorovan.blend.rts (tests/reference_image.py) against 64 frames
    accumulated on the GPU through the C ABI: sky to a grey level, silhouette, untextured glossy floor."""
    import reference_image as ri
    sc = dr.Scene.load(ri.scene_path(tmp_path), "")
    sc.build_bvh()
    ctx.upload(sc)
    s = sc.settings()
    assert (s.width, s.height) == (ri.W, ri.H)
    st = dr.pack_settings13(s, 1)
    frames = 64
    ctx.accum_reset(ri.W, ri.H)
    ctx.render_accumulate(st, ri.W, ri.H, s.background, 1, 1000003, frames)
    stats = ri.compare(ri.display(ctx.accum_read(), frames), ri.reference_image())
    print(stats)
    ri.check(stats)


def test_gpu_reproduces_the_reference_image_with_textures(dr, ctx, tmp_path):
    """The reference's saved frame for samples/bolter2.blend.rts (albedo texture, environment map, camera after
    LEFT x3 DOWN x2) against 64 frames accumulated on the GPU."""
    import reference_image as ri
    path, texdir = ri.bolter_scene(tmp_path)
    sc = dr.Scene.load(path, texdir)
    sc.build_bvh()
    ctx.upload(sc)
    s = sc.settings()
    assert (s.width, s.height) == (ri.W, ri.H) and s.backtex >= 0
    s.campos[0] += ri.BOLTER_KEYS[0]; s.campos[1] += ri.BOLTER_KEYS[1]; s.campos[2] += ri.BOLTER_KEYS[2]
    st = dr.pack_settings13(s, 1)
    frames = 64
    ctx.accum_reset(ri.W, ri.H)
    ctx.render_accumulate(st, ri.W, ri.H, s.background, 1, 1000003, frames)
    stats = ri.compare_full(ri.display(ctx.accum_read(), frames), ri.bolter_reference_image())
    print(stats)
    ri.check_full(stats)


def test_getnormal_known_answers(dr, orc, ctx, synth, tmp_path):
    """getnormal K:703-773 by itself: flat face normals from the file, the -20 sentinels (cross product instead of the file's
    normal; no vertex normals), smooth interpolation, spheres (divide by the radius) and the unsupported type (normalise);
    normal and texture coordinate bitwise against the oracle for every object of each scene."""
    tri = "%s,2,0.8,0.2,0.2,0.3,0,%s,0,%s,%s,%s,0.1,0.2,0.9,0.1,0.5,0.8,%d,0,no,no"
    v0, v1, v2 = "-1.000000,-1.000000,0.500000", "1.000000,-0.500000,0.250000", "0.250000,1.000000,-0.300000"
    rows = [
        tri % (v0, v1, v2, "0.100000,0.200000,0.970000", "0.0,0.0,1.0,0.0,0.6,0.8,0.6,0.0,0.8", 0),          # file normal, flat
        tri % (v0, v1, v2, "0.100000,0.200000,0.970000", "0.0,0.0,1.0,0.0,0.6,0.8,0.6,0.0,0.8", 1),          # smooth: vertex normals
        tri % (v0, v1, v2, "0.100000,0.200000,-20.000000", "0.0,0.0,1.0,0.0,0.6,0.8,0.6,0.0,0.8", 1),        # face sentinel: cross product, smooth ignored
        tri % (v0, v1, v2, "0.100000,0.200000,0.970000", "0.0,0.0,-20.0,0.0,0.6,0.8,0.6,0.0,0.8", 1),        # n1.z sentinel: file normal although smooth
        "0.500000,0.250000,-2.000000,0,0.2,0.8,0.2,0.1,0,0.750000,0.000000,0.000000,0",                         # sphere, radius 0.75
        "-2.000000,0.250000,-1.000000,1,0.2,0.2,0.8,0.1,0,0.500000,0.000000,0.000000,0",                        # type 1: neither sphere nor triangle
    ]
    p = tmp_path / "normals.rts"
    p.write_text("*,0,0,5,0.0,0,0,0,5,40,4,1,1,no,64,64\n" + "\n".join(rows) + "\n")
    rng = np.random.default_rng(8)
    for path, tex in ((str(p), ""), (os.path.join(SCENES, "cow.rts"), synth["tex"]), (os.path.join(SCENES, "scene.rts"), ""),
                      (os.path.join(synth["dir"], "bunny_small.rts"), "")):
        ps, os_ = _load_both(dr, orc, path, tex)
        ctx.upload(ps)
        nobj = ps.num_objects
        n = max(2000, 4 * nobj)
        idx = (np.arange(n) % nobj).astype(np.int32)
        o = rng.uniform(-6, 6, size=(n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        t = rng.uniform(0.1, 9, size=n).astype(np.float32)
        gn, gt = ctx.kat_normal(idx, o, d, t)
        rn, rt = os_.kat_normal(idx, o, d, t)
        assert np.array_equal(bits(gn), bits(rn)), path
        assert np.array_equal(bits(gt), bits(rt)), path
    # the sentinels really select different branches on the hand-made scene
    ps, os_ = _load_both(dr, orc, str(p), "")
    ctx.upload(ps)
    o = np.tile(np.array([[0.1, 0.0, 4.0]], dtype=np.float32), (6, 1)); d = np.tile(np.array([[0.0, 0.05, -1.0]], dtype=np.float32), (6, 1))
    gn, _ = ctx.kat_normal(np.arange(6, dtype=np.int32), o, d, np.full(6, 3.5, dtype=np.float32))
    assert not np.array_equal(gn[0], gn[1]) and not np.array_equal(gn[0], gn[2]) and np.array_equal(gn[0], gn[3])


ALL_REFERENCE_SCENES = ["cam", "cow", "cube", "cubeold", "glass", "light", "lots", "mats", "norm", "rough.blend", "scene", "smooth",
                        "suzane", "textest", "uv", "whee"]


@pytest.mark.parametrize("name", ALL_REFERENCE_SCENES)
def test_every_committed_reference_scene_renders_like_the_oracle(dr, orc, ctx, synth, name):
    """Each of the sixteen sample scenes taken over from the reference (13- to 38-column generations, with and without a settings
    line, textures by name where a stand-in exists): one frame with the default kernel and traversal (persistent, wide walk) and
    one with the threaded walk, identical to the oracle's; the threaded walk's counters equal the oracle's."""
    path = os.path.join(SCENES, name + ".rts")
    for mode in (2, 0):
        g, r, stats, rc = _render_pair(dr, orc, ctx, path, synth["tex"], 256, 160, 1, 4242, mode=mode, kernel=1)
        _assert_frames(g, r, "%s traversal %d" % (name, mode))
        assert stats["rays"] == rc["rays"] and stats["shades"] == rc["S"] and stats["texels"] == rc["T"]
        if mode == 0:
            assert stats["node_visits"] == rc["V"] and stats["prim_tests"] == rc["L"]


def test_work_sharing_drain_renders_like_the_oracle(dr, orc, ctx, synth, tmp_path):
    """The drain phase of short launches hands subtrees of the rays still walking to idle lanes (shared best hit by ds_min_u64 on the
    (t, slot) key).  With coop_steps = 1 every ray that can be shared is: fuzzed scenes (coincident triangles: ties must still go
    to the lower slot, whichever lane finds them), spheres, textures, spp > 1 -- frames identical to the oracle's."""
    from scene_fuzz import random_scene
    rng = np.random.default_rng(123)
    names = ["synth_albedo.ppm", "synth_rough.ppm", "synth_env.ppm", "a.ppm"]
    cases = [(random_scene(rng, int(rng.integers(50, 900)), str(tmp_path / ("share%d.rts" % k)), W=96, H=64, textures=names), synth["tex"], 96, 64) for k in range(6)]
    cases += [(os.path.join(SCENES, "scene.rts"), "", 320, 192), (os.path.join(synth["dir"], "hf_small.rts"), "", 320, 192),
              (os.path.join(synth["dir"], "city_small.rts"), "", 200, 120)]
    defaults = {k: ctx.get_option(k) for k in ("coop_steps", "coop_lanes", "coop_rounds", "split_parts", "split_steps", "split_waves")}
    # split_parts / split_steps: tiles whose longest pixel took split_steps node steps in the previous frame are handed out in parts and
    # the rest of each wave helps from the start; _render_pair renders every frame twice (the second launch has the first one's costs),
    # so the split path runs
    combos = ({"coop_steps": 1, "coop_lanes": 64, "split_parts": 1}, {"coop_steps": 4, "coop_rounds": 1, "split_parts": 1}, {"coop_steps": 8, "coop_rounds": 4},
              {"coop_steps": 1, "split_parts": 4, "split_steps": 16, "coop_rounds": 3}, {"coop_steps": 2, "split_parts": 2, "split_steps": 32},
              {"split_parts": 8, "split_steps": 16, "split_waves": 50}, {"split_parts": 8, "split_steps": 64, "coop_steps": 1, "coop_rounds": 16, "split_waves": 1000})
    for combo in combos:
        for k, v in combo.items():
            ctx.set_option(k, v)
        try:
            for path, tex, W, H in cases:
                g, r, stats, rc = _render_pair(dr, orc, ctx, path, tex, W, H, 1, 4321, mode=2, kernel=1)
                _assert_frames(g, r, "%s work sharing %r" % (os.path.basename(path), combo))
        finally:
            for k, v in defaults.items():
                ctx.set_option(k, v)


def test_fuzz_campaign_slice(dr, orc, ctx, synth, tmp_path):
    """A 200-scene slice of tools/fuzz_campaign.py inside the suite: random scenes x random settings of the work-sharing drain, split
    tiles, occupancy, batching, tile-order refresh and the pipeline (streams, lean build) x frame sizes that are not multiples of 8;
    single frames three times, batched and pipelined accumulation -- everything equal to the oracle's frames."""
    from fuzz_driver import run_campaign
    names = ["synth_albedo.ppm", "synth_rough.ppm", "synth_env.ppm", "a.ppm"]
    n, frames, bad = run_campaign(dr, orc, ctx, 200, 20261004, str(tmp_path), synth["tex"], names)
    print("fuzz slice: %d scenes, %d frames, %d mismatching" % (n, frames, len(bad)))
    assert not bad, bad[:5]
