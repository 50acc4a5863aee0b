"""BASELINE.json's full-size configuration on the GPU (C4 stand-in: 1 002 528 triangles, 1920x1080):
size-independent properties + the oracle on a sampled set of block columns.  The scene file is the one
bench.py generates and caches (tools/scenegen heightfield)."""
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

W, H = 1920, 1080


OTHER_CONFIGS = [
    ("C2 bunnyish 1280x720", ["bunnyish", "6", "1280", "720"], 1280, 720, False, 81922),
    ("C5 city 3840x2160", ["city", "200", "3840", "2160"], 3840, 2160, False, 480002),
    ("C3 textured 1920x1080", ["matball", "1920", "1080"], 1920, 1080, True, 87),
]


@pytest.fixture(scope="module")
def other_scenes(tmp_path_factory, synth):
    """The stand-in scenes of the other configs, generated before anything in this module touches the GPU."""
    import subprocess
    from conftest import SCENEGEN
    d = tmp_path_factory.mktemp("configs")
    paths = {}
    for name, gen_args, *_ in OTHER_CONFIGS:
        p = str(d / (gen_args[0] + ".rts"))
        subprocess.check_call([SCENEGEN, gen_args[0], p] + gen_args[1:])
        paths[name] = p
    return paths


@pytest.fixture(scope="module")
def big(tmp_path_factory, other_scenes):
    import sys
    sys.path.insert(0, ROOT)
    import bench
    import dogeray_amd as dr
    assert dr.device_count() >= 1
    path = bench.ensure_scene(os.environ.get("DOGERAY_BENCH_CACHE", "/tmp/dogeray_bench"), 709, W, H)
    scene = dr.Scene.load(path, "")
    assert scene.num_objects == 1002528
    scene.build_bvh()
    ctx = dr.Context(0).upload(scene)
    yield dr, scene, ctx, path
    ctx.close()


def test_bvh_is_the_reference_shape_at_1m(big):
    dr, scene, ctx, path = big
    nodes, used = scene.bvh()
    n = scene.num_objects
    assert used == 2 * n - 1 and len(nodes) == 2 * (n + 1)
    live = nodes[nodes["active"] == 1]
    leaf = live[live["end"] == 1]
    assert len(leaf) == n and len(np.unique(leaf["under"])) == n
    inner = live[live["end"] == 0]
    assert np.array_equal(inner["children"][:, 1], inner["children"][:, 0] + 1)
    depth = int(np.ceil(np.log2(n)))
    assert depth == 20


def test_frame_properties(big):
    dr, scene, ctx, path = big
    s = scene.settings()
    st = dr.pack_settings13(s, 1, spp=1)
    seed = 1 + 1000003 * 4
    ctx.set_option("kernel", 1)
    a = ctx.render_frame(st, W, H, s.background, seed)
    assert np.array_equal(a, ctx.render_frame(st, W, H, s.background, seed))          # idempotent: same seed, same frame
    assert not np.array_equal(a, ctx.render_frame(st, W, H, s.background, seed + 1))  # and the seed matters
    ctx.set_option("kernel", 0)
    assert np.array_equal(a, ctx.render_frame(st, W, H, s.background, seed))          # per-tile kernel: same frame
    for mode in (1, 0):
        ctx.set_traversal(mode)
        assert np.array_equal(a, ctx.render_frame(st, W, H, s.background, seed))      # ordered / threaded traversal (tile kernel): same frame
    ctx.set_option("kernel", 1)
    assert np.array_equal(a, ctx.render_frame(st, W, H, s.background, seed))          # threaded walk, persistent kernel
    ctx.set_traversal(2)
    assert ctx.get_option("traversal") == 2 and 10 <= ctx.get_option("wide_depth") <= 17
    # stripes of 8 ranks compose to the full frame
    total = np.zeros_like(a)
    for r in range(8):
        ctx.set_stripe(8, r)
        total += ctx.render_frame(st, W, H, s.background, seed)
    ctx.set_stripe(1, 0)
    assert np.array_equal(total, a)
    # a plausible image: the terrain fills roughly half the frame, nothing negative, sky pixels bright
    assert a.min() >= 0 and 0.3 < (a.sum(axis=2) > 0).mean() <= 1.0


def test_accumulation_is_the_sum_of_frames(big):
    dr, scene, ctx, path = big
    s = scene.settings()
    st = dr.pack_settings13(s, 1, spp=1)
    total = np.zeros((W, H, 3), dtype=np.int64)
    for k in range(5):
        total += ctx.render_frame(st, W, H, s.background, 77 + 1000003 * k)
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 77, 1000003, 5)      # one batched launch
    assert np.array_equal(ctx.accum_read().astype(np.int64), total)
    img = ctx.accum_present(5)
    want = np.clip(total // 5, 0, 255).astype(np.uint8).transpose(1, 0, 2)
    assert np.array_equal(img, want)


def test_sampled_columns_match_the_oracle(big):
    """Every 24th 8-pixel block column of one full-size frame, pixel for pixel, plus the counters."""
    dr, scene, ctx, path = big
    from oracle import orc
    osc = orc.Scene(path)
    osc.build_bvh()
    s = scene.settings()
    st = dr.pack_settings13(s, 1, spp=1)
    seed = 424242
    mod = 24
    ref, rc = osc.render(st, W, H, s.background, seed, nthreads=os.cpu_count() or 8, col_mod=mod, col_rem=5)
    ctx.set_stripe(mod, 5)
    wide = ctx.render_frame(st, W, H, s.background, seed)          # the default traversal (wide walk)
    ctx.set_traversal(0)                                           # threaded walk: its counters are the reference's
    ctx.enable_counters(True)
    ctx.stats_reset()
    got = ctx.render_frame(st, W, H, s.background, seed)
    stats = ctx.stats()
    ctx.enable_counters(False)
    ctx.set_traversal(2)
    ctx.set_stripe(1, 0)
    assert np.array_equal(wide, got)
    same = float(np.all(got == ref, axis=2).mean())
    print("sampled columns: %.6f of pixels identical; %d rays" % (same, rc["rays"]))
    assert same == 1.0
    assert (stats["rays"], stats["node_visits"], stats["prim_tests"], stats["shades"]) == (rc["rays"], rc["V"], rc["L"], rc["S"])


# The other BASELINE.json configs at their full sizes (stand-ins per SURVEY 8(d)): parity-test cases, not bench lines.
@pytest.mark.parametrize("name,gen_args,Wc,Hc,tex,tris", OTHER_CONFIGS)
def test_other_configs_full_size(other_scenes, synth, name, gen_args, Wc, Hc, tex, tris):
    import dogeray_amd as dr
    from oracle import orc
    path = other_scenes[name]
    texdir = synth["tex"] if tex else ""
    scene = dr.Scene.load(path, texdir)
    assert scene.num_objects == tris
    scene.build_bvh()
    s = scene.settings()
    assert (s.width, s.height) == (Wc, Hc)
    ctx = dr.Context(0).upload(scene)
    st = dr.pack_settings13(s, 1)
    a = ctx.render_frame(st, Wc, Hc, s.background, 31337)
    ctx.set_option("kernel", 0)
    assert np.array_equal(a, ctx.render_frame(st, Wc, Hc, s.background, 31337))
    for mode in (1, 0):
        ctx.set_traversal(mode)
        assert np.array_equal(a, ctx.render_frame(st, Wc, Hc, s.background, 31337))
    ctx.set_traversal(2)
    ctx.set_option("kernel", 1)
    ctx.accum_reset(Wc, Hc)
    ctx.render_accumulate(st, Wc, Hc, s.background, 31337, 1000003, 4)
    acc = ctx.accum_read().astype(np.int64)
    assert np.array_equal(acc[..., :] - a, sum(ctx.render_frame(st, Wc, Hc, s.background, 31337 + 1000003 * k).astype(np.int64) for k in (1, 2, 3)))
    # the oracle on every 16th block column
    osc = orc.Scene(path, texdir if tex else None)
    osc.build_bvh()
    ref, _ = osc.render(st, Wc, Hc, s.background, 31337, nthreads=os.cpu_count() or 8, col_mod=16, col_rem=3)
    cols = (np.arange(Wc) // 8) % 16 == 3
    same = float(np.all(a[cols] == ref[cols], axis=2).mean())
    print("%s: %.6f of sampled pixels identical to the oracle" % (name, same))
    assert same == 1.0
    ctx.close()


def test_c3_bolter2_with_its_textures_at_1920x1080(tmp_path):
    """SURVEY 8(d) C3 as specified: samples/bolter2.blend.rts (3 892 triangles, 37 columns) with samples/boltersmall.ppm
    (1000x1000) and samples/env.ppm (800x600) at 1920x1080 -- the fixtures of tests/golden/reference_image -- default
    kernel + wide walk against the tile kernel and the threaded walk, accumulation, and the oracle on every 12th block column."""
    import dogeray_amd as dr
    import reference_image as ri
    from oracle import orc
    path, texdir = ri.bolter_scene(tmp_path)
    scene = dr.Scene.load(path, texdir)
    assert scene.num_objects == 3892 and len(scene.textures()) == 2
    scene.build_bvh()
    s = scene.settings()
    assert s.backtex >= 0
    Wc, Hc = 1920, 1080
    ctx = dr.Context(0).upload(scene)
    st = dr.pack_settings13(s, 1)
    a = ctx.render_frame(st, Wc, Hc, s.background, 2024)
    ctx.set_traversal(0)
    assert np.array_equal(a, ctx.render_frame(st, Wc, Hc, s.background, 2024))
    ctx.set_option("kernel", 0)
    assert np.array_equal(a, ctx.render_frame(st, Wc, Hc, s.background, 2024))
    ctx.set_option("kernel", 1)
    ctx.set_traversal(2)
    ctx.accum_reset(Wc, Hc)
    ctx.render_accumulate(st, Wc, Hc, s.background, 2024, 1000003, 3)
    acc = ctx.accum_read().astype(np.int64)
    assert np.array_equal(acc - a, sum(ctx.render_frame(st, Wc, Hc, s.background, 2024 + 1000003 * k).astype(np.int64) for k in (1, 2)))
    osc = orc.Scene(path, texdir)
    osc.build_bvh()
    ref, rc = osc.render(st, Wc, Hc, s.background, 2024, nthreads=os.cpu_count() or 8, col_mod=12, col_rem=7)
    cols = (np.arange(Wc) // 8) % 12 == 7
    same = float(np.all(a[cols] == ref[cols], axis=2).mean())
    print("C3 bolter2 1920x1080: %.6f of sampled pixels identical to the oracle, %d texel fetches on the sample" % (same, rc["T"]))
    assert same == 1.0 and rc["T"] > 10000
    ctx.close()
