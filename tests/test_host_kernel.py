"""The product's device functions compiled for the HOST (tools/host_kernel.cpp: dogeray_amd/csrc/device_core.hpp with -DDR_HOST_BUILD,
unchanged arithmetic) against the oracle, pixel for pixel -- a check of the kernel's source that needs no GPU, and the "host-C++
compile of the same kernel" north_star names as the CPU baseline (bench.py cpu_baseline kind "same-source").  Test infrastructure: the
product library has no CPU path."""
import os
import sys

import numpy as np
import pytest

from conftest import SCENES, CUBE_SETTINGS, with_settings, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def hk():
    import host_kernel
    host_kernel.build()
    return host_kernel


@pytest.fixture(scope="module")
def orc():
    from oracle import orc as o
    return o


def _both(hk, orc, path, tex, W, H, div, seed, traversal, spp=None, depth=None):
    a = orc.Scene(path, tex or None)
    a.build_bvh()
    st = a.settings()
    s13 = orc.settings13(st, div)
    if spp is not None:
        s13[10] = spp
    if depth is not None:
        s13[9] = depth
    want, wc = a.render(s13, W, H, st.background, seed, nthreads=4)
    b = hk.Scene(path, tex or "")
    got, gc = b.render(s13, W, H, st.background, seed, traversal=traversal, nthreads=4)
    return got, gc, want, wc


@pytest.mark.parametrize("traversal", [2, 0, 1])
def test_host_build_of_the_kernel_renders_like_the_oracle(hk, orc, synth, tmp_path, traversal):
    from scene_fuzz import random_scene
    rng = np.random.default_rng(5)
    names = ["synth_albedo.ppm", "synth_rough.ppm", "synth_env.ppm", "a.ppm"]
    cube = with_settings(os.path.join(SCENES, "cube.rts"), str(tmp_path / "cube.rts"), CUBE_SETTINGS)
    cases = [(cube, "", 256, 256, 1), (cube, "", 256, 256, 4), (os.path.join(SCENES, "scene.rts"), "", 160, 96, 1), (os.path.join(SCENES, "glass.rts"), "", 160, 96, 1),
             (os.path.join(SCENES, "rough.blend.rts"), synth["tex"], 160, 96, 1), (os.path.join(SCENES, "cow.rts"), synth["tex"], 160, 96, 1),
             (os.path.join(synth["dir"], "matball.rts"), synth["tex"], 128, 128, 1), (os.path.join(synth["dir"], "hf_small.rts"), "", 160, 96, 1)]
    cases += [(random_scene(rng, int(rng.integers(2, 500)), str(tmp_path / ("hk%d.rts" % k)), W=96, H=64, textures=names), synth["tex"], 96, 64, 1) for k in range(5)]
    # scenes a twentieth / a fiftieth of the size: triangles as small as the reference's 0.01 padding, which the wide tree enters with their own bounds
    cases += [(random_scene(rng, int(rng.integers(50, 600)), str(tmp_path / ("hks%d.rts" % k)), W=96, H=64, textures=names, scale=(0.05, 0.02, 0.2)[k]), synth["tex"], 96, 64, 1) for k in range(3)]
    for path, tex, W, H, div in cases:
        got, gc, want, wc = _both(hk, orc, path, tex, W, H, div, 4242, traversal)
        same = float(np.all(got == want, axis=2).mean())
        assert same == 1.0, "%s traversal %d: %.6f of pixels identical" % (os.path.basename(path), traversal, same)
        assert gc["rays"] == wc["rays"] and gc["S"] == wc["S"] and gc["T"] == wc["T"] and gc["samples"] == wc["samples"]
        if traversal == 0:       # the threaded walk visits in the reference's order: its counters are the reference's
            assert gc["V"] == wc["V"] and gc["L"] == wc["L"]


@pytest.mark.parametrize("n,ratio", [(64, 1.5), (200, 1.1)])
def test_deepest_wide_tree_renders_like_the_oracle(hk, orc, tmp_path, n, ratio):
    """a tree of exactly WIDE_MAX_DEPTH levels (the kernel's per-lane stack full to its last word) against the oracle and the threaded walk"""
    from scene_fuzz import stadium_scene
    path = stadium_scene(str(tmp_path / "stadium.rts"), n, ratio)
    sc = hk.Scene(path, "")
    assert sc.has_wide and sc.wide_depth() == 17
    for traversal in (2, 0):
        got, gc, want, wc = _both(hk, orc, path, "", 96, 64, 1, 99, traversal)
        assert np.array_equal(got, want) and gc["rays"] == wc["rays"] and gc["S"] == wc["S"]
        assert gc["rays"] > 96 * 64      # some paths bounce between the triangles


def test_restated_uniform_draws_round_like_the_plain_expressions(hk):
    """device_core.hpp computes curand_uniform_double(), (float)(u * 2 - 1) and the unit-sphere rejection test with fewer operations; every one of
    them must round exactly as the plain expression does: 20 million generator outputs, the corners of the 53-bit integer, every float within 2^-18
    of 1 for the rejection test"""
    assert hk.lib().hk_check_uniform(20_000_000, 12345) == 0


def test_merged_rejection_loop_draws_the_reference_points(hk):
    """rand_points_merged rejects candidates from a 32-bit approximation and converts only the accepted one exactly (out of the generator's state):
    on 3 million generator states it must return rand_in_unit_sphere's / rand_in_unit_disk's point bit for bit and leave the generator in the
    same state; the largest gap between the approximate and the exact squared length stays far inside the 2^-17 the proof allows"""
    import ctypes as C
    gap = C.c_double(0)
    assert hk.lib().hk_check_reject(3_000_000, 777, C.byref(gap)) == 0
    assert 0 < gap.value < 2.0 ** -19


def test_host_build_stripes_and_thread_counts_do_not_change_pixels(hk, orc, synth):
    path = os.path.join(synth["dir"], "city_small.rts")
    a = orc.Scene(path)
    a.build_bvh()
    st = a.settings()
    s13 = orc.settings13(st, 1)
    b = hk.Scene(path)
    full, _ = b.render(s13, 200, 120, st.background, 9, nthreads=1)
    want, _ = a.render(s13, 200, 120, st.background, 9, nthreads=4)
    assert np.array_equal(full, want)
    parts = np.zeros_like(full)
    for rem in range(3):
        p, _ = b.render(s13, 200, 120, st.background, 9, nthreads=3, col_mod=3, col_rem=rem)
        parts += p
    assert np.array_equal(parts, full)
