"""CPU tests of the product's host side (C++ behind the C ABI): .rts reader, texture discovery,
bit-exact BVH builder, error behaviour -- each against the oracle on the same files -- and the
C ABI surface itself (every symbol of include/dogeray_amd.h is exported; no silent CPU fallback)."""
import ctypes
import glob
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, SCENES, SCENEGEN


@pytest.fixture(scope="module")
def dr():
    import dogeray_amd
    return dogeray_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import orc as o
    return o


SCENE_FILES = sorted(glob.glob(os.path.join(SCENES, "*.rts")))


def assert_same_objects(ps, os_):
    po, oo = ps.objects(), os_.objects()
    assert po.dtype.itemsize == oo.dtype.itemsize == 168
    assert po.tobytes() == oo.tobytes()
    assert bytes(ps.settings()) == bytes(os_.settings())


def assert_same_bvh(ps, os_):
    ob = os_.build_bvh()
    ps.build_bvh()
    pb, used = ps.bvh()
    assert used == ob["used"] and len(pb) == len(ob["hit"])
    live = ob["active"] == 1
    leaf = live & (ob["end"] == 1)
    inner = live & (ob["end"] == 0)
    assert np.array_equal(pb["active"], ob["active"])
    for mine, theirs, mask in (("count", "count", live), ("end", "end", live), ("hit_node", "hit", live), ("miss_node", "miss", live),
                               ("under", "under", leaf)):
        assert np.array_equal(pb[mine][mask], ob[theirs][mask]), mine
    assert np.array_equal(pb["children"][inner][:, 0], ob["child0"][inner])
    assert np.array_equal(pb["children"][inner][:, 1], ob["child1"][inner])
    assert pb["min"][live].tobytes() == ob["min"][live].tobytes()      # float bounds bit for bit
    assert pb["max"][live].tobytes() == ob["max"][live].tobytes()


@pytest.mark.parametrize("path", SCENE_FILES, ids=[os.path.basename(p) for p in SCENE_FILES])
def test_reader_and_bvh_match_oracle_on_samples(dr, orc, synth, path):
    """Every copied reference sample (13/16/19/28/36/37/38-column generations, partial lines,
    spheres, texture names) parses and builds to the same bytes as the oracle."""
    ps = dr.Scene.load(path, synth["tex"])
    os_ = orc.Scene(path, synth["tex"])
    assert ps.num_objects == os_.n
    assert_same_objects(ps, os_)
    if os_.n >= 2:
        assert_same_bvh(ps, os_)


def test_texture_discovery_and_name_resolution(dr, orc, synth):
    ps = dr.Scene.load(os.path.join(SCENES, "cow.rts"), synth["tex"])
    os_ = orc.Scene(os.path.join(SCENES, "cow.rts"), synth["tex"])
    pt, ot = ps.textures(), os_.textures()
    assert len(pt) == len(ot) == 7
    for a, b in zip(pt, ot):
        assert a.shape == b.shape and np.array_equal(a, b)
        assert not a[..., 3].any()                       # sdkLoadPPM4: alpha = 0
    objs = ps.objects()
    assert set(np.unique(objs["texnum"][:-1])) == {0, 6}   # a.ppm and testtwo.ppm in sorted directory order
    # capitals in the query never match the lower-cased path (K:1177-1178)
    d = os.path.join(synth["dir"], "caps")
    os.makedirs(d, exist_ok=True)
    subprocess.check_call([SCENEGEN, "ppm", os.path.join(d, "Checker.ppm"), "8", "8", "0"])
    p = os.path.join(synth["dir"], "caps.rts")
    line = "0,0,0,2,1,1,1,0,0,1,0,0,0,0,1,0,0,0,1,0,0,1,0,0,1,0,0,1,0,0,1,0,0,1,0,0,%s,no\n"
    open(p, "w").write(line % "Checker.ppm" + line % "checker.ppm")
    for sc in (dr.Scene.load(p, d).objects(), orc.Scene(p, d).objects()):
        assert list(sc["texnum"][:2]) == [-1, 0]
    assert dr.Scene.load(p, "").objects()["texnum"][1] == -1   # "" = no texture directory


def test_synthetic_scenes_match_oracle(dr, orc, synth):
    for name in ("matball.rts", "hf_small.rts", "bunny_small.rts", "city_small.rts"):
        path = os.path.join(synth["dir"], name)
        ps, os_ = dr.Scene.load(path, synth["tex"]), orc.Scene(path, synth["tex"])
        assert_same_objects(ps, os_)
        assert_same_bvh(ps, os_)


def test_bvh_thread_count_does_not_matter(dr, synth, tmp_path):
    path = str(tmp_path / "hf.rts")
    subprocess.check_call([SCENEGEN, "heightfield", path, "150", "320", "192"])   # 44 402 triangles: above the fan-out threshold
    ps = dr.Scene.load(path)
    ps.build_bvh(1)
    a, _ = ps.bvh()
    for t in (2, 3, 8, 0):
        ps.build_bvh(t)
        b, _ = ps.bvh()
        assert a.tobytes() == b.tobytes(), t


def test_reader_edge_cases(dr, orc, tmp_path):
    tri = "1,2,3,2,0.5,0.5,0.5,0,0,1,1,1,0,2,2,2"
    cases = {
        "crlf.rts": tri + "\r\n" + tri + "\r\n",                          # '\r' stays in the last field; stof ignores it
        "noeol.rts": tri + "\n" + tri,                                     # last line without newline
        "comments.rts": "/a comment\n" + tri + "\n/another\n*,1,2,3,0.02,0,0,0,5,60.9,7.2,3\n" + tri + "\n",
        "short.rts": "00\n.1\n" + tri + "\n",                              # partial lines keep the struct defaults
        "spaces.rts": " 1, 2,3 ,2,0.5,0.5,0.5,0,0,1,1,1,0,2,2,2\n" + tri + "\n",   # stof skips leading blanks, ignores trailing ones
        "extra.rts": tri + "," + ",".join(["0"] * 22) + ",no,no,99,junk\n" + tri + "\n",   # columns past 37 are never converted
        "settings_all.rts": "*,1,2,3,0.02,4,5,6,7,50,9,2,0.5,no,640,360\n" + tri + "\n" + tri + "\n",
    }
    for name, text in cases.items():
        p = tmp_path / name
        p.write_bytes(text.encode())
        ps, os_ = dr.Scene.load(str(p)), orc.Scene(str(p))
        assert ps.num_objects == os_.n, name
        assert_same_objects(ps, os_)
    s = dr.Scene.load(str(tmp_path / "comments.rts")).settings()
    assert (s.fov, s.max_depth, s.spp) == (60, 7, 3) and abs(s.aperture - 0.02) < 1e-9     # stoi("60.9") == 60
    s = dr.Scene.load(str(tmp_path / "settings_all.rts")).settings()
    assert (s.width, s.height, s.backtex) == (640, 360, -1)
    d = dr.Scene.load(str(tmp_path / "short.rts")).objects()
    assert d["norm"][0].tolist() == [-2, -3, -20] and d["texnum"][0] == -1 and d["t1"][0].tolist() == [0, 1, 0]   # K:55-71


def test_error_behaviour(dr, tmp_path):
    """Where the reference throws or recurses forever, the C ABI returns a negative status."""
    tri = "1,2,3,2,0.5,0.5,0.5,0,0,1,1,1,0,2,2,2"
    bad = {"empty_line.rts": tri + "\n\n" + tri + "\n", "trailing_comma.rts": tri + ",\n" + tri + "\n",
           "letters.rts": "abc,2,3,2\n" + tri + "\n", "bad_int.rts": "1,2,3,x\n" + tri + "\n"}
    for name, text in bad.items():
        p = tmp_path / name
        p.write_text(text)
        with pytest.raises(dr.DogerayError) as e:
            dr.Scene.load(str(p))
        assert e.value.code == -3, name                       # DR_ERR_PARSE
        assert "line" in str(e.value)
    with pytest.raises(dr.DogerayError) as e:
        dr.Scene.load(str(tmp_path / "nope.rts"))
    assert e.value.code == -2                                 # DR_ERR_IO
    with pytest.raises(dr.DogerayError) as e:
        dr.Scene.load(str(tmp_path / "empty_line.rts"), str(tmp_path / "no_such_dir"))
    assert e.value.code == -2
    one = tmp_path / "one.rts"
    one.write_text(tri + "\n")
    s = dr.Scene.load(str(one))
    with pytest.raises(dr.DogerayError) as e:
        s.build_bvh()
    assert e.value.code == -4                                 # DR_ERR_SCENE


def test_random_field_is_reproducible(dr, tmp_path):
    p = tmp_path / "r.rts"
    p.write_text("r,r,r,2,r,0.5,0.5,0,0,1,1,1,0,2,2,2\n1,2,3,2,0.5,0.5,0.5,0,0,1,1,1,0,2,2,2\n")
    a, b = dr.Scene.load(str(p)).objects(), dr.Scene.load(str(p)).objects()
    assert a.tobytes() == b.tobytes()
    assert 0 <= a["pos"][0][0] < 1 and a["pos"][0][0] != a["pos"][0][1]


def test_scene_from_caller_arrays(dr, synth):
    """The CudaStarter-level entry: objects (+ optionally the caller's own BVH) handed over as arrays."""
    src = dr.Scene.load(os.path.join(SCENES, "cow.rts"), synth["tex"])
    src.build_bvh()
    nodes, used = src.bvh()
    a = dr.Scene.from_arrays(src.objects(), src.settings(), bvh=nodes, textures=src.textures())
    b = dr.Scene.from_arrays(src.objects(), src.settings(), textures=src.textures())
    b.build_bvh()
    for sc in (a, b):
        assert sc.num_objects == src.num_objects
        assert sc.objects().tobytes() == src.objects().tobytes()
        assert sc.bvh()[0].tobytes() == nodes.tobytes() and sc.bvh()[1] == used
        assert len(sc.textures()) == 7 and np.array_equal(sc.textures()[6], src.textures()[6])
    with pytest.raises(dr.DogerayError):
        dr.Scene.from_arrays(src.objects(), src.settings(), bvh=nodes[:-2])


# ------------------------------------------------------------------------------ C ABI surface
def header_functions():
    text = open(os.path.join(ROOT, "include", "dogeray_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(dr):
    names = header_functions()
    assert len(names) >= 35
    L = ctypes.CDLL(os.path.join(ROOT, "dogeray_amd", "libdogeray_amd.so"))
    for n in names:
        assert hasattr(L, n), "libdogeray_amd.so does not export %s" % n
    assert sorted(dr.API_SYMBOLS) == names, "python binding and header disagree"
    assert dr.lib().dr_abi_version() == 2


def test_struct_layouts_match_header(dr):
    assert ctypes.sizeof(dr.DrObject) == 168 and dr.OBJECT_DTYPE.itemsize == 168
    assert dr.BVH_DTYPE.itemsize == 56
    assert ctypes.sizeof(dr.DrSettings) == 60


def test_no_cpu_fallback(dr):
    """Without a GPU the device entry points fail loudly; nothing renders on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert dr.device_count() == 0
    with pytest.raises(dr.DogerayError) as e:
        dr.Context(0)
    assert e.value.code == -5 and "no CPU fallback" in str(e.value)


def test_product_never_touches_the_oracle():
    """dogeray_amd/ must not import, link or open anything under oracle/."""
    for root, _, files in os.walk(os.path.join(ROOT, "dogeray_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "liboracle" not in text and "dogeray_oracle" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
    out = subprocess.run(["ldd", os.path.join(ROOT, "dogeray_amd", "libdogeray_amd.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_host_application_reports_missing_gpu_or_renders(tmp_path):
    """The C++ application (dogeray_main.cpp) speaks only the C ABI: with a GPU it renders and exports,
    without one it stops at context creation with the library's error (no CPU rendering)."""
    exe = os.path.join(ROOT, "dogeray_amd", "bin", "dogeray")
    assert os.path.exists(exe)
    out = str(tmp_path / "cube.bmp")
    r = subprocess.run([exe, os.path.join(SCENES, "cube.rts"), "--textures", "", "--frames", "3", "--width", "128", "--height", "128",
                        "--out", out, "--quiet"], capture_output=True, text=True, timeout=120)
    assert "97 tris" in r.stdout and "194 nodes total" in r.stdout          # objnum = 96 + 1, bvhnum = 2 * 97 (K:2056,2094)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stderr
        data = open(out, "rb").read()
        assert data[:2] == b"BM" and len(data) == 122 + 128 * 128 * 4          # the layout SDL_SaveBMP gives the reference's export
        ref_hdr = open(os.path.join(GOLDEN, "reference_image", "eorovan.blend.rts.bmp.header"), "rb").read()
        same = [i for i in range(122) if i not in range(2, 6) and i not in range(18, 26) and i not in range(34, 38)]    # all but the three size fields
        assert all(data[i] == ref_hdr[i] for i in same)
    else:
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


def test_host_application_cache(tmp_path, dr):
    """dogeray --cache writes scene.rtsb after the build and starts from it the second time (same object/node report)."""
    import shutil
    exe = os.path.join(ROOT, "dogeray_amd", "bin", "dogeray")
    scene = shutil.copy(os.path.join(SCENES, "cube.rts"), str(tmp_path / "cube.rts"))
    cmd = [exe, scene, "--textures", "", "--frames", "1", "--width", "64", "--height", "64", "--cache"]
    first = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert os.path.exists(scene + "b") and "taken from" not in first.stdout
    cached = dr.Scene.load_binary(scene + "b")
    fresh = dr.Scene.load(scene, "")
    fresh.build_bvh()
    assert cached.bvh()[0].tobytes() == fresh.bvh()[0].tobytes() and cached.objects().tobytes() == fresh.objects().tobytes()
    second = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert "taken from" in second.stdout and "97 tris" in second.stdout and "194 nodes total" in second.stdout
    assert first.returncode == second.returncode


def test_fuzzed_scenes_parse_and_build_like_the_oracle(dr, orc, synth, tmp_path):
    """Random scenes (mixed column counts, spheres, duplicates, degenerate triangles): objects and BVH identical."""
    from scene_fuzz import random_scene
    rng = np.random.default_rng(2024)
    names = ["synth_albedo.ppm", "synth_rough.ppm", "synth_env.ppm", "a.ppm"]
    for k in range(25):
        n = int(rng.integers(2, 400))
        path = random_scene(rng, n, str(tmp_path / ("fuzz%d.rts" % k)), textures=names)
        ps, os_ = dr.Scene.load(path, synth["tex"]), orc.Scene(path, synth["tex"])
        assert ps.num_objects == os_.n == n
        assert_same_objects(ps, os_)
        assert_same_bvh(ps, os_)


def test_rts_writer_round_trip(dr, synth, tmp_path):
    """dogeray_amd.rts_io.write_rts -> Scene.load reproduces the scene (values that six decimals carry)."""
    from dogeray_amd import rts_io
    src = dr.Scene.load(os.path.join(synth["dir"], "matball.rts"), synth["tex"])
    objs = src.objects()[:-1]
    names = ["a.ppm", "bah.ppm", "env.ppm", "synth_albedo.ppm", "synth_env.ppm", "synth_rough.ppm", "testtwo.ppm"]   # sorted directory order
    out = rts_io.write_rts(str(tmp_path / "rt.rts"), objs, src.settings(), texture_names=names)
    assert rts_io.validate_rts(out) == []
    back = dr.Scene.load(out, synth["tex"])
    assert back.num_objects == src.num_objects
    assert back.objects().tobytes() == src.objects().tobytes()       # the generator wrote %f too, so nothing is lost
    assert bytes(back.settings()) == bytes(src.settings())
    # shorter generations of the format keep the struct defaults for the missing columns
    short = dr.Scene.load(rts_io.write_rts(str(tmp_path / "rt16.rts"), objs, None, ncols=16), "").objects()
    assert np.array_equal(short["pos"], src.objects()["pos"]) and short["texnum"].max() == -1 and short["norm"][0].tolist() == [-2, -3, -20]
    bad = tmp_path / "bad.rts"
    bad.write_text("1,2,x,2,0.5\n")
    assert len(rts_io.validate_rts(str(bad))) == 2


def test_rtsb_round_trip(dr, synth, tmp_path):
    """.rtsb sidecar: the reloaded scene is the loaded + built scene byte for byte; damaged files are refused."""
    src = dr.Scene.load(os.path.join(synth["dir"], "matball.rts"), synth["tex"])
    unbuilt = dr.Scene.load_binary(src.save_binary(str(tmp_path / "unbuilt.rtsb")))
    assert unbuilt.bvh()[0].size == 0 and unbuilt.objects().tobytes() == src.objects().tobytes()
    src.build_bvh()
    path = src.save_binary(str(tmp_path / "m.rtsb"))
    back = dr.Scene.load_binary(path)
    assert back.num_objects == src.num_objects
    assert back.objects().tobytes() == src.objects().tobytes()
    assert bytes(back.settings()) == bytes(src.settings())
    assert back.bvh()[1] == src.bvh()[1] and back.bvh()[0].tobytes() == src.bvh()[0].tobytes()
    assert len(back.textures()) == len(src.textures()) > 0
    assert all(a.tobytes() == b.tobytes() for a, b in zip(back.textures(), src.textures()))
    unbuilt.build_bvh()                                   # a cache saved before the build builds the same tree
    assert unbuilt.bvh()[0].tobytes() == src.bvh()[0].tobytes()
    raw = open(path, "rb").read()
    cases = {"flip": raw[:5000] + bytes([raw[5000] ^ 1]) + raw[5001:], "cut": raw[:-7], "long": raw + b"x", "magic": b"X" + raw[1:], "empty": b""}
    for name, data in cases.items():
        p = tmp_path / (name + ".rtsb")
        p.write_bytes(data)
        with pytest.raises(dr.DogerayError) as e:
            dr.Scene.load_binary(str(p))
        assert e.value.code == dr.ERR_PARSE, name
    with pytest.raises(dr.DogerayError) as e:
        dr.Scene.load_binary(str(tmp_path / "missing.rtsb"))
    assert e.value.code == dr.ERR_IO


def test_every_option_is_documented_in_the_header():
    """Every name dr_context_set_option / dr_context_get_option accepts (csrc/context.cpp) appears, quoted, in include/dogeray_amd.h."""
    import re
    src = open(os.path.join(ROOT, "dogeray_amd", "csrc", "context.cpp")).read()
    hdr = open(os.path.join(ROOT, "include", "dogeray_amd.h")).read()
    names = set(re.findall(r'name == "(\w+)"', src)) | set(re.findall(r'\bn == "(\w+)"', src))
    assert len(names) > 20
    missing = sorted(n for n in names if '"%s"' % n not in hdr)
    assert not missing, missing
