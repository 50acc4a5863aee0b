import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLDEN, "scenes")
TEXTURES = os.path.join(GOLDEN, "textures")
SCENEGEN = os.path.join(ROOT, "tools", "scenegen")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle, the scene generator and the product library exist (compiles if stale)."""
    from oracle import orc
    orc.build()
    if not os.path.exists(SCENEGEN) or os.path.getmtime(SCENEGEN) < os.path.getmtime(SCENEGEN + ".cpp"):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", SCENEGEN, SCENEGEN + ".cpp"])
    from dogeray_amd import build as b
    b.build()                     # no-op when libdogeray_amd.so and bin/dogeray are newer than their sources
    return True


@pytest.fixture(scope="session")
def synth(tmp_path_factory):
    """Directory with generated scenes + textures (deterministic; see tools/scenegen.cpp)."""
    d = str(tmp_path_factory.mktemp("synth"))

    def gen(*args):
        subprocess.check_call([SCENEGEN] + [str(a) for a in args])

    tex = os.path.join(d, "tex")
    os.makedirs(tex)
    gen("ppm", os.path.join(tex, "synth_albedo.ppm"), 128, 128, 0)
    gen("ppm", os.path.join(tex, "synth_rough.ppm"), 64, 64, 1)
    gen("ppm", os.path.join(tex, "synth_env.ppm"), 256, 128, 2)
    # names the copied reference samples ask for (cow.rts, textest.rts, rough.blend.rts); the
    # reference's own files are MBs large, these are small stand-ins with the same names
    gen("ppm", os.path.join(tex, "testtwo.ppm"), 96, 64, 0)
    gen("ppm", os.path.join(tex, "bah.ppm"), 64, 64, 1)
    gen("ppm", os.path.join(tex, "env.ppm"), 128, 64, 2)
    with open(os.path.join(TEXTURES, "a.ppm"), "rb") as f:
        open(os.path.join(tex, "a.ppm"), "wb").write(f.read())
    gen("matball", os.path.join(d, "matball.rts"), 256, 256)
    gen("heightfield", os.path.join(d, "hf_small.rts"), 71, 320, 192)      # 9 800 triangles
    gen("bunnyish", os.path.join(d, "bunny_small.rts"), 3, 320, 192)       # 1 282 triangles
    gen("city", os.path.join(d, "city_small.rts"), 12, 320, 192)           # 1 730 triangles
    return {"dir": d, "tex": tex}


def with_settings(src, dst, line):
    """Copy a .rts, dropping its own '*' line(s) and prepending `line`."""
    with open(src) as f:
        body = [l for l in f.read().split("\n") if not l.startswith("*")]
    while body and body[-1] == "":
        body.pop()
    with open(dst, "w") as f:
        f.write(line + "\n" + "\n".join(body) + "\n")
    return dst


CUBE_SETTINGS = "*,7.358891,-6.925791,4.958309,0.01,0,0,0,3,45,10,1,1,no,256,256"


def frame_stats(a, b):
    """(fraction of pixels identical, max abs channel difference) between two int32[W,H,3] frames."""
    same = np.all(a == b, axis=2)
    return float(same.mean()), int(np.abs(a.astype(np.int64) - b.astype(np.int64)).max())
