#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_golden.json.

There are NO reference-produced vectors for this path: the reference has no tests and seeds its RNG
from clock() (kernel.cu K:1065), and it cannot be built in this image.  What is pinned here is the
ORACLE ITSELF (oracle/dogeray_oracle.cpp): digests of its BVH arrays and of its int3 frames for the
copied sample scenes and two generated ones, so that an accidental change to the restatement -- the
thing every GPU parity test is measured against -- shows up as a test failure.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orc          # noqa: E402
from conftest import CUBE_SETTINGS, SCENES, SCENEGEN, with_settings   # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def bvh_digest(b):
    live = b["active"] == 1
    leaf = live & (b["end"] == 1)
    inner = live & (b["end"] == 0)
    return digest(b["active"], b["count"][live], b["end"][live], b["hit"][live], b["miss"][live], b["min"][live], b["max"][live],
                  b["under"][leaf], b["child0"][inner], b["child1"][inner])


def cases(tmp):
    out = []
    cube = with_settings(os.path.join(SCENES, "cube.rts"), os.path.join(tmp, "cube256.rts"), CUBE_SETTINGS)
    out.append(("cube256", cube, None, 256, 256))
    for name in ("scene.rts", "lots.rts", "glass.rts", "mats.rts", "smooth.rts", "cubeold.rts", "norm.rts", "uv.rts", "whee.rts"):
        out.append((name, os.path.join(SCENES, name), None, 160, 96))
    subprocess.check_call([SCENEGEN, "heightfield", os.path.join(tmp, "hf.rts"), "41", "160", "96"])
    out.append(("heightfield41", os.path.join(tmp, "hf.rts"), None, 160, 96))
    tex = os.path.join(tmp, "tex")
    os.makedirs(tex, exist_ok=True)
    subprocess.check_call([SCENEGEN, "ppm", os.path.join(tex, "synth_albedo.ppm"), "128", "128", "0"])
    subprocess.check_call([SCENEGEN, "ppm", os.path.join(tex, "synth_rough.ppm"), "64", "64", "1"])
    subprocess.check_call([SCENEGEN, "ppm", os.path.join(tex, "synth_env.ppm"), "256", "128", "2"])
    subprocess.check_call([SCENEGEN, "matball", os.path.join(tmp, "matball.rts"), "160", "96"])
    out.append(("matball", os.path.join(tmp, "matball.rts"), tex, 160, 96))
    return out


def compute():
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, path, tex, W, H in cases(tmp):
            s = orc.Scene(path, tex)
            st = s.settings()
            entry = {"objects": digest(s.objects()), "n": s.n}
            if s.n >= 2:
                entry["bvh"] = bvh_digest(s.build_bvh())
                frames = {}
                for div, seed in ((1, 1), (1, 1 + 1000003), (4, 7)):
                    img, c = s.render(orc.settings13(st, div), W, H, st.background, seed, nthreads=4)
                    frames["div%d_seed%d" % (div, seed)] = {"sha256": digest(img), "counters": c}
                entry["frames"] = frames
            res[name] = entry
    res["xorwow"] = {"seed0_first8_u32": [int(x) for x in orc.kat_rng_u32(0, 8)],
                     "seed12345_first4_double_bits": [int(x) for x in orc.kat_rng(12345, 4).view(np.uint64)]}
    return res


if __name__ == "__main__":
    orc.build()
    data = compute()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_golden.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print("wrote", path)
