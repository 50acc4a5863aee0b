"""GPU box: a longer parity campaign than the test suite runs (tests/fuzz_driver.py: random scenes x random scheduling options x odd frame sizes,
single frames, batched and pipelined accumulation, every result against the oracle).
   python tools/fuzz_campaign.py [scenes=40] [seed=1]        FUZZ_SIZES=8,9,16,24,33: tiny frames"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dogeray_amd as dr
from oracle import orc
from fuzz_driver import run_campaign

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = tempfile.mkdtemp(prefix="dogeray_fuzz_")
tex = os.path.join(d, "tex"); os.makedirs(tex)
gen = os.path.join(ROOT, "tools", "scenegen")
names = ["synth_albedo.ppm", "synth_rough.ppm", "synth_env.ppm", "a.ppm"]
for name, w, h, k in ((names[0], 128, 128, 0), (names[1], 64, 64, 1), (names[2], 256, 128, 2), (names[3], 32, 32, 0)):
    subprocess.check_call([gen, "ppm", os.path.join(tex, name), str(w), str(h), str(k)])
sizes = [int(v) for v in os.environ["FUZZ_SIZES"].split(",")] if os.environ.get("FUZZ_SIZES") else None
ctx = dr.Context(0)
n, frames, bad = run_campaign(dr, orc, ctx, n_scenes, seed, d, tex, names, sizes=sizes, log=lambda m: print(m, flush=True))
print("fuzz campaign: %d scenes, %d frames, %d mismatching scenes" % (n, frames, len(bad)))
sys.exit(1 if bad else 0)
