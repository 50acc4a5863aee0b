"""Experiment (GPU box): the two-paths-per-lane kernel and the waves-with-roles kernel against the one-path kernel on the bench
scene: identical accumulators, ms/frame, wave-level step counts.  python tools/exp_paired.py [frames] [variants]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dogeray_amd as dr
path = bench.ensure_scene("/tmp/dogeray_bench", 709, 1920, 1080)
sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
st = dr.pack_settings13(s, 1, spp=1)
W, H = 1920, 1080
res = {}
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
VARIANTS = (("one-path", {"paired": 0, "roles": 0}), ("roles7", {"paired": 0, "roles": 7}), ("roles3", {"paired": 0, "roles": 3}), ("roles6", {"paired": 0, "roles": 6}), ("paired 48", {"paired": 1, "roles": 0, "pair_thresh": 48}))
if len(sys.argv) > 2: VARIANTS = tuple(v for v in VARIANTS if v[0] in sys.argv[2].split(",") or v[0] == "one-path")
for name, opts in VARIANTS:
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.accum_reset(W, H)
    ctx.render_accumulate(st, W, H, s.background, 1, 1000003, frames)       # warm-up, tile order
    best = 1e9
    for rep in range(5):
        ctx.accum_reset(W, H); ctx.stats_reset()
        ctx.render_accumulate(st, W, H, s.background, 1, 1000003, frames)
        stt = ctx.stats()
        best = min(best, stt["kernel_ms"] / frames)
    res[name] = ctx.accum_read()
    d = stt["diag"]
    print("%-10s %.4f ms/frame  clock %.0f MHz  wave-cycles/frame %.4g;  per frame: iterations %.3g phases %.3g (%.1f lanes served each) node steps %.3g leaf steps %.3g" % (
        name, best, 100.0 * d[0] / max(1, d[7]), d[0] / frames, d[2] / frames, d[3] / frames, d[6] / max(1, d[3]), d[4] / frames, d[5] / frames), flush=True)
for name in res:
    print(name, "identical to one-path:", bool(np.array_equal(res[name], res["one-path"])))
