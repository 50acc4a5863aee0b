"""Experiment: per-ray box tests of the threaded vs ordered traversal on the bench scene's primary rays
and on random secondary-like rays (run on the GPU box)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dogeray_amd as dr
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
path = "/tmp/dogeray_bench/heightfield_709_1920x1080.rts"
sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
W, H = 1920, 1080
# pinhole camera rays, one per pixel on a 4x-subsampled grid (K:1016-1073 without jitter/lens)
frm = np.array(s.campos[:], dtype=np.float64); at = np.array(s.look[:], dtype=np.float64)
w = (frm - at) / np.linalg.norm(frm - at); u = np.cross([0, 1, 0], w); u /= np.linalg.norm(u); v = np.cross(w, u)
vh = 2 * np.tan(np.radians(s.fov) / 2); vw = vh * W / H; f = s.focus_dist
hor, ver = f * vw * u, f * vh * v; llc = frm - hor / 2 - ver / 2 - f * w
xs, ys = np.meshgrid(np.arange(0, W, 2), np.arange(0, H, 2), indexing="ij")
nu, nv = (xs.ravel() + 0.5) / W, (ys.ravel() + 0.5) / H
d = (llc[None] + nu[:, None] * hor[None] + nv[:, None] * ver[None] - frm[None]).astype(np.float32)
o = np.repeat(frm[None].astype(np.float32), len(d), 0)
res = {}
for mode in (0, 1):
    ctx.set_traversal(mode)
    t, idx, vis = ctx.kat_hit(o, d, want_visits=True)
    res[mode] = (t, idx, vis)
    print("primary rays mode %d: mean %.1f median %.0f p99 %.0f p99.9 %.0f max %d  hits %.3f" % (mode, vis.mean(), np.median(vis), np.percentile(vis, 99), np.percentile(vis, 99.9), vis.max(), (t > 0).mean()))
assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32)) and np.array_equal(res[0][1], res[1][1])
heavy = res[0][2] > 500
print("rays with > 500 threaded visits: %d; their ordered visits: mean %.1f max %d" % (heavy.sum(), res[1][2][heavy].mean() if heavy.any() else 0, res[1][2][heavy].max() if heavy.any() else 0))
# secondary-like rays: start on the surface (hit points), random hemisphere directions
hp = o + res[0][0][:, None] * d
ok = res[0][0] > 0
rng = np.random.default_rng(0)
d2 = rng.normal(size=(ok.sum(), 3)).astype(np.float32); d2[:, 1] = -np.abs(d2[:, 1])
d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
o2 = hp[ok].astype(np.float32)
for mode in (0, 1):
    ctx.set_traversal(mode)
    t, idx, vis = ctx.kat_hit(o2, d2, want_visits=True)
    res[mode] = (t, idx, vis)
    print("bounce rays mode %d: mean %.1f median %.0f p99 %.0f p99.9 %.0f max %d  hits %.3f" % (mode, vis.mean(), np.median(vis), np.percentile(vis, 99), np.percentile(vis, 99.9), vis.max(), (t > 0).mean()))
assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32)) and np.array_equal(res[0][1], res[1][1])
heavy = res[0][2] > 500
print("bounce rays with > 500 threaded visits: %d; their ordered visits: mean %.1f max %d" % (heavy.sum(), res[1][2][heavy].mean() if heavy.any() else 0, res[1][2][heavy].max() if heavy.any() else 0))
