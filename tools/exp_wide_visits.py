"""Experiment (GPU box): records visited per ray by the wide walk (dr_kat_hit visits) on the bench scene's primary rays and on
random bounce rays from their hit points: mean / median / p99 / p99.9 / max, against the threaded walk's box tests."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dogeray_amd as dr
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/dogeray_bench/heightfield_709_1920x1080.rts"
sc = dr.Scene.load(path, ""); sc.build_bvh(); s = sc.settings()
ctx = dr.Context(0).upload(sc)
W, H = s.width, s.height
frm = np.array(s.campos[:], dtype=np.float64); at = np.array(s.look[:], dtype=np.float64)
w = (frm - at) / np.linalg.norm(frm - at); u = np.cross([0, 1, 0], w); u /= np.linalg.norm(u); v = np.cross(w, u)
vh = 2 * np.tan(np.radians(s.fov) / 2); vw = vh * W / H; f = s.focus_dist
hor, ver = f * vw * u, f * vh * v; llc = frm - hor / 2 - ver / 2 - f * w
xs, ys = np.meshgrid(np.arange(0, W, 2), np.arange(0, H, 2), indexing="ij")
nu, nv = (xs.ravel() + 0.5) / W, (ys.ravel() + 0.5) / H
d = (llc[None] + nu[:, None] * hor[None] + nv[:, None] * ver[None] - frm[None]).astype(np.float32)
o = np.repeat(frm[None].astype(np.float32), len(d), 0)
def show(name, o, d):
    out = {}
    for mode in (0, 2):
        ctx.set_traversal(mode)
        t, idx, vis = ctx.kat_hit(o, d, want_visits=True)
        out[mode] = (t, idx)
        print("%s traversal %d: mean %.1f median %.0f p99 %.0f p99.9 %.0f max %d  hits %.3f" % (name, mode, vis.mean(), np.median(vis), np.percentile(vis, 99), np.percentile(vis, 99.9), vis.max(), (t > 0).mean()))
        if mode == 2:
            worst = np.argsort(vis)[-3:]
            for i in worst: print("    worst ray: visits %d o %s d %s" % (vis[i], o[i], d[i]))
    assert np.array_equal(out[0][0].view(np.uint32), out[2][0].view(np.uint32)) and np.array_equal(out[0][1], out[2][1])
    return out[0][0]
t = show("primary rays", o, d)
ok = t > 0
rng = np.random.default_rng(0)
d2 = rng.normal(size=(ok.sum(), 3)).astype(np.float32); d2[:, 1] = -np.abs(d2[:, 1]); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
show("bounce rays", (o + t[:, None] * d)[ok].astype(np.float32), d2)
