"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (dogeray_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcObj(C.Structure):
    _fields_ = [("type", C.c_int32), ("pos", C.c_float * 3), ("rot", C.c_float * 3), ("norm", C.c_float * 3),
                ("n1", C.c_float * 3), ("n2", C.c_float * 3), ("n3", C.c_float * 3), ("t1", C.c_float * 3),
                ("t2", C.c_float * 3), ("t3", C.c_float * 3), ("smooth", C.c_int32), ("tex", C.c_int32),
                ("mat", C.c_int32), ("dim", C.c_float * 3), ("col", C.c_float * 3), ("texnum", C.c_int32),
                ("rtexnum", C.c_int32), ("addional", C.c_float * 3)]


OBJ_DTYPE = np.dtype([("type", "<i4"), ("pos", "<f4", 3), ("rot", "<f4", 3), ("norm", "<f4", 3), ("n1", "<f4", 3),
                      ("n2", "<f4", 3), ("n3", "<f4", 3), ("t1", "<f4", 3), ("t2", "<f4", 3), ("t3", "<f4", 3),
                      ("smooth", "<i4"), ("tex", "<i4"), ("mat", "<i4"), ("dim", "<f4", 3), ("col", "<f4", 3),
                      ("texnum", "<i4"), ("rtexnum", "<i4"), ("addional", "<f4", 3)])
assert OBJ_DTYPE.itemsize == C.sizeof(OrcObj) == 168


class OrcSettings(C.Structure):
    _fields_ = [("campos", C.c_float * 3), ("look", C.c_float * 3), ("aperture", C.c_float),
                ("focus_dist", C.c_float), ("fov", C.c_int32), ("max_depth", C.c_int32), ("spp", C.c_int32),
                ("background", C.c_float), ("backtex", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


class OrcCounters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("V", C.c_uint64), ("L", C.c_uint64), ("S", C.c_uint64), ("T", C.c_uint64),
                ("samples", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(force=False):
    """Compile oracle/liboracle*.so with g++ (make)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


_libs = {}


def lib(variant=""):
    name = "liboracle%s.so" % variant
    if name in _libs:
        return _libs[name]
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
    L.orc_last_error.restype = C.c_char_p
    L.orc_scene_load.restype = vp
    L.orc_scene_load.argtypes = [C.c_char_p, C.c_char_p]
    L.orc_scene_free.argtypes = [vp]
    for f in ("orc_num_objects", "orc_num_textures", "orc_build_bvh", "orc_bvh_size", "orc_bvh_used"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_int
    L.orc_texture_info.argtypes = [vp, C.c_int, ip, ip]
    L.orc_texture_data.argtypes = [vp, C.c_int, C.c_void_p]
    L.orc_get_settings.argtypes = [vp, C.POINTER(OrcSettings)]
    L.orc_get_objects.argtypes = [vp, C.c_void_p]
    L.orc_get_bvh.argtypes = [vp] + [C.c_void_p] * 10
    L.orc_render.argtypes = [vp, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_void_p,
                             C.POINTER(OrcCounters), C.c_int, C.c_int, C.c_int]
    L.orc_render.restype = C.c_int
    L.orc_render_visits.argtypes = L.orc_render.argtypes + [C.c_void_p]
    L.orc_render_visits.restype = C.c_int
    L.orc_kat_rng.argtypes = [C.c_uint64, C.c_int, C.c_void_p]
    L.orc_kat_rng_u32.argtypes = [C.c_uint64, C.c_int, C.c_void_p]
    L.orc_kat_aabb.argtypes = [C.c_int] + [C.c_void_p] * 6
    L.orc_kat_tri.argtypes = [C.c_int] + [C.c_void_p] * 6
    L.orc_kat_sphere.argtypes = [C.c_int] + [C.c_void_p] * 5
    L.orc_kat_hit.argtypes = [vp, C.c_int] + [C.c_void_p] * 4
    L.orc_kat_normal.argtypes = [vp, C.c_int] + [C.c_void_p] * 6
    L.orc_kat_optics.argtypes = [C.c_int] + [C.c_void_p] * 6
    _libs[name] = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Scene:
    """One parsed .rts scene held by the oracle."""

    def __init__(self, rts_path, texdir=None, variant=""):
        self.L = lib(variant)
        self.h = self.L.orc_scene_load(os.fsencode(rts_path), os.fsencode(texdir) if texdir else None)
        if not self.h:
            raise RuntimeError(self.L.orc_last_error().decode())
        self.n = self.L.orc_num_objects(self.h)

    def close(self):
        if self.h:
            self.L.orc_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def settings(self):
        s = OrcSettings()
        self.L.orc_get_settings(self.h, C.byref(s))
        return s

    def objects(self):
        out = np.zeros(self.n + 1, dtype=OBJ_DTYPE)
        self.L.orc_get_objects(self.h, _p(out))
        return out

    def textures(self):
        res = []
        for i in range(self.L.orc_num_textures(self.h)):
            w, h = C.c_int32(), C.c_int32()
            self.L.orc_texture_info(self.h, i, C.byref(w), C.byref(h))
            a = np.zeros((h.value, w.value, 4), dtype=np.uint8)
            self.L.orc_texture_data(self.h, i, _p(a))
            res.append(a)
        return res

    def build_bvh(self):
        if self.L.orc_build_bvh(self.h) != 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        n = self.L.orc_bvh_size(self.h)
        names = ["active", "child0", "child1", "count", "hit", "miss", "under", "end"]
        arrs = {k: np.zeros(n, dtype=np.int32) for k in names}
        arrs["min"] = np.zeros((n, 3), dtype=np.float32)
        arrs["max"] = np.zeros((n, 3), dtype=np.float32)
        self.L.orc_get_bvh(self.h, *[_p(arrs[k]) for k in names + ["min", "max"]])
        arrs["used"] = self.L.orc_bvh_used(self.h)
        return arrs

    def render(self, settings13, W, H, bgint, frame_seed, nthreads=1, col_mod=1, col_rem=0):
        """One CudaStarter-equivalent frame.  Returns (int32[W, H, 3] indexed [x, y], counters dict)."""
        st = _f32(settings13)
        assert st.shape == (13,)
        out = np.zeros((W, H, 3), dtype=np.int32)
        c = OrcCounters()
        rc = self.L.orc_render(self.h, _p(st), W, H, float(bgint), int(frame_seed) & (2 ** 64 - 1), _p(out),
                               C.byref(c), nthreads, col_mod, col_rem)
        if rc != 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return out, c.as_dict()

    def render_visits(self, settings13, W, H, bgint, frame_seed, nthreads=1, col_mod=1, col_rem=0):
        """render() plus uint32[W, H] node visits per pixel."""
        st = _f32(settings13)
        out = np.zeros((W, H, 3), dtype=np.int32)
        vis = np.zeros((W, H), dtype=np.uint32)
        c = OrcCounters()
        rc = self.L.orc_render_visits(self.h, _p(st), W, H, float(bgint), int(frame_seed) & (2 ** 64 - 1), _p(out),
                                      C.byref(c), nthreads, col_mod, col_rem, _p(vis))
        if rc != 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return out, c.as_dict(), vis

    def kat_normal(self, obj_index, o, d, t):
        idx = np.ascontiguousarray(obj_index, dtype=np.int32)
        o, d, t = _f32(o), _f32(d), _f32(t)
        n = idx.shape[0]
        nrm = np.zeros((n, 3), dtype=np.float32)
        tc = np.zeros((n, 3), dtype=np.float32)
        self.L.orc_kat_normal(self.h, n, _p(idx), _p(o), _p(d), _p(t), _p(nrm), _p(tc))
        return nrm, tc

    def kat_hit(self, o, d):
        o, d = _f32(o), _f32(d)
        n = o.shape[0]
        t = np.zeros(n, dtype=np.float32)
        idx = np.zeros(n, dtype=np.int32)
        self.L.orc_kat_hit(self.h, n, _p(o), _p(d), _p(t), _p(idx))
        return t, idx


def settings13(s, divisor, spp=None, depth=None):
    """The settings[13] array CudaStarter packs (K:2581) from an OrcSettings-like object."""
    return np.array([s.campos[0], s.campos[1], s.campos[2], s.look[0], s.look[1], s.look[2], s.aperture,
                     s.focus_dist, s.fov, s.max_depth if depth is None else depth, s.spp if spp is None else spp,
                     divisor, s.backtex], dtype=np.float32)


def kat_rng(seed, n):
    out = np.zeros(n, dtype=np.float64)
    lib().orc_kat_rng(seed, n, _p(out))
    return out


def kat_rng_u32(seed, n):
    out = np.zeros(n, dtype=np.uint32)
    lib().orc_kat_rng_u32(seed, n, _p(out))
    return out


def kat_aabb(o, d, mn, mx):
    o, d, mn, mx = map(_f32, (o, d, mn, mx))
    n = o.shape[0]
    hit = np.zeros(n, dtype=np.int32)
    dist = np.zeros(n, dtype=np.float32)
    lib().orc_kat_aabb(n, _p(o), _p(d), _p(mn), _p(mx), _p(hit), _p(dist))
    return hit, dist


def kat_tri(o, d, v0, v1, v2):
    o, d, v0, v1, v2 = map(_f32, (o, d, v0, v1, v2))
    n = o.shape[0]
    t = np.zeros(n, dtype=np.float32)
    lib().orc_kat_tri(n, _p(o), _p(d), _p(v0), _p(v1), _p(v2), _p(t))
    return t


def kat_sphere(o, d, c, r):
    o, d, c, r = map(_f32, (o, d, c, r))
    n = o.shape[0]
    t = np.zeros(n, dtype=np.float32)
    lib().orc_kat_sphere(n, _p(o), _p(d), _p(c), _p(r), _p(t))
    return t


def kat_optics(v, nrm, eta):
    v, nrm, eta = map(_f32, (v, nrm, eta))
    n = v.shape[0]
    refl = np.zeros((n, 3), dtype=np.float32)
    refr = np.zeros((n, 3), dtype=np.float32)
    sch = np.zeros(n, dtype=np.float32)
    lib().orc_kat_optics(n, _p(v), _p(nrm), _p(eta), _p(refl), _p(refr), _p(sch))
    return refl, refr, sch
