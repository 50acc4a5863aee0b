"""dogeray_amd -- MI355X-native render path for DOGERAY scenes.

Thin ctypes binding of libdogeray_amd.so (C ABI in include/dogeray_amd.h).  All rendering
happens in the HIP library; there is no Python or CPU fallback: if the shared library is
missing or no GPU is present the calls below raise.

The Python surface mirrors the reference's host flow (kernel.cu main(), K:2021-2557):
    Scene.load(path, texture_dir)   ~ getnum + getppm* + read            (K:2055-2071)
    scene.build_bvh()               ~ build_bvh                          (K:2091)
    Context(device).upload(scene)   ~ what CudaStarter re-uploads per frame (K:2618-2629)
    ctx.render_frame(settings13, ...)  ~ CudaStarter(outputr, ..., divisor) (K:2562)
    ProgressiveRenderer             ~ the present loop's preview ladder + accumulation (K:2154-2224)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("DOGERAY_AMD_LIB") or os.path.join(_HERE, "libdogeray_amd.so")      # the override serves kernel experiments

TRAVERSAL_THREADED = 0
TRAVERSAL_ORDERED = 1
TRAVERSAL_WIDE = 2          # the default: 4-way tree over the reference's leaves
KERNEL_TILE = 0
KERNEL_PERSISTENT = 1
ERR_INVALID, ERR_IO, ERR_PARSE, ERR_SCENE, ERR_DEVICE, ERR_NOMEM = -1, -2, -3, -4, -5, -6      # enum dr_status


class DogerayError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dogeray_amd error %d: %s" % (code, msg))
        self.code = code


class DrObject(C.Structure):
    _fields_ = [("type", C.c_int32), ("pos", C.c_float * 3), ("rot", C.c_float * 3), ("norm", C.c_float * 3),
                ("n1", C.c_float * 3), ("n2", C.c_float * 3), ("n3", C.c_float * 3), ("t1", C.c_float * 3),
                ("t2", C.c_float * 3), ("t3", C.c_float * 3), ("smooth", C.c_int32), ("tex", C.c_int32),
                ("mat", C.c_int32), ("dim", C.c_float * 3), ("col", C.c_float * 3), ("texnum", C.c_int32),
                ("rtexnum", C.c_int32), ("addional", C.c_float * 3)]


OBJECT_DTYPE = np.dtype([("type", "<i4"), ("pos", "<f4", 3), ("rot", "<f4", 3), ("norm", "<f4", 3),
                         ("n1", "<f4", 3), ("n2", "<f4", 3), ("n3", "<f4", 3), ("t1", "<f4", 3), ("t2", "<f4", 3),
                         ("t3", "<f4", 3), ("smooth", "<i4"), ("tex", "<i4"), ("mat", "<i4"), ("dim", "<f4", 3),
                         ("col", "<f4", 3), ("texnum", "<i4"), ("rtexnum", "<i4"), ("addional", "<f4", 3)])
BVH_DTYPE = np.dtype([("active", "<i4"), ("children", "<i4", 2), ("count", "<i4"), ("hit_node", "<i4"),
                      ("miss_node", "<i4"), ("under", "<i4"), ("min", "<f4", 3), ("max", "<f4", 3), ("end", "<i4")])


class DrSettings(C.Structure):
    _fields_ = [("campos", C.c_float * 3), ("look", C.c_float * 3), ("aperture", C.c_float),
                ("focus_dist", C.c_float), ("fov", C.c_int32), ("max_depth", C.c_int32), ("spp", C.c_int32),
                ("background", C.c_float), ("backtex", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


class DrStats(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("launches", C.c_uint64), ("samples", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("prim_tests", C.c_uint64), ("shades", C.c_uint64), ("texels", C.c_uint64), ("kernel_ms", C.c_double),
                ("trav_slots", C.c_uint64), ("ray_slots", C.c_uint64), ("diag", C.c_uint64 * 8)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["diag"] = list(self.diag)
        return d


# every symbol include/dogeray_amd.h declares: (name, restype, argtypes)
_VP = C.c_void_p
_API = [
    ("dr_last_error", C.c_char_p, []),
    ("dr_abi_version", C.c_int, []),
    ("dr_scene_load", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_VP)]),
    ("dr_scene_free", None, [_VP]),
    ("dr_scene_create_from_arrays", C.c_int, [_VP, C.c_int, C.POINTER(DrSettings), _VP, C.c_int, C.POINTER(_VP)]),
    ("dr_scene_add_texture", C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_char_p]),
    ("dr_scene_num_objects", C.c_int, [_VP]),
    ("dr_scene_get_objects", C.c_int, [_VP, _VP]),
    ("dr_scene_get_settings", C.c_int, [_VP, C.POINTER(DrSettings)]),
    ("dr_scene_set_settings", C.c_int, [_VP, C.POINTER(DrSettings)]),
    ("dr_scene_num_textures", C.c_int, [_VP]),
    ("dr_scene_texture_info", C.c_int, [_VP, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("dr_scene_texture_data", C.c_int, [_VP, C.c_int, _VP]),
    ("dr_scene_build_bvh", C.c_int, [_VP, C.c_int]),
    ("dr_scene_bvh_size", C.c_int, [_VP]),
    ("dr_scene_bvh_used", C.c_int, [_VP]),
    ("dr_scene_get_bvh", C.c_int, [_VP, _VP]),
    ("dr_scene_save_binary", C.c_int, [_VP, C.c_char_p]),
    ("dr_scene_load_binary", C.c_int, [C.c_char_p, C.POINTER(_VP)]),
    ("dr_device_count", C.c_int, []),
    ("dr_context_create", C.c_int, [C.c_int, C.POINTER(_VP)]),
    ("dr_context_destroy", None, [_VP]),
    ("dr_context_upload_scene", C.c_int, [_VP, _VP]),
    ("dr_context_set_stripe", C.c_int, [_VP, C.c_int, C.c_int]),
    ("dr_context_set_traversal", C.c_int, [_VP, C.c_int]),
    ("dr_context_set_option", C.c_int, [_VP, C.c_char_p, C.c_int]),
    ("dr_context_get_option", C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int)]),
    ("dr_render_frame", C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_float, C.c_uint64, _VP]),
    ("dr_accum_reset", C.c_int, [_VP, C.c_int, C.c_int]),
    ("dr_render_accumulate", C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_int]),
    ("dr_accum_read", C.c_int, [_VP, _VP]),
    ("dr_accum_present", C.c_int, [_VP, C.c_int, _VP]),
    ("dr_accum_device_ptr", C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_uint64)]),
    ("dr_render_accumulate_async", C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_int]),
    ("dr_context_synchronize", C.c_int, [_VP]),
    ("dr_pipeline_submit", C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]),
    ("dr_pipeline_wait", C.c_int, [_VP, C.c_uint64, _VP]),
    ("dr_pipeline_image", C.c_int, [_VP, C.c_uint64, C.POINTER(_VP)]),
    ("dr_render_accumulate_pipelined", C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_int]),
    ("dr_context_stream", C.c_int, [_VP, C.POINTER(_VP)]),
    ("dr_accum_pack_stripe", C.c_int, [_VP, C.c_int, C.POINTER(_VP), C.POINTER(C.c_uint64)]),
    ("dr_accum_unpack_stripes", C.c_int, [_VP, _VP, C.c_uint64, C.c_int, C.c_int, _VP]),
    ("dr_accum_reserve_pack", C.c_int, [_VP, C.c_int]),
    ("dr_group_create", C.c_int, [C.c_int, _VP, C.POINTER(_VP)]),
    ("dr_group_destroy", None, [_VP]),
    ("dr_group_size", C.c_int, [_VP]),
    ("dr_group_uses_rccl", C.c_int, [_VP]),
    ("dr_group_rccl_ranks", C.c_int, [_VP]),
    ("dr_group_context", _VP, [_VP, C.c_int]),
    ("dr_group_upload_scene", C.c_int, [_VP, _VP]),
    ("dr_group_accum_reset", C.c_int, [_VP, C.c_int, C.c_int]),
    ("dr_group_render_accumulate", C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_int, C.c_int]),
    ("dr_stats_enable_counters", C.c_int, [_VP, C.c_int]),
    ("dr_stats_reset", C.c_int, [_VP]),
    ("dr_stats_get", C.c_int, [_VP, C.POINTER(DrStats)]),
    ("dr_context_probe_trace", C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("dr_stats_phase_counts", C.c_int, [_VP, C.POINTER(C.c_ulonglong), C.c_int]),
    ("dr_stats_wave_log", C.c_int, [_VP, C.POINTER(C.c_ulonglong), C.c_int, C.POINTER(C.c_int)]),
    ("dr_stats_pixel_cost", C.c_int, [_VP, C.POINTER(C.c_uint), C.c_size_t, C.POINTER(C.c_size_t)]),
    ("dr_context_probe_gather", C.c_int, [_VP, C.c_uint32, C.c_int, C.POINTER(C.c_double)]),
    ("dr_kat_rng", C.c_int, [_VP, C.c_uint64, C.c_int, _VP]),
    ("dr_kat_aabb", C.c_int, [_VP, C.c_int] + [_VP] * 6),
    ("dr_kat_tri", C.c_int, [_VP, C.c_int] + [_VP] * 6),
    ("dr_kat_node_planes", C.c_int, [_VP, C.c_int] + [_VP] * 5),
    ("dr_kat_sphere", C.c_int, [_VP, C.c_int] + [_VP] * 5),
    ("dr_kat_optics", C.c_int, [_VP, C.c_int] + [_VP] * 6),
    ("dr_kat_hit", C.c_int, [_VP, C.c_int] + [_VP] * 5),
    ("dr_kat_normal", C.c_int, [_VP, C.c_int] + [_VP] * 6),
]
API_SYMBOLS = [a[0] for a in _API]

_lib = None


def lib():
    """The loaded libdogeray_amd.so.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError("libdogeray_amd.so is missing: run `python -m dogeray_amd.build` "
                              "(or __graft_entry__.build()); dogeray_amd has no fallback path")
        # One process must hold ONE HIP runtime.  PyTorch-ROCm ships its own libamdhip64; if this
        # library pulled in /opt/rocm's copy first, torch's later initialisation finds "no HIP GPUs".
        # Importing torch first (when it is installed) makes both resolve to the same runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_LIB_PATH)
        for name, res, args in _API:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise DogerayError(rc, lib().dr_last_error().decode(errors="replace"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count():
    return lib().dr_device_count()


def pack_settings13(s, divisor, spp=None, depth=None):
    """float settings[13] exactly as CudaStarter packs it (K:2581)."""
    return np.array([s.campos[0], s.campos[1], s.campos[2], s.look[0], s.look[1], s.look[2], s.aperture,
                     s.focus_dist, s.fov, s.max_depth if depth is None else depth, s.spp if spp is None else spp,
                     divisor, s.backtex], dtype=np.float32)


class Scene:
    """Host scene: objects, settings, textures, BVH (allobjects / nbvhtree / globals of the reference)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load(cls, rts_path, texture_dir=""):
        """texture_dir: directory scanned for *ppm* entries; None = process cwd (reference behaviour),
        "" = none."""
        h = _VP()
        td = None if texture_dir is None else os.fsencode(texture_dir)
        _check(lib().dr_scene_load(os.fsencode(rts_path), td, C.byref(h)))
        return cls(h)

    @classmethod
    def load_binary(cls, rtsb_path):
        """A scene saved with save_binary(): objects, settings, textures and (if it was built) the BVH."""
        h = _VP()
        _check(lib().dr_scene_load_binary(os.fsencode(rtsb_path), C.byref(h)))
        return cls(h)

    def save_binary(self, rtsb_path):
        _check(lib().dr_scene_save_binary(self._h, os.fsencode(rtsb_path)))
        return rtsb_path

    @classmethod
    def from_arrays(cls, objects, settings=None, bvh=None, textures=()):
        """objects: OBJECT_DTYPE[N + 1]; bvh: BVH_DTYPE[2 * (N + 1)] or None; textures: uint8[h, w, 4] arrays."""
        objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
        n = len(objects) - 1
        h = _VP()
        bp, bn = None, 0
        if bvh is not None:
            bvh = np.ascontiguousarray(bvh, dtype=BVH_DTYPE)
            bp, bn = _p(bvh), len(bvh)
        _check(lib().dr_scene_create_from_arrays(_p(objects), n, C.byref(settings) if settings is not None else None, bp, bn, C.byref(h)))
        sc = cls(h)
        for i, t in enumerate(textures):
            t = np.ascontiguousarray(t, dtype=np.uint8)
            rc = lib().dr_scene_add_texture(h, _p(t), t.shape[1], t.shape[0], ("texture%d" % i).encode())
            if rc < 0:
                _check(rc)
        return sc

    def close(self):
        if self._h:
            lib().dr_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_objects(self):
        return lib().dr_scene_num_objects(self._h)

    def objects(self):
        out = np.zeros(self.num_objects + 1, dtype=OBJECT_DTYPE)
        _check(lib().dr_scene_get_objects(self._h, _p(out)))
        return out

    def settings(self):
        s = DrSettings()
        _check(lib().dr_scene_get_settings(self._h, C.byref(s)))
        return s

    def set_settings(self, s):
        _check(lib().dr_scene_set_settings(self._h, C.byref(s)))

    def textures(self):
        res = []
        for i in range(lib().dr_scene_num_textures(self._h)):
            w, h = C.c_int(), C.c_int()
            _check(lib().dr_scene_texture_info(self._h, i, C.byref(w), C.byref(h)))
            a = np.zeros((h.value, w.value, 4), dtype=np.uint8)
            _check(lib().dr_scene_texture_data(self._h, i, _p(a)))
            res.append(a)
        return res

    def build_bvh(self, nthreads=0):
        _check(lib().dr_scene_build_bvh(self._h, nthreads))

    def bvh(self):
        n = lib().dr_scene_bvh_size(self._h)
        out = np.zeros(n, dtype=BVH_DTYPE)
        _check(lib().dr_scene_get_bvh(self._h, _p(out)))
        return out, lib().dr_scene_bvh_used(self._h)


class Context:
    """One GPU with a resident scene; render_frame() is the CudaStarter replacement."""

    def __init__(self, device=0):
        h = _VP()
        _check(lib().dr_context_create(device, C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            lib().dr_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene):
        _check(lib().dr_context_upload_scene(self._h, scene._h))
        return self

    def set_stripe(self, mod, rem):
        _check(lib().dr_context_set_stripe(self._h, mod, rem))

    def set_traversal(self, mode):
        _check(lib().dr_context_set_traversal(self._h, mode))

    def set_option(self, name, value):
        """Tuning knob ("kernel", "batch_frames", "feedback", "occupancy", "schedule", ...: include/dogeray_amd.h); never changes a pixel."""
        _check(lib().dr_context_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int()
        _check(lib().dr_context_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def render_frame(self, settings13, W, H, background, frame_seed, download=True):
        """Returns int32[W, H, 3] indexed [x, y] (the reference's column-major int3 buffer) or None."""
        st = _f32(settings13)
        assert st.shape == (13,)
        out = np.empty((W, H, 3), dtype=np.int32) if download else None
        _check(lib().dr_render_frame(self._h, _p(st), W, H, float(background), int(frame_seed) & (2 ** 64 - 1),
                                     _p(out) if download else None))
        return out

    def accum_reset(self, W, H):
        _check(lib().dr_accum_reset(self._h, W, H))
        self._acc_shape = (W, H, 3)

    def render_accumulate(self, settings13, W, H, background, frame_seed, seed_stride, nframes):
        st = _f32(settings13)
        _check(lib().dr_render_accumulate(self._h, _p(st), W, H, float(background), int(frame_seed) & (2 ** 64 - 1),
                                          int(seed_stride) & (2 ** 64 - 1), nframes))

    def render_accumulate_async(self, settings13, W, H, background, frame_seed, seed_stride, nframes):
        """Queues the launches and returns (at most two batches in flight); synchronize() waits and updates stats()."""
        st = _f32(settings13)
        _check(lib().dr_render_accumulate_async(self._h, _p(st), W, H, float(background), int(frame_seed) & (2 ** 64 - 1),
                                                int(seed_stride) & (2 ** 64 - 1), nframes))

    def pipeline_submit(self, settings13, W, H, background, frame_seed, present_divide_by=0):
        """Queues one frame of the pipelined present loop (dr_pipeline_submit); returns its ticket."""
        st = _f32(settings13)
        t = C.c_uint64(0)
        _check(lib().dr_pipeline_submit(self._h, _p(st), W, H, float(background), int(frame_seed) & (2 ** 64 - 1), int(present_divide_by), C.byref(t)))
        return int(t.value)

    def pipeline_wait(self, ticket, want_image=False, in_place=False):
        """Waits for a submitted frame; with want_image the RGB8 image [H, W, 3] of exactly the frames up to that ticket (in_place: a
        read-only view of the library's pinned download buffer, valid until pipe_streams + 1 more GROUPS of frames have been launched)."""
        if not want_image:
            _check(lib().dr_pipeline_wait(self._h, int(ticket), None))
            return None
        W, H, _ = self._acc_shape
        if in_place:
            _check(lib().dr_pipeline_wait(self._h, int(ticket), None))
            ptr = _VP()
            _check(lib().dr_pipeline_image(self._h, int(ticket), C.byref(ptr)))
            buf = (C.c_uint8 * (W * H * 3)).from_address(ptr.value)
            img = np.frombuffer(buf, dtype=np.uint8).reshape(H, W, 3)
            img.flags.writeable = False
            return img
        img = np.empty((H, W, 3), dtype=np.uint8)
        _check(lib().dr_pipeline_wait(self._h, int(ticket), _p(img)))
        return img

    def render_accumulate_pipelined(self, settings13, W, H, background, frame_seed, seed_stride, nframes):
        st = _f32(settings13)
        _check(lib().dr_render_accumulate_pipelined(self._h, _p(st), W, H, float(background), int(frame_seed) & (2 ** 64 - 1),
                                                    int(seed_stride) & (2 ** 64 - 1), nframes))

    def synchronize(self):
        _check(lib().dr_context_synchronize(self._h))

    def stream_ptr(self):
        """The context's hipStream_t as an integer (torch.cuda.ExternalStream(ptr) orders torch work against it)."""
        p = _VP()
        _check(lib().dr_context_stream(self._h, C.byref(p)))
        return p.value or 0

    def accum_pack_stripe(self, slot):
        """Queues the packing of this context's stripe into library buffer `slot` (0/1); returns (device pointer, bytes)."""
        p, n = _VP(), C.c_uint64()
        _check(lib().dr_accum_pack_stripe(self._h, slot, C.byref(p), C.byref(n)))
        return p.value, n.value

    def accum_unpack_stripes(self, packed_ptr, rank_stride_bytes, world, first_rank=1, stream_ptr=None):
        _check(lib().dr_accum_unpack_stripes(self._h, C.c_void_p(packed_ptr), int(rank_stride_bytes), world, first_rank,
                                             C.c_void_p(stream_ptr) if stream_ptr else None))

    def accum_read(self):
        out = np.empty(self._acc_shape, dtype=np.int32)
        _check(lib().dr_accum_read(self._h, _p(out)))
        return out

    def accum_present(self, divide_by):
        """uint8[H, W, 3] row-major image: clamp(acc / divide_by, 0, 255) (K:2287)."""
        W, H, _ = self._acc_shape
        out = np.empty((H, W, 3), dtype=np.uint8)
        _check(lib().dr_accum_present(self._h, divide_by, _p(out)))
        return out

    def accum_device_ptr(self):
        p, n = _VP(), C.c_uint64()
        _check(lib().dr_accum_device_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def enable_counters(self, on=True):
        _check(lib().dr_stats_enable_counters(self._h, 1 if on else 0))

    def stats_reset(self):
        _check(lib().dr_stats_reset(self._h))

    def stats(self):
        s = DrStats()
        _check(lib().dr_stats_get(self._h, C.byref(s)))
        return s.as_dict()

    def phase_counts(self, n=32):
        """The shade / refill phase's budget counters of the counting build (dr_stats_phase_counts)."""
        buf = (C.c_ulonglong * n)()
        _check(lib().dr_stats_phase_counts(self._h, buf, n))
        return [int(v) for v in buf]

    def wave_log(self, max_waves=16384):
        """(n, 16) uint64: begin, queue-empty, end stamps (100 MHz ticks) and iterations after the queue was empty, per wave of the
        last persistent launch; needs set_option("wave_log", 1) before the launch."""
        out = np.zeros((max_waves, 16), dtype=np.uint64)
        n = C.c_int()
        _check(lib().dr_stats_wave_log(self._h, out.ctypes.data_as(C.POINTER(C.c_ulonglong)), max_waves, C.byref(n)))
        return out[:n.value]

    def pixel_cost(self, W, H):
        """(W, H) uint32 node steps per pixel of the last frame the persistent kernel recorded (empty array: none yet)."""
        gx, gy = (W + 7) // 8, (H + 7) // 8
        out = np.zeros(gx * gy * 64, dtype=np.uint32)
        n = C.c_size_t()
        _check(lib().dr_stats_pixel_cost(self._h, out.ctypes.data_as(C.POINTER(C.c_uint)), out.size, C.byref(n)))
        if n.value < out.size:
            return np.zeros((0, 0), dtype=np.uint32)
        return out.reshape(gx, gy, 8, 8).transpose(0, 2, 1, 3).reshape(gx * 8, gy * 8)[:W, :H]

    def probe_trace(self, settings13, W, H, background, frame_seed, frames=2, variant=2):
        """Trace-only probe (dr_context_probe_trace): (rays per second, rays, results differing from the one-ray-per-lane walk)."""
        st = _f32(settings13)
        r = C.c_double(); n = C.c_uint64(); bad = C.c_uint64()
        _check(lib().dr_context_probe_trace(self._h, _p(st), W, H, float(background), int(frame_seed) & (2 ** 64 - 1), int(frames), int(variant), C.byref(r), C.byref(n), C.byref(bad)))
        return r.value, int(n.value), int(bad.value)

    def probe_gather(self, hot_records=0, iters=2000):
        """Records/s of divergent, dependent 64-byte fetches from the resident wide array (bench.py roofline.gather)."""
        v = C.c_double()
        _check(lib().dr_context_probe_gather(self._h, int(hot_records), int(iters), C.byref(v)))
        return v.value

    # ---- known-answer hooks (tests)
    def kat_rng(self, seed, n):
        out = np.zeros(n, dtype=np.float64)
        _check(lib().dr_kat_rng(self._h, seed, n, _p(out)))
        return out

    def kat_aabb(self, o, d, mn, mx):
        o, d, mn, mx = map(_f32, (o, d, mn, mx))
        n = o.shape[0]
        hit = np.zeros(n, dtype=np.int32)
        dist = np.zeros(n, dtype=np.float32)
        _check(lib().dr_kat_aabb(self._h, n, _p(o), _p(d), _p(mn), _p(mx), _p(hit), _p(dist)))
        return hit, dist

    def kat_tri(self, o, d, v0, v1, v2):
        o, d, v0, v1, v2 = map(_f32, (o, d, v0, v1, v2))
        n = o.shape[0]
        t = np.zeros(n, dtype=np.float32)
        _check(lib().dr_kat_tri(self._h, n, _p(o), _p(d), _p(v0), _p(v1), _p(v2), _p(t)))
        return t

    def kat_node_planes(self, w, a, b):
        """(t_mix, t_cvt), 4 per word: the node test's plane arithmetic through v_fma_mix_f32 / through a conversion and an fma"""
        w = np.ascontiguousarray(w, np.uint32); a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        n = len(w)
        t1 = np.empty(n * 4, np.float32); t2 = np.empty(n * 4, np.float32)
        _check(lib().dr_kat_node_planes(self._h, n, _p(w), _p(a), _p(b), _p(t1), _p(t2)))
        return t1, t2

    def kat_sphere(self, o, d, c, r):
        o, d, c, r = map(_f32, (o, d, c, r))
        n = o.shape[0]
        t = np.zeros(n, dtype=np.float32)
        _check(lib().dr_kat_sphere(self._h, n, _p(o), _p(d), _p(c), _p(r), _p(t)))
        return t

    def kat_optics(self, v, nrm, eta):
        v, nrm, eta = map(_f32, (v, nrm, eta))
        n = v.shape[0]
        refl = np.zeros((n, 3), dtype=np.float32)
        refr = np.zeros((n, 3), dtype=np.float32)
        sch = np.zeros(n, dtype=np.float32)
        _check(lib().dr_kat_optics(self._h, n, _p(v), _p(nrm), _p(eta), _p(refl), _p(refr), _p(sch)))
        return refl, refr, sch

    def kat_normal(self, obj_index, o, d, t):
        """getnormal K:703-773: (normal[n, 3] before the facing flip, texco[n, 3]) for hits of rays (o, d) at t on objects obj_index."""
        idx = np.ascontiguousarray(obj_index, dtype=np.int32)
        o, d, t = _f32(o), _f32(d), _f32(t)
        n = idx.shape[0]
        nrm = np.zeros((n, 3), dtype=np.float32)
        tc = np.zeros((n, 3), dtype=np.float32)
        _check(lib().dr_kat_normal(self._h, n, _p(idx), _p(o), _p(d), _p(t), _p(nrm), _p(tc)))
        return nrm, tc

    def kat_hit(self, o, d, want_visits=False):
        o, d = _f32(o), _f32(d)
        n = o.shape[0]
        t = np.zeros(n, dtype=np.float32)
        idx = np.zeros(n, dtype=np.int32)
        vis = np.zeros(n, dtype=np.int32) if want_visits else None
        _check(lib().dr_kat_hit(self._h, n, _p(o), _p(d), _p(t), _p(idx), _p(vis) if want_visits else None))
        return (t, idx, vis) if want_visits else (t, idx)


class Group:
    """dr_group: one process, one context + one host thread per GPU, stripes gathered to rank 0 (RCCL, or peer copies when
    ranks share a device).  devices: list of ordinals, e.g. [0, 1, 2, 3] -- or [0, 0, 0] to rehearse on one GPU."""

    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*devices)
        h = _VP()
        _check(lib().dr_group_create(len(devices), devs, C.byref(h)))
        self._h = h
        self.size = len(devices)

    def close(self):
        if self._h:
            lib().dr_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def uses_rccl(self):
        return bool(lib().dr_group_uses_rccl(self._h))

    @property
    def rccl_ranks(self):
        """Ranks of the RCCL communicator (ncclCommCount); 0 with the copy transport."""
        return int(lib().dr_group_rccl_ranks(self._h))

    def context(self, rank):
        """Borrowed Context of one rank (owned by the group: do not close it)."""
        c = Context.__new__(Context)
        c._h = _VP(lib().dr_group_context(self._h, rank))
        c.device = None
        c.close = lambda: None
        return c

    def upload(self, scene):
        _check(lib().dr_group_upload_scene(self._h, scene._h))
        return self

    def accum_reset(self, W, H):
        _check(lib().dr_group_accum_reset(self._h, W, H))
        self._acc_shape = (W, H, 3)

    def render_accumulate(self, settings13, W, H, background, frame_seed, seed_stride, nframes, gather_every=0):
        st = _f32(settings13)
        _check(lib().dr_group_render_accumulate(self._h, _p(st), W, H, float(background), int(frame_seed) & (2 ** 64 - 1),
                                                int(seed_stride) & (2 ** 64 - 1), nframes, gather_every))

    def accum_read(self):
        c = self.context(0)
        c._acc_shape = self._acc_shape
        return c.accum_read()


class ProgressiveRenderer:
    """The present loop's render schedule (kernel.cu K:2154-2224), headless.

    iter 0..3: preview ladder at 1/8, 1/4, 1/2, 1/1 resolution into `outr` (iter 0 with the
    file's spp/depth, iter 1..3 with spp 1 / depth 2); iter >= 4: full-resolution frames with the
    file's spp/depth added to `outr`.  The displayed value is clamp(outr / (iter - pnum), 0, 255)
    with integer division (K:2287).  frame k uses seed seed_base + k * seed_stride.
    """

    LADDER = (8, 4, 2, 1)

    def __init__(self, ctx, scene_settings, seed_base=1, seed_stride=1000003):
        self.ctx = ctx
        self.s = scene_settings
        self.W, self.H = scene_settings.width, scene_settings.height
        self.seed_base, self.seed_stride = seed_base, seed_stride
        self.iter = 0
        self.frames_rendered = 0
        self.ctx.accum_reset(self.W, self.H)

    def _seed(self):
        return self.seed_base + self.frames_rendered * self.seed_stride

    def step(self):
        """One present-loop iteration.  Returns (td, divide_by) for display."""
        s, c = self.s, self.ctx
        if self.iter < 4:
            div = self.LADDER[self.iter]
            spp, depth = (s.spp, s.max_depth) if self.iter == 0 else (1, 2)
            st = pack_settings13(s, div, spp=spp, depth=depth)
            c.accum_reset(self.W, self.H)           # CudaStarter overwrites outr on these calls
            c.render_accumulate(st, self.W, self.H, s.background, self._seed(), 0, 1)
            pnum = self.iter
            td = div
        else:
            st = pack_settings13(s, 1)
            c.render_accumulate(st, self.W, self.H, s.background, self._seed(), 0, 1)
            pnum = 3
            td = 1
        self.frames_rendered += 1
        self.iter += 1
        return td, self.iter - pnum

    def image(self, divide_by):
        return self.ctx.accum_present(divide_by)

    def run_pipelined(self, nframes, on_image=None, in_flight=None):
        """The accumulating part of the loop (iter >= 4), `nframes` frames, pipelined (dr_pipeline_submit / dr_pipeline_wait): up to `in_flight` frames
        (default: two groups, 2 x option pipe_group) are queued before the oldest one's image is waited for.  on_image(iter, divide_by, rgb[H, W, 3])
        gets every displayed image, each exactly clamp(sum of the frames so far / divide_by, 0, 255)."""
        assert self.iter >= 4, "run the preview ladder (four step() calls) first"
        s, c = self.s, self.ctx
        c._acc_shape = (self.W, self.H, 3)
        st = pack_settings13(s, 1)
        if in_flight is None:
            in_flight = 2 * max(1, c.get_option("pipe_group"))
        pending = []
        for k in range(nframes):
            self.iter += 1
            div = self.iter - 3
            pending.append((c.pipeline_submit(st, self.W, self.H, s.background, self._seed(), div), self.iter, div))
            self.frames_rendered += 1
            if len(pending) >= in_flight:
                t, it, dv = pending.pop(0)
                img = c.pipeline_wait(t, want_image=True)
                if on_image:
                    on_image(it, dv, img)
        for t, it, dv in pending:
            img = c.pipeline_wait(t, want_image=True)
            if on_image:
                on_image(it, dv, img)
