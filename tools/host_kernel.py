"""ctypes binding of tools/libhostkernel.so: the product's device functions (dogeray_amd/csrc/device_core.hpp) compiled for the host
(tools/host_kernel.cpp).  Test and bench infrastructure -- never part of the product library."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SO = os.path.join(_HERE, "libhostkernel.so")
_CSRC = os.path.join(_ROOT, "dogeray_amd", "csrc")
_SOURCES = [os.path.join(_HERE, "host_kernel.cpp")] + [os.path.join(_CSRC, f) for f in ("rts_reader.cpp", "bvh_builder.cpp", "linearise.cpp", "wide_builder.cpp", "capi_host.cpp")]
_lib = None


def build(force=False):
    deps = _SOURCES + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".h", ".hpp"))] + [os.path.join(_HERE, "host_kernel", "host_stubs.hpp")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(d) <= os.path.getmtime(_SO) for d in deps):
        return _SO
    cxx = "/opt/rocm/lib/llvm/bin/clang++"          # ext_vector_type (u32x4) is a clang extension
    if not os.path.exists(cxx):
        cxx = "clang++"
    # -ffp-contract=off as on the device; -mfma so that the explicit fmaf of the folded node test is one instruction
    cmd = [cxx, "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-pthread", "-DDR_HOST_BUILD=1",
           "-I" + os.path.join(_HERE, "host_kernel"), "-o", _SO] + _SOURCES
    subprocess.check_call(cmd)
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.hk_last_error.restype = C.c_char_p
        L.hk_scene_load.restype = C.c_void_p
        L.hk_scene_load.argtypes = [C.c_char_p, C.c_char_p]
        L.hk_scene_free.argtypes = [C.c_void_p]
        L.hk_has_wide.argtypes = [C.c_void_p]
        L.hk_wide_depth.argtypes = [C.c_void_p]
        L.hk_check_uniform.restype = C.c_longlong
        L.hk_check_uniform.argtypes = [C.c_longlong, C.c_ulonglong]
        L.hk_ray_margin.restype = None
        L.hk_ray_margin.argtypes = [C.c_longlong, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.hk_tri_hit.restype = None
        L.hk_tri_hit.argtypes = [C.c_longlong] + [C.c_void_p] * 6
        L.hk_check_reject.restype = C.c_longlong
        L.hk_check_reject.argtypes = [C.c_longlong, C.c_ulonglong, C.POINTER(C.c_double)]
        L.hk_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class Scene:
    def __init__(self, rts_path, texdir=""):
        self.h = lib().hk_scene_load(os.fsencode(rts_path), os.fsencode(texdir or ""))
        if not self.h:
            raise RuntimeError(lib().hk_last_error().decode())

    def __del__(self):
        try:
            if self.h:
                lib().hk_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def has_wide(self):
        return bool(lib().hk_has_wide(self.h))

    def wide_depth(self):
        return int(lib().hk_wide_depth(self.h))

    def render(self, settings13, W, H, background, frame_seed, traversal=2, nthreads=1, col_mod=1, col_rem=0, count=True):
        """One frame: (int32[W, H, 3] indexed [x, y], counters dict or None)."""
        st = np.ascontiguousarray(settings13, dtype=np.float32)
        out = np.zeros((W, H, 3), dtype=np.int32)
        ctr = (C.c_uint64 * 6)()
        rc = lib().hk_render(self.h, st.ctypes.data, W, H, float(background), int(frame_seed) & (2 ** 64 - 1), traversal, nthreads, col_mod, col_rem,
                             out.ctypes.data, C.cast(ctr, C.c_void_p) if count else None)
        if rc != 0:
            raise RuntimeError(lib().hk_last_error().decode())
        return out, (dict(zip(("rays", "V", "L", "S", "T", "samples"), [int(v) for v in ctr])) if count else None)
