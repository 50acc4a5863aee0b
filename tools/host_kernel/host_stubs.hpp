// What dogeray_amd/csrc/device_core.hpp needs of <hip/hip_runtime.h> when it is compiled for the HOST (-DDR_HOST_BUILD, clang++ -x c++):
// the function qualifiers, float4, the bit casts and an atomicAdd.  Test / bench infrastructure (tools/host_kernel.cpp); the product
// library never sees this file and has no CPU path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))

struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
// frames of one batch may finish a pixel concurrently on the device; the host build renders one frame per call, every pixel by one thread
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
