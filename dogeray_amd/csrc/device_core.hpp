// Device functions of the path-tracing megakernel (gfx950).
//
// Each function states which reference function it replaces (kernel.cu, K:<line>).  The
// arithmetic (operation order, float/double promotion, comparison direction) follows the
// reference expression by expression, because a one-ulp difference flips hit/miss branches
// and rejection loops; what changes is where operands come from (SoA records, per-ray
// precomputation) and how lanes are scheduled.  Build with -ffp-contract=off: no FMA
// contraction, IEEE division and square root.
#pragma once
#ifdef DR_HOST_BUILD            // tools/host_kernel.cpp: the same arithmetic compiled for the host (a CPU baseline and a CPU-side check; never in the product library)
#include "host_stubs.hpp"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "device_layout.h"

// tools/instr_budget.py builds the render kernels with -DDR_ISA_MARKS=1 -save-temps and counts the instructions between these comments
#ifndef DR_ISA_MARKS
#define DR_ISA_MARKS 0
#endif
#if DR_ISA_MARKS
#define DR_MARK(s) asm volatile("; DRMARK " s)
#else
#define DR_MARK(s) do {} while (0)
#endif

namespace dr {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float a, float b, float c) { V3 r; r.x = a; r.y = b; r.z = c; return r; }
__device__ __forceinline__ V3 splat(float a) { return mk(a, a, a); }
__device__ __forceinline__ V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator/(V3 a, V3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float length(V3 a) { return __builtin_sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 normalized(V3 v) {              // getNormalizedVec K:179
  float inv = 1.0f / __builtin_sqrtf(dot(v, v));
  return mk(v.x * inv, v.y * inv, v.z * inv);
}

// float -> int as CUDA's cvt.rzi.s32.f32: saturating, NaN -> 0 (K:802,1083-1085)
__device__ __forceinline__ int f2i(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
}

// ------------------------------------------------------------------ RNG: cuRAND XORWOW
// curand_init(seed, 0, 0) / curand() / curand_uniform_double() (call sites K:644,657,1065-1068)
__device__ __forceinline__ double bits_double(uint32_t hi, uint32_t lo) {      // the double with these two words
#ifndef DR_HOST_BUILD
  // (the high word is a constant: set where it is used -- hoisted out of the kernel's main loop, the two constants of Xorwow::z_plus_half would
  // occupy two registers through every node and leaf step, where there are none to spare)
  asm volatile("v_mov_b32 %0, %1" : "=v"(hi) : "s"(hi));
#endif
  const uint64_t b = ((uint64_t)hi << 32) | lo;
  double r; __builtin_memcpy(&r, &b, 8);
  return r;
}
struct Xorwow {
  uint32_t v0, v1, v2, v3, v4, d;
  __device__ __forceinline__ void init(uint64_t seed) {
    uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    d = 6615241u + t1 + t0;
    v0 = 123456789u + t0;
    v1 = 362436069u ^ t0;
    v2 = 521288629u + t1;
    v3 = 88675123u ^ t1;
    v4 = 5783321u + t0;
  }
  __device__ __forceinline__ uint32_t next() {
    uint32_t t = v0 ^ (v0 >> 2);
    v0 = v1; v1 = v2; v2 = v3; v3 = v4;
#ifndef DR_HOST_BUILD
    // t << 1 as t + t and the three-way xor as ONE v_bitop3_b32: both at the fast class's 2.4 cycles; spelled in plain C++ the compiler turns t + t back
    // into a 4.2-cycle shift and emits three v_xor (7 instead of 8 instructions per output, -0.4 % of the frame: profiles/r4_m_rejection_loop.txt)
    uint32_t t2; asm("v_add_u32 %0, %1, %1" : "=v"(t2) : "v"(t));
    v4 = __builtin_amdgcn_bitop3_b32(t, t2, v4 << 4, 0x96) ^ v4;
#else
    v4 = (v4 ^ (v4 << 4)) ^ (t ^ (t << 1));
#endif
    d += 362437u;
    return v4 + d;
  }
  // curand_uniform_double(): z = x ^ (y << 21) (53 bits), (double)z * 2^-53 + 2^-54.  The conversion and the arithmetic behind it are restated
  // with fewer operations and THE SAME roundings (tools/host_kernel.cpp hk_check_uniform compares them with the plain expressions):
  //  * (double)z = ((2^84 + hi * 2^32) - (2^84 + 2^52)) + (2^52 + lo), hi = z >> 32, lo = the low word: both terms are doubles whose high word is a
  //    constant and whose low word is hi / lo, and both additions are exact -- two f64 adds instead of two conversions, a scaling and an add
  //    (and no 64-bit shift);
  //  * RN(z * 2^-53 + 2^-54) = RN(z + 0.5) * 2^-53 (a power of two commutes with the rounding);
  //  * (float)(u * 2 - 1) of the unit-sphere draw = (float)fma(RN(z + 0.5), 2^-52, -1): u * 2 is exact, so the subtraction's rounding is the fma's.
  __device__ __forceinline__ double z_plus_half() {               // RN((double)z + 0.5)
    const uint32_t x = next(), y = next();
    const uint32_t lo = x ^ (y << 21), hi = y >> 11;
    const double d_hi = bits_double(0x45300000u, hi), d_lo = bits_double(0x43300000u, lo);
    return ((d_hi - 19342813118337666422669312.0) + d_lo) + 0.5;
  }
  __device__ __forceinline__ double uniform_double() { return z_plus_half() * 0x1p-53; }
  __device__ __forceinline__ float uniform_pm1() { return (float)__builtin_fma(z_plus_half(), 0x1p-52, -1.0); }      // (float)(uniform_double() * 2 - 1)
};

// The rejection test of K:645 / K:992 is pow(length(p), 2.0f) >= 1 with length = sqrtf(dot): l = RN(sqrt(d2)), then RN(l * l) >= 1.  Away from 1 the
// outcome is that of d2 >= 1 itself: for d2 <= 1 - 2^-20, sqrt(d2) <= 1 - 2^-21 (a float), so l <= 1 - 2^-21 and l * l <= 1 - 2^-20 + 2^-42 rounds
// below 1; for d2 >= 1 + 2^-20, sqrt(d2) >= 1 + 2^-21 - 2^-43 rounds to at least 1 + 2^-21 and l * l > 1.  Only inside that band is the correctly
// rounded square root (some fifteen instructions) really taken -- and the wave branches around it when none of its lanes is in the band.
__device__ __forceinline__ bool outside_unit(float d2) {
  const bool band = __builtin_fabsf(d2 - 1.0f) < 0x1p-20f;
  bool out = d2 >= 1;
#ifndef DR_HOST_BUILD
  if (__builtin_amdgcn_ballot_w64(band) != 0ull)                  // (wave-uniform)
#endif
  {
#ifndef DR_HOST_BUILD
    asm volatile("; rare: a lane within 2^-20 of the unit sphere");      // (keeps the compiler from hoisting the square root out of the branch)
#endif
    if (band) {
      const float l = __builtin_sqrtf(d2);
      out = l * l >= 1;                                           // pow(len, 2.0f)
    }
  }
  return out;
}
__device__ __forceinline__ V3 rand_in_unit_sphere(Xorwow& r) {   // K:640-648
  for (;;) {
    float x = r.uniform_pm1();
    float y = r.uniform_pm1();
    float z = r.uniform_pm1();
    V3 p = mk(x, y, z);
    if (outside_unit(dot(p, p))) continue;
    return p;
  }
}
__device__ __forceinline__ float randy(Xorwow& r) { return (float)r.uniform_double(); }   // K:651
__device__ __forceinline__ V3 rand_in_unit_disk(Xorwow& r) {     // K:988-994
  for (;;) {
    float x = randy(r) * 2 - 1;
    float y = randy(r) * 2 - 1;
    V3 p = mk(x, y, 0);
    if (outside_unit(dot(p, p))) continue;
    return p;
  }
}

// One rejection loop for a whole wave: lanes with kind 3 draw a point in the unit sphere (K:640-648: three uniforms per candidate, in double, x y z),
// lanes with kind 2 a point in the unit disk (K:988-994: two uniforms per candidate, in float), lanes with kind 0 nothing -- each lane's sequence of
// draws is exactly that of rand_in_unit_sphere / rand_in_unit_disk, but the wave runs the loop once, for as many turns as its unluckiest lane needs,
// instead of once per kind (the persistent kernel's shade / refill phase: the lanes that scatter and the lanes that start a path).
// (turns: counting builds only -- the number of candidates this lane drew; the wave runs the loop for as many turns as its unluckiest lane)
// A turn of that loop does not convert its candidate exactly.  It classifies it from the top 32 bits of each 53-bit draw -- T = y ^ (x >> 21) are the
// top 32 bits of z = x ^ (y << 21) | (y >> 11) << 32 -- as c~ = RN32(RN32((float)T) * 2^-31 - 1), and rejects it only when the approximate squared length
// says "certainly outside"; whatever is not rejected is converted exactly AFTER the loop and put to the reference's own test (outside_unit), and a lane
// whose candidate fails that test goes back into the loop.  So the sequence of draws and the accepted point are the reference's as long as
//     d2~ >= 1 + 2^-16   implies   outside_unit(d2) for the exactly converted candidate:
//  * with v = 2 (z + 0.5) 2^-53 - 1 the real number both conversions approximate: exact conversion |c - v| <= 3 * 2^-25 (sphere lanes: RN64 then RN32 of
//    fma(Z, 2^-52, -1) <= 2^-25 + 2^-52; disk lanes: RN32(u) * 2 - 1 with one more rounding <= 2^-25 when the result is below -0.5); the approximation
//    |c~ - v| <= 2^-24 ((float)T, half an ulp of 2^32, times 2^-31) + 2^-25 (the fma's rounding) + 2^-31 (the 21 bits dropped) -- so |c~ - c| < 2^-22;
//  * three coordinates of magnitude <= 1: the real sums of squares differ by less than 3 * 2 * 2^-22 < 2^-19; either float evaluation (three products <= 1,
//    two sums < 4) is within 5 * 2^-23 of its real sum; so |d2~ - d2| < 2^-19 + 2 * 5 * 2^-23 < 2^-17;
//  * hence d2~ >= 1 + 2^-16 gives d2 > 1 + 2^-17 > 1 + 2^-20, where outside_unit() is true (its own argument above).
// The accepted candidate's words are not kept through the loop: XORWOW's state IS its last five outputs (v0..v4 = w[n-4..n], output k = w[k] + d[k], d[k] =
// d - (n - k) * 362437), so the last four draws of a disk lane are rebuilt from the state and a sphere lane keeps one word, the first of its six.
// (tools/host_kernel.cpp hk_check_reject compares the loop with rand_in_unit_sphere / rand_in_unit_disk, draw for draw.)
__device__ __forceinline__ float reject_coord(uint32_t x, uint32_t y) {
  return __builtin_fmaf((float)(y ^ (x >> 21)), 0x1p-31f, -1.0f);
}
__device__ __forceinline__ double z_plus_half_of(uint32_t x, uint32_t y) {      // Xorwow::z_plus_half() of the two outputs x, y
  const uint32_t lo = x ^ (y << 21), hi = y >> 11;
  const double d_hi = bits_double(0x45300000u, hi), d_lo = bits_double(0x43300000u, lo);
  return ((d_hi - 19342813118337666422669312.0) + d_lo) + 0.5;
}
template <bool COUNT = false>
__device__ __forceinline__ V3 rand_points_merged(Xorwow& r, int kind, unsigned* turns = nullptr) {
  V3 p = mk(0, 0, 0);
  bool todo = kind != 0;
  uint32_t first = 0;
  for (;;) {
    while (todo) {
      DR_MARK("reject_turn");
      if (COUNT) (*turns)++;
      const uint32_t a = r.next(), b = r.next(), c = r.next(), e = r.next();
      first = a;
      const float x = reject_coord(a, b), y = reject_coord(c, e);
      float d2 = __builtin_fmaf(y, y, x * x);
      if (kind == 3) {
        const uint32_t f = r.next(), g = r.next();
        const float z = reject_coord(f, g);
        d2 = __builtin_fmaf(z, z, d2);
      }
      todo = d2 >= 1.0f + 0x1p-16f;
    }
    DR_MARK("reject_done");
    if (kind == 0) break;
    // the candidate's words, out of the generator's state
    const uint32_t d0 = r.d, d1 = d0 - 362437u, d2w = d1 - 362437u, d3 = d2w - 362437u, d4 = d3 - 362437u;
    const uint32_t o0 = r.v4 + d0, o1 = r.v3 + d1, o2 = r.v2 + d2w, o3 = r.v1 + d3, o4 = r.v0 + d4;
    const bool sphere = kind == 3;
    const double zx = z_plus_half_of(sphere ? first : o3, sphere ? o4 : o2), zy = z_plus_half_of(sphere ? o3 : o1, sphere ? o2 : o0);
    float x, y, z = 0.0f;
    if (sphere) {
      x = (float)__builtin_fma(zx, 0x1p-52, -1.0); y = (float)__builtin_fma(zy, 0x1p-52, -1.0);
      z = (float)__builtin_fma(z_plus_half_of(o1, o0), 0x1p-52, -1.0);
    } else {
      x = (float)(zx * 0x1p-53) * 2 - 1; y = (float)(zy * 0x1p-53) * 2 - 1;      // randy() * 2 - 1
    }
    p = mk(x, y, z);
    todo = outside_unit(dot(p, p));
    if (!todo) break;
  }
  return p;
}

// ------------------------------------------------------------------ intersection
// Which plane of a box a ray meets first depends on the sign of 1/direction only (aabb2's swap, K:262-266).  SignCmp decides by comparing, as
// the reference does.  SignMask (persistent kernel) keeps the outcome of those three comparisons as per-lane words, all ones or all zeros,
// rebuilt whenever the lane's ray changes: a select is then (if_pos & ~m) | (if_neg & m), ONE v_bitop3_b32 -- which issues in 2.4 SIMD cycles
// where the compare and the v_cndmask take 4.2 each (tools/valu_rate.hip): 14 instead of 38 cycles per node step and per leaf step.
struct SignCmp {
  V3 inv;
  __device__ __forceinline__ unsigned sel(int axis, unsigned if_pos, unsigned if_neg) const { return (axis == 0 ? inv.x : axis == 1 ? inv.y : inv.z) < 0.0f ? if_neg : if_pos; }
};
struct SignMask {
  unsigned x, y, z;
  __device__ __forceinline__ unsigned sel(int axis, unsigned if_pos, unsigned if_neg) const {
    const unsigned m = axis == 0 ? x : axis == 1 ? y : z;
#ifdef DR_HOST_BUILD
    return (if_pos & ~m) | (if_neg & m);
#else
    return __builtin_amdgcn_bitop3_b32(if_pos, if_neg, m, 0xd8);      // m ? if_neg : if_pos, bit by bit (written out, the compiler splits the expression into three operations)
#endif
  }
};
__device__ __forceinline__ SignMask sign_mask(V3 inv) {
  SignMask g; g.x = inv.x < 0.0f ? 0xffffffffu : 0u; g.y = inv.y < 0.0f ? 0xffffffffu : 0u; g.z = inv.z < 0.0f ? 0xffffffffu : 0u;
  return g;
}
// aabb2 K:244-274 with the reciprocal direction hoisted out of the node loop (same divide,
// same operands) and the per-axis early return folded into one final comparison: t_min only
// grows, t_max only shrinks and neither can become NaN, so "t_max <= t_min after some axis"
// and "t_max <= t_min after the last axis" are the same predicate.
template <class Sign>
__device__ __forceinline__ bool slab_sel(V3 o, V3 inv, const Sign& sg, const float mn[3], const float mx[3], float& dist) {      // slab() with the planes chosen by sg
  auto pf = [&sg](int axis, float if_pos, float if_neg) { return __uint_as_float(sg.sel(axis, __float_as_uint(if_pos), __float_as_uint(if_neg))); };
  float nx = pf(0, mn[0], mx[0]), fx = pf(0, mx[0], mn[0]);
  float ny = pf(1, mn[1], mx[1]), fy = pf(1, mx[1], mn[1]);
  float nz = pf(2, mn[2], mx[2]), fz = pf(2, mx[2], mn[2]);
  float t0x = (nx - o.x) * inv.x, t1x = (fx - o.x) * inv.x;
  float t0y = (ny - o.y) * inv.y, t1y = (fy - o.y) * inv.y;
  float t0z = (nz - o.z) * inv.z, t1z = (fz - o.z) * inv.z;
  float t_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(t0x, 0.0f), t0y), t0z);
  float t_max = __builtin_fminf(__builtin_fminf(__builtin_fminf(t1x, 10000.0f), t1y), t1z);
  dist = t_min;
  return t_max > t_min;
}
__device__ __forceinline__ bool slab(V3 o, V3 inv, const float mn[3], const float mx[3], float& dist) {
  // The reference swaps (t0, t1) when invD < 0, i.e. the plane entered first is max for a
  // negative direction: select the planes first (same products afterwards).  "x > m ? x : m" with
  // m never NaN is fmax(x, m) (a NaN x is ignored either way; the sign of a zero result feeds
  // comparisons only), so the three-axis chain folds into max3 / min3.
  float nx = inv.x < 0.0f ? mx[0] : mn[0], fx = inv.x < 0.0f ? mn[0] : mx[0];
  float ny = inv.y < 0.0f ? mx[1] : mn[1], fy = inv.y < 0.0f ? mn[1] : mx[1];
  float nz = inv.z < 0.0f ? mx[2] : mn[2], fz = inv.z < 0.0f ? mn[2] : mx[2];
  float t0x = (nx - o.x) * inv.x, t1x = (fx - o.x) * inv.x;
  float t0y = (ny - o.y) * inv.y, t1y = (fy - o.y) * inv.y;
  float t0z = (nz - o.z) * inv.z, t1z = (fz - o.z) * inv.z;
  float t_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(t0x, 0.0f), t0y), t0z);
  float t_max = __builtin_fminf(__builtin_fminf(__builtin_fminf(t1x, 10000.0f), t1y), t1z);
  dist = t_min;
  return t_max > t_min;
}

// hit_tri K:277-313 on {v0, e1 = v1 - v0, e2 = v2 - v0}; returns t or -1.
// 1.0/a is a double division narrowed to float in the reference; narrowing a correctly
// rounded double quotient of two floats equals the correctly rounded float quotient.
__device__ __forceinline__ float tri_hit(V3 ro, V3 rd, V3 v0, V3 e1, V3 e2) {
  const float EPS = 0.0001f;
  V3 h = cross(rd, e2);
  float a = dot(e1, h);
  if (a > -EPS && a < EPS) return -1.0f;
  float f = 1.0f / a;
  V3 s = ro - v0;
  float u = f * dot(s, h);
  if (u < 0.0f || u > 1.0f) return -1.0f;
  V3 q = cross(s, e1);
  float v = f * dot(rd, q);
  if (v < 0.0f || u + v > 1.0f) return -1.0f;
  float t = f * dot(e2, q);
  return t > EPS ? t : -1.0f;
}

// hit_sphere K:316-333
__device__ __forceinline__ float sphere_hit(V3 c, float radius, V3 o, V3 d) {
  V3 oc = o - c;
  float ld = length(d);
  float a = ld * ld;
  float half_b = dot(oc, d);
  float lo = length(oc);
  float cc = lo * lo - radius * radius;
  float disc = half_b * half_b - a * cc;
  if (disc < 0) return -1.0f;
  return (-half_b - __builtin_sqrtf(disc)) / a;
}

// singlehit K:432-464 on one primitive record (three 16-byte units); returns t or -1
__device__ __forceinline__ float prim_hit_regs(float4 A, float4 B, float4 C, V3 o, V3 d) {
  int type = __float_as_int(C.y);
  float dist = -1.0f;
  if (type == 2) dist = tri_hit(o, d, mk(A.x, A.y, A.z), mk(A.w, B.x, B.y), mk(B.z, B.w, C.x));
  else if (type == 0) dist = sphere_hit(mk(A.x, A.y, A.z), A.w, o, d);
  if (dist < 10000.0f && dist > -0.0f) return dist;          // K:449
  return -1.0f;
}
// the same on the walk array's leaf record: kind (WALK_KIND_*), v0 / centre, e1 (e1.x = radius), e2
__device__ __forceinline__ float prim_hit_kind(int kind, V3 v0, V3 e1, V3 e2, V3 o, V3 d) {
  float dist = -1.0f;
  if (kind == WALK_KIND_TRIANGLE) dist = tri_hit(o, d, v0, e1, e2);
  else if (kind == WALK_KIND_SPHERE) dist = sphere_hit(v0, e1.x, o, d);
  if (dist < 10000.0f && dist > -0.0f) return dist;          // K:449
  return -1.0f;
}
// successor of the leaf record that link `node` (bit 0 set) points to: the next record (4 units on), or the end
// of the walk.  The walk array ends with a terminator record (an internal node with an empty box and both links
// -1), so the plain build needs no end test: the last leaf's successor is that record (two instructions instead
// of eight on the hot path).  The counting build stops at the last leaf itself so that its visit counter stays
// the reference's.
template <bool COUNT>
__device__ __forceinline__ int leaf_successor(int node, int info) {
  const int next = node + (2 * WALK_UNITS_LEAF - 1) + ((info >> 28) & 1);
  if (COUNT) return (info >> 29) & 1 ? -1 : next;
  return next;
}
__device__ __forceinline__ float prim_hit(const DevPrim* __restrict__ prims, int slot, V3 o, V3 d) {
  const float4* p = reinterpret_cast<const float4*>(prims + slot);
  return prim_hit_regs(p[0], p[1], p[2], o, d);
}

struct Hit { float t; int slot; };

struct Ctr { unsigned rays, V, L, S, T, samples, trav_slots, ray_slots; };   // *_slots: 64 per wave-level iteration / query (SIMD efficiency probes)

// hit() K:468-512 in the reference's order, over the walk array (device_layout.h): a node that
// is entered continues with its hit link (its first child), a node that is skipped, and every
// leaf, with its miss link.  The link says whether its target is a leaf, so a leaf's box and
// primitive are fetched together and the triangle test costs no second memory round trip.
// The loop body is its own function so that the persistent kernel can interleave node steps of
// different rays (lanes re-armed while their neighbours are still walking).
struct Trav { int node; float best_t; int best_slot; };   // node: link to visit next, < 0 = walk finished

// "No hit yet" is t = 10000: singlehit only returns distances below 10000 (K:449) and aabb2's exit distance starts at 10000 (K:246), so every
// comparison against the running best (box entry < best, t < best) comes out as with the 1e7 of hit() K:470 -- and the wide walk's node
// test needs no min(best, 10000) of its own.
__device__ __forceinline__ void trav_begin(Trav& tr) { tr.node = 0; tr.best_t = 10000.0f; tr.best_slot = -1; }

// The walk array is read through a buffer descriptor with 128-bit buffer loads: the compiler may
// not re-slice those into narrower / unaligned pieces (it does so with plain float4 loads: the box
// came in as dwordx2 + unaligned dwordx4 + dwordx3), so a node costs exactly two 16-byte requests
// and a leaf four, all issued before the first wait.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifdef DR_HOST_BUILD
struct WalkRsrc { const unsigned char* base; unsigned bytes; };      // a buffer descriptor's base and range; loads past the range return 0, as the hardware's do
__device__ __forceinline__ WalkRsrc walk_rsrc(const RenderParams& P) { WalkRsrc r; r.base = (const unsigned char*)P.walk; r.bytes = P.walk_bytes; return r; }
__device__ __forceinline__ WalkRsrc wide_rsrc(const RenderParams& P) { WalkRsrc r; r.base = (const unsigned char*)P.wide; r.bytes = P.wide_bytes; return r; }
__device__ __forceinline__ u32x4 host_buffer_load_b128(WalkRsrc r, unsigned byte_off) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if ((unsigned long long)byte_off + 16ull <= (unsigned long long)r.bytes) memcpy(&v, r.base + byte_off, 16);
  return v;
}
__device__ __forceinline__ float4 ld_unit(WalkRsrc r, unsigned byte_off) {
  u32x4 v = host_buffer_load_b128(r, byte_off);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
#else
typedef __amdgpu_buffer_rsrc_t WalkRsrc;
__device__ __forceinline__ WalkRsrc walk_rsrc(const RenderParams& P) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)P.walk, 0, (int)P.walk_bytes, 0x00020000);
}
__device__ __forceinline__ WalkRsrc wide_rsrc(const RenderParams& P) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)P.wide, 0, (int)P.wide_bytes, 0x00020000);
}
__device__ __forceinline__ float4 ld_unit(WalkRsrc r, unsigned byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
#endif

template <bool COUNT>
__device__ __forceinline__ void trav_step(WalkRsrc walk, V3 o, V3 d, V3 inv, Trav& tr, Ctr& c) {
  const bool leaf = tr.node & 1;
  const unsigned off = (unsigned)(tr.node >> 1) << 4;
  float4 A = ld_unit(walk, off), B = ld_unit(walk, off + 16);
  float4 C = A, D = A;
  if (leaf) { C = ld_unit(walk, off + 32); D = ld_unit(walk, off + 48); }
  float mn[3] = {A.x, A.y, A.z}, mx[3] = {B.x, B.y, B.z};
  const int w0 = __float_as_int(A.w);
  const int next_miss = leaf ? leaf_successor<COUNT>(tr.node, w0) : __float_as_int(B.w);
  float dist;
  if (COUNT) c.V++;
  bool h = slab(o, inv, mn, mx, dist);
  if (h && dist < tr.best_t) {
    if (leaf) {
      if (COUNT) c.L++;
      float t = prim_hit_kind((w0 >> WALK_SLOT_BITS) & 3, mk(B.w, C.x, C.y), mk(C.z, C.w, D.x), mk(D.y, D.z, D.w), o, d);
      if (t > -0.01f && t < tr.best_t) { tr.best_t = t; tr.best_slot = w0 & ((1 << WALK_SLOT_BITS) - 1); }   // K:488 (t is -1 or > 0)
      tr.node = next_miss;
    } else {
      tr.node = w0;
    }
  } else {
    tr.node = next_miss;
  }
}

// Same step, but a leaf whose box passes is not tested on the spot: the lane keeps the primitive
// it has just fetched and waits ("parks") until enough lanes of the wave have one, so the
// ~90-instruction triangle test runs once for many lanes instead of on nearly every iteration for
// one or two.  The per-lane sequence of tests and updates is unchanged.
// C and D stay 128-bit register tuples from the load to the test (as scalars the allocator scattered them and
// copied all eight back and forth on every step).
struct ParkedLeaf { float v0x; u32x4 C, D; int info; bool parked; };
#ifdef DR_HOST_BUILD
__device__ __forceinline__ u32x4 ld_unit_raw(WalkRsrc r, unsigned byte_off) { return host_buffer_load_b128(r, byte_off); }
#else
__device__ __forceinline__ u32x4 ld_unit_raw(WalkRsrc r, unsigned byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0); }
#endif

template <bool COUNT>
__device__ __forceinline__ void trav_step_park(WalkRsrc walk, V3 o, V3 inv, Trav& tr, ParkedLeaf& pk, Ctr& c) {
  const bool leaf = tr.node & 1;
  const unsigned off = (unsigned)(tr.node >> 1) << 4;
  float4 A = ld_unit(walk, off), B = ld_unit(walk, off + 16);
  // a stepping lane is not parked, so its parked record is free: the leaf's primitive lands there directly
  if (leaf) { pk.C = ld_unit_raw(walk, off + 32); pk.D = ld_unit_raw(walk, off + 48); }
  float mn[3] = {A.x, A.y, A.z}, mx[3] = {B.x, B.y, B.z};
  const int w0 = __float_as_int(A.w);
  // selects, not branches: both sides are two or three instructions, a divergent branch costs more than that
  const int succ = leaf_successor<COUNT>(tr.node, w0);
  const int next_miss = leaf ? succ : __float_as_int(B.w);
  float dist;
  if (COUNT) c.V++;
  bool h = slab(o, inv, mn, mx, dist) && dist < tr.best_t;
  pk.info = leaf ? w0 : pk.info;
  pk.v0x = leaf ? B.w : pk.v0x;
  pk.parked = h && leaf;
  tr.node = (h && !leaf) ? w0 : next_miss;
}
template <bool COUNT>
__device__ __forceinline__ void parked_test(V3 o, V3 d, Trav& tr, ParkedLeaf& pk, Ctr& c) {
  if (COUNT) c.L++;
  auto f = [](unsigned v) { return __uint_as_float(v); };
  float t = prim_hit_kind((pk.info >> WALK_SLOT_BITS) & 3, mk(pk.v0x, f(pk.C.x), f(pk.C.y)), mk(f(pk.C.z), f(pk.C.w), f(pk.D.x)), mk(f(pk.D.y), f(pk.D.z), f(pk.D.w)), o, d);
  if (t > -0.01f && t < tr.best_t) { tr.best_t = t; tr.best_slot = pk.info & ((1 << WALK_SLOT_BITS) - 1); }   // K:488
  pk.parked = false;
}

// Cooperative closest hit: all 64 lanes of a wave answer ONE query.  The wave keeps a stack of internal
// nodes (pair records) in LDS; per round every lane pops one, tests its two child boxes against the wave's
// best t so far, tests leaf children on the spot and pushes internal children that pass.  A ray that costs
// one lane a thousand dependent steps is finished in a few dozen rounds.  What comes out is the
// lexicographic minimum (t, slot) over every leaf whose box and ancestors' boxes the ray enters nearer
// than the best t -- the same hit as hit() K:468-512, on the same condition as the ordered traversal
// (equal t goes to the lower slot = the leaf the reference's walk reaches first).  Used only while a
// launch drains (persistent kernel): idle lanes shorten the few long rays that set the launch time.
constexpr int WAVE_LDS_DWORDS = 24 * 64;  // per-wave LDS region of the persistent kernel (6 KiB): phase stash / cooperative stack
constexpr int WIDE_STASH = 10;             // phase stash of the WIDE persistent kernel, dwords per lane (behind the WIDE_STACK stack words)
constexpr int COOP_STACK = WAVE_LDS_DWORDS;   // node stack entries; a deeper frontier falls back to the plain walk

#ifndef DR_HOST_BUILD          // wave-level code: not part of the host build
__device__ __forceinline__ float wave_min_f32(float v) {
  for (int off = 32; off > 0; off >>= 1) v = __builtin_fminf(v, __shfl_xor(v, off, 64));
  return v;
}

// returns false (and no result) if the stack would overflow; the caller then keeps walking the plain way
__device__ __forceinline__ bool coop_closest_hit(const DevPair* __restrict__ pairs, const DevPrim* __restrict__ prims,
                                                 V3 o, V3 d, V3 inv, float bound_t, int bound_slot,
                                                 int* __restrict__ stack, Hit& out) {
  const int lane = (int)__lane_id();
  float best_t = bound_t;
  int best_slot = bound_slot < 0 ? 0x7fffffff : bound_slot;
  float prune_t = bound_t;
  int n = 1;
  if (lane == 0) stack[0] = 0;               // the root's pair
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  bool ok = true;
  while (n > 0) {
    const int take = n < 64 ? n : 64;
    int pair = -1;
    if (lane < take) pair = stack[n - 1 - lane];
    n -= take;
    bool push0 = false, push1 = false;
    int c0 = 0, c1 = 0;
    if (pair >= 0) {
      const float4* pp = reinterpret_cast<const float4*>(pairs + pair);
      float4 A = pp[0], B = pp[1], C = pp[2], D = pp[3];
      c0 = __float_as_int(A.w); c1 = __float_as_int(B.w);
      float mn0[3] = {A.x, A.y, A.z}, mx0[3] = {B.x, B.y, B.z};
      float mn1[3] = {C.x, C.y, C.z}, mx1[3] = {D.x, D.y, D.z};
      float d0, d1;
      const bool h0 = slab(o, inv, mn0, mx0, d0) && d0 <= prune_t;   // <=: a box entered exactly at the best t may hold a tie with a lower slot
      const bool h1 = slab(o, inv, mn1, mx1, d1) && d1 <= prune_t;
      if (h0 && c0 < 0) {
        const int slot = ~c0;
        float t = prim_hit(prims, slot, o, d);
        if (t > 0.0f && (t < best_t || (t == best_t && slot < best_slot))) { best_t = t; best_slot = slot; }
      }
      if (h1 && c1 < 0) {
        const int slot = ~c1;
        float t = prim_hit(prims, slot, o, d);
        if (t > 0.0f && (t < best_t || (t == best_t && slot < best_slot))) { best_t = t; best_slot = slot; }
      }
      push0 = h0 && c0 >= 0;
      push1 = h1 && c1 >= 0;
    }
    const unsigned long long m0 = __ballot(push0), m1 = __ballot(push1);
    const int n0 = __popcll(m0), n1 = __popcll(m1);
    if (n + n0 + n1 > COOP_STACK) { ok = false; break; }
    __builtin_amdgcn_wave_barrier();            // every lane has read its entry before anyone overwrites it
    if (push0) stack[n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u))] = c0;
    if (push1) stack[n + n0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u))] = c1;
    n += n0 + n1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    prune_t = wave_min_f32(best_t);
  }
  if (!ok) return false;
  // lexicographic minimum (t, slot) over the lanes
  for (int off = 32; off > 0; off >>= 1) {
    const float ot = __shfl_xor(best_t, off, 64);
    const int os = __shfl_xor(best_slot, off, 64);
    const bool take_other = ot < best_t || (ot == best_t && os < best_slot);
    best_t = take_other ? ot : best_t;
    best_slot = take_other ? os : best_slot;
  }
  out.t = best_t;
  out.slot = best_slot == 0x7fffffff ? -1 : best_slot;
  return true;
}
#endif

// ------------------------------------------------------------------ wide walk
// hit() K:468-512 over the 4-way tree of wide_builder.cpp (device_layout.h "wide walk").  The answer is the
// lexicographic minimum (t, slot) over the leaves whose own (exact, reference) box the ray enters no farther than
// the best t -- the reference's hit, by the argument at closest_hit_ordered below: internal boxes only have to
// ENCLOSE the leaves under them (the host checks every decoded plane with this very fmaf), and the slab test is
// monotone in the box, so a subtree is skipped only if every leaf in it would fail its own test.
//
// Per lane: `node` = record index << 1 | is-leaf; the children of a node that were entered but not yet visited
// are one word (first child's index << 8 | leaf mask << 4 | pending mask); the newest such word lives in a
// register (`top`), older ones on a per-lane stack in LDS (word k of lane l at stack[k * 64 + l]: conflict-free).
struct WideStack { unsigned top; int sp; int sb; };   // LDS words [sb, sp): sb > 0 once the oldest words have been handed to helper lanes

// Tests the four children of a node; returns the mask of those the ray may enter no farther than best_t, and the key
// of the nearest (entry distance bits with the child number in the two lowest bits).
//
// Arithmetic: the ray is folded into the node's grid once per node -- a = scale * inv, b = (origin - o) * inv -- and a plane
// costs one fma: t = fma(byte, a, b -/+ m).  The result only has to be CONSERVATIVE with respect to the slab test on the
// decoded plane p = fmaf(byte, scale, origin), which the host checked to enclose the exact boxes: near planes not later, far
// planes not earlier than fl(fl(p - o) * inv).  Both computations differ from the real number (byte * scale + origin - o) * inv
// by at most
//   |inv| * u * (3 P + 2 |o|)   [decode, subtract, multiply]   and   u |inv| (3 |o| + P) + 2 u m + u |t|   [this one],
// u = 2^-24, P = the largest |plane coordinate| of the scene's nodes; the sum is u |inv| (5 P + 6 |o|) + 3 u m < m =
// 8 u |inv| (P + |o|) (scale is a power of two in [2^-60, 2^36], so a is exact; tools/study_wide_walk.cpp checks at every node of
// every sample scene that this mask covers the decode-and-slab one).
//  * a plane byte is read as the f16 DENORMAL byte * 2^-24 (half-word 0x00bb) and the node record stores scale * 2^24
//    (device_layout.h), so that v_fma_mix_f32 converts the byte on the fly: fma(byte * 2^-24, (scale * 2^24) * inv, b) is the SAME
//    real number rounded once as fma(byte, scale * inv, b) -- one instruction per plane instead of a conversion and an fma, and two
//    plane bytes are unpacked by one v_and / v_perm (the kernel runs with f16 denormals on, .amdhsa_float_denorm_mode_16_64 3;
//    dr_kat_node_planes checks the instruction).  Powers of two scale exactly as long as nothing overflows: |inv| <= 2^60
//    (clamped), scale <= 2^36 (wide_builder.cpp refuses larger grids), so a <= 2^120.
//  * b -/+ m = fma(origin, inv, -(fl(o * inv) +- m)) with the ray-only part precomputed by wide_ray().  o * inv could overflow
//    for |o| > 2^67 where (origin - o) * inv did not: an axis with |o| >= 2^60 is left unconstrained (NaN margin).
//  * the cap of the exit distance is the running best itself (trav_begin: "no hit yet" = 10000).
// Extreme directions.  The test uses inv clamped to +-2^60 (WideRay::inv): for a zero direction component slab()
// computes (p - o) * inf = +-inf, or NaN (ignored) when p == o; (p - o) * 2^60 -/+ m, with m >= 2^20 then, lands on
// the same side of [0, 10000] for every plane at least m * 2^-60 away from the origin and leaves the nearer ones
// unconstrained -- a superset again, and such rays (a few per frame: N + random can cancel exactly) are still culled
// along that axis instead of walking the whole slab.  A NaN or > 2^60-long direction component makes m NaN: every plane of
// that axis becomes NaN and is ignored by max3 / min3 (the axis is not constrained at all).
// (The variants this replaced -- decode-and-slab, conversion + fma, compare + select -- are in the history of this file and
// A/B-measured in profiles/r3_m_node_step_ab.txt.)
struct WideRay { V3 inv, on, of; };      // inv: clamped 1/direction; on / of: fl(o * inv) + m, fl(o * inv) - m (what the near / far planes subtract)
__device__ __forceinline__ WideRay wide_ray_none() {      // a lane without a ray
  WideRay w;
  w.inv = w.on = w.of = mk(0, 0, 0);
  return w;
}
// The position margin of a ray when small triangles sit in the tree with their own bounds (WideMu, wide_tree = 2; proof: DESIGN.md 4.10).  The reference
// accepts a hit when Moeller-Trumbore, in floats, says so AND the leaf's padded box is entered; a tree built over the padded boxes visits every such leaf.
// Built over the triangles' own bounds B it still does if every box test is loosened by mu, where mu bounds how far outside B (per axis, and along the ray
// in units of position) a ray can pass and still be ACCEPTED by the float test.  With u = 2^-24, |a| >= 1e-4 for every accepted hit (hit_tri's cut-off),
// s = o - v0, E = |e1| |e2|, L = |e1| + |e2| (Euclidean norms):
//     |u_f - u^| , |v_f - v^| <= 6.03e-4 |d| |e| (8.52 |s| + 7.11 |e'|) + 2.02 u      (u^, v^, t^: the exact solution of o + t d = v0 + u e1 + v e2)
//     |t_f - t^| |d|          <= 6.03e-4 E |d| (8.52 |s| + 7.11 |t_f| |d|) + 2.02 u |t_f| |d|,   |t_f| |d| <= |s| + L + 0.02
// so the exact ray is inside B + mu_pos at t^, and B + mu_pos + |d| |t_f - t^| is entered no later than t_f and left after it, for
//     mu = |d| E (0.0197 |s| + 0.0086 L + 8.6e-5) + 1.2e-7 |s| + 3.0e-7 L      -- rounded up below to 0.021, 0.015, 1.2e-4 and 2^-21 (|s| + 4 L + 1), |s| <= |o| + |v0|.
// Above the cap the margin is the reference's own padding (B + 0.01 (1 + 2^-16) + the absolute term encloses the padded box): such rays -- camera rays with a
// long direction vector, mostly -- walk the tree exactly as they would the tree over padded boxes.  Rays or scenes beyond 2^30 take the cap too (no overflow
// inside the bound's arithmetic below that).
__device__ __forceinline__ float wide_ray_margin(V3 o, V3 d, float mu_e, float mu_l, float mu_v) {
  if (!(mu_e > 0.0f)) return 0.0f;
  const float on = __builtin_sqrtf(dot(o, o)) * 1.0001f, dn = __builtin_sqrtf(dot(d, d)) * 1.0001f, s = on + mu_v;
  const float abs_term = (s + 4.0f * mu_l + 1.0f) * 0x1p-21f;
  const float cap = 0.01f * (1.0f + 0x1p-16f) + abs_term;
  const float m = dn * mu_e * 1.0001f * (0.021f * s + 0.015f * mu_l + 1.2e-4f) * 1.0001f + abs_term;
  const bool tame = dn <= 0x1p30f && s <= 0x1p30f && dn * mu_e <= 2.0f;      // (false for NaNs)
  return (tame && m < cap) ? m : cap;
}
__device__ __forceinline__ WideRay wide_ray(V3 o, V3 d, V3 inv, float pmax, float mu_e, float mu_l, float mu_v) {      // (mu_*: the scene's WideMu)
  WideRay w;
  const float extra = wide_ray_margin(o, d, mu_e, mu_l, mu_v);
  auto one = [pmax, extra](float oa, float ia, float& ic, float& on, float& of) {
    ic = __builtin_fminf(__builtin_fmaxf(ia, -0x1p60f), 0x1p60f);                    // NaN -> -2^60, and the margin below is NaN
    const float k = __builtin_fmaf(pmax + __builtin_fabsf(oa), 0x1p-21f, 0x1p-40f) + extra;
    const float ai = __builtin_fabsf(ic);
    const float m = (ia == ia && ai > 0x1p-60f && __builtin_fabsf(oa) < 0x1p60f) ? ai * k : __builtin_nanf("");
    const float p = oa * ic;
    on = p + m; of = p - m;
  };
  one(o.x, inv.x, w.inv.x, w.on.x, w.of.x); one(o.y, inv.y, w.inv.y, w.on.y, w.of.y); one(o.z, inv.z, w.inv.z, w.on.z, w.of.z);
  return w;
}

// The four plane bytes of a word as f16 denormals: (byte0, byte2) and (byte1, byte3), each pair in one register
#ifdef DR_HOST_BUILD
struct PlanePairs { unsigned even, odd; };
__device__ __forceinline__ PlanePairs plane_pairs(unsigned w) { PlanePairs p; p.even = w & 0x00ff00ffu; p.odd = (w >> 8) & 0x00ff00ffu; return p; }
__device__ __forceinline__ float plane_t(const PlanePairs& p, int k, float a, float b) {     // fma(byte_k * 2^-24, a, b), rounded once
  const unsigned h = ((k & 1) ? p.odd : p.even) >> ((k & 2) ? 16 : 0) & 0xffffu;
  return __builtin_fmaf((float)h * 0x1p-24f, a, b);
}
#else
typedef _Float16 DrHalf2 __attribute__((ext_vector_type(2)));
struct PlanePairs { DrHalf2 even, odd; };
__device__ __forceinline__ PlanePairs plane_pairs(unsigned w) {
  PlanePairs p;
  p.even = __builtin_bit_cast(DrHalf2, w & 0x00ff00ffu);
  p.odd = __builtin_bit_cast(DrHalf2, __builtin_amdgcn_perm(0u, w, 0x0c030c01u));      // bytes {1, zero, 3, zero} of w: one v_perm_b32
  return p;
}
__device__ __forceinline__ float plane_t(const PlanePairs& p, int k, float a, float b) {     // v_fma_mix_f32: the f16 operand is widened inside the instruction
  const _Float16 h = (k & 1) ? ((k & 2) ? p.odd.y : p.odd.x) : ((k & 2) ? p.even.y : p.even.x);
  return __builtin_fmaf((float)h, a, b);
}
#endif

// three-input bit operations (one v_bitop3_b32 each on the device; spelled out, the compiler re-associates them into slower pairs)
#ifdef DR_HOST_BUILD
__device__ __forceinline__ unsigned bit_select(unsigned if0, unsigned if1, unsigned m) { return (if0 & ~m) | (if1 & m); }
__device__ __forceinline__ unsigned bit_and_or(unsigned a, unsigned b, unsigned c) { return (a & b) | c; }
__device__ __forceinline__ unsigned bit_andn(unsigned a, unsigned b, unsigned c) { return ~a & b & c; }
#else
__device__ __forceinline__ unsigned bit_select(unsigned if0, unsigned if1, unsigned m) { return __builtin_amdgcn_bitop3_b32(if0, if1, m, 0xd8); }
__device__ __forceinline__ unsigned bit_and_or(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xea); }
__device__ __forceinline__ unsigned bit_andn(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x08); }
#endif
template <class Sign>
__device__ __forceinline__ unsigned wide_node_test(u32x4 A, u32x4 B, u32x4 C, const WideRay& wr, const Sign& sg, float best_t, unsigned& near_key) {
  const float ox = __uint_as_float(A.x), oy = __uint_as_float(A.y), oz = __uint_as_float(A.z);
  const float sx24 = __uint_as_float(C.z << 16), sy24 = __uint_as_float(C.z & 0xffff0000u), sz24 = __uint_as_float(C.w);      // scale * 2^24 (x, y: upper halves of their floats)
  const V3 inv = wr.inv;
  // the plane entered first is `hi` for a negative direction (slab(): same rule, so the comparison stays plane by plane)
  // (sg holds the signs of the plain 1/direction; the clamped one differs for a NaN only, and a NaN axis has NaN planes whichever word is read)
  const unsigned nxw = sg.sel(0, B.x, B.w), fxw = sg.sel(0, B.w, B.x);
  const unsigned nyw = sg.sel(1, B.y, C.x), fyw = sg.sel(1, C.x, B.y);
  const unsigned nzw = sg.sel(2, B.z, C.y), fzw = sg.sel(2, C.y, B.z);
  unsigned fail[4], kk[4];
  const float tcap = best_t;
  const float ax = sx24 * inv.x, ay = sy24 * inv.y, az = sz24 * inv.z;
  const float bxn = __builtin_fmaf(ox, inv.x, -wr.on.x), bxf = __builtin_fmaf(ox, inv.x, -wr.of.x);
  const float byn = __builtin_fmaf(oy, inv.y, -wr.on.y), byf = __builtin_fmaf(oy, inv.y, -wr.of.y);
  const float bzn = __builtin_fmaf(oz, inv.z, -wr.on.z), bzf = __builtin_fmaf(oz, inv.z, -wr.of.z);
  const PlanePairs pnx = plane_pairs(nxw), pfx = plane_pairs(fxw), pny = plane_pairs(nyw), pfy = plane_pairs(fyw), pnz = plane_pairs(nzw), pfz = plane_pairs(fzw);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float t0x = plane_t(pnx, k, ax, bxn), t1x = plane_t(pfx, k, ax, bxf);
    const float t0y = plane_t(pny, k, ay, byn), t1y = plane_t(pfy, k, ay, byf);
    const float t0z = plane_t(pnz, k, az, bzn), t1z = plane_t(pfz, k, az, bzf);
    const float t_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(t0x, t0y), t0z), 0.0f);
    const float t_max = __builtin_fminf(__builtin_fminf(__builtin_fminf(t1x, t1y), t1z), tcap);
    // A child is entered when t_max >= t_min (>= where slab() has > and <=: a superset, which is all an internal node needs).  Taken from the SIGN
    // of t_max - t_min: neither is ever NaN (max / min ignore NaNs; 0 and the cap are numbers), t_max is finite or -inf, so the difference is a
    // number, +0 when they are equal.  (It is -0 only for t_max = -0, t_min = +0, where slab()'s own t_max > t_min fails anyway: still a superset.)
    // The sign, spread over a word, goes into the key (a failed child's key is all ones) and into the mask with fast bit operations instead of
    // compares and selects (tools/valu_rate.hip: 2.4 against 4.2 SIMD cycles each).
    fail[k] = (unsigned)((int)__float_as_uint(t_max - t_min) >> 31);
    kk[k] = bit_and_or(__float_as_uint(t_min), ~3u, fail[k]) | (unsigned)k;      // t_min >= 0: its bits order like the value
  }
  const unsigned k01 = kk[0] < kk[1] ? kk[0] : kk[1], k23 = kk[2] < kk[3] ? kk[2] : kk[3];
  near_key = k01 < k23 ? k01 : k23;
  // bit k of z = child k failed (the bits above are child 3's): three bitwise selects, then ~z & valid mask in one more
  const unsigned z = bit_select(bit_select(fail[3], fail[2], 4u), bit_select(fail[1], fail[0], 1u), 3u);      // bits 0-1 from the second, the rest from the first
  return bit_andn(z, A.w >> 24, 15u);
}

__device__ __forceinline__ unsigned wide_node_test(u32x4 A, u32x4 B, u32x4 C, V3 inv_plain, const WideRay& wr, float best_t, unsigned& near_key) {
  SignCmp sg; sg.inv = inv_plain;      // the planes chosen by comparing, as the reference does (per-tile kernel, host studies)
  return wide_node_test(A, B, C, wr, sg, best_t, near_key);
}

// (t, slot) as one 64-bit key whose unsigned order is the lexicographic order of the pair (t is positive, or the 10000 of
// "no hit yet" with slot -1 = the largest unsigned): what lanes that share one ray agree on with ds_min_u64
__device__ __forceinline__ unsigned long long hit_key(float t, int slot) { return ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)slot; }

// next record of this lane: the lowest pending child of the newest stack word, or -1 when nothing is left
__device__ __forceinline__ void wide_pop(Trav& tr, WideStack& ws, const int* __restrict__ stack) {
  if (ws.top == 0u) {
    if (ws.sp == ws.sb) { tr.node = -1; return; }
    ws.sp--;
    ws.top = (unsigned)stack[ws.sp * 64];
  }
  const int j = __builtin_ctz(ws.top);                       // the pending mask is never empty in a stored word
  tr.node = (int)((((ws.top >> 8) + (unsigned)j) << 1) | ((ws.top >> (4 + j)) & 1u));
  ws.top &= ws.top - 1u;
  ws.top = (ws.top & 15u) ? ws.top : 0u;
}

// One record = four 16-byte units; a node is its first three (device_layout.h), a leaf all four: the fourth is fetched by the lanes at a leaf only (a
// lane knows from the parent's leaf mask which it is) -- every load instruction of a step costs about 1 % of the frame (profiles/r4_o_node_three_units.txt).
// The empty asm pins the loads in front of the first use: without it hipcc sinks the leaf's primitive units behind the box test, a second dependent round trip.
struct WideRec { u32x4 A, B, C, D; };
__device__ __forceinline__ WideRec wide_fetch(WalkRsrc wide, int node) {
  const unsigned off = (unsigned)(node >> 1) << 6;
  WideRec r;
  r.A = ld_unit_raw(wide, off); r.B = ld_unit_raw(wide, off + 16); r.C = ld_unit_raw(wide, off + 32);
#ifndef DR_HOST_BUILD
  asm volatile("" : "=v"(r.D));      // (a node lane's fourth unit is never read: no value, no instruction)
  if (node & 1) r.D = ld_unit_raw(wide, off + 48);
  asm volatile("" : "+v"(r.A), "+v"(r.B), "+v"(r.C), "+v"(r.D));
#else
  r.D = ld_unit_raw(wide, off + 48);
#endif
  return r;
}

template <bool COUNT, class Sign>
__device__ __forceinline__ void wide_node_compute(const WideRec& r, const WideRay& wr, const Sign& sg, Trav& tr, WideStack& ws, int* __restrict__ stack, Ctr& c) {
  DR_MARK("node_begin");
  if (COUNT) c.V++;
  unsigned key;
  const unsigned mask = wide_node_test(r.A, r.B, r.C, wr, sg, tr.best_t, key);
  if (mask != 0u) {
    // nearest entered child next; the others wait as one stack word.  An unused child slot (inverted box, valid bit
    // clear) can only pass on a degenerate grid or ray; if it even has the smallest key, take the lowest valid one.
    int near = (int)(key & 3u);
    near = ((mask >> near) & 1u) ? near : __builtin_ctz(mask);
    const unsigned base = r.A.w & 0xffffffu, leafmask = r.A.w >> 28;
    const unsigned rest = mask & ~(1u << near);
    if (rest != 0u) {
      if (ws.top != 0u) { if (ws.sp < WIDE_STACK) { stack[ws.sp * 64] = (int)ws.top; ws.sp++; } }   // the host bounds the depth; the guard only protects LDS
      ws.top = (base << 8) | (leafmask << 4) | rest;
    }
    tr.node = (int)(((base + (unsigned)near) << 1) | ((leafmask >> near) & 1u));
  } else {
    wide_pop(tr, ws, stack);
  }
  DR_MARK("node_end");
}

template <bool COUNT, class Sign>
__device__ __forceinline__ void wide_leaf_compute(const WideRec& r, V3 o, V3 d, V3 inv, const Sign& sg, Trav& tr, WideStack& ws, const int* __restrict__ stack, Ctr& c) {
  auto f = [](unsigned v) { return __uint_as_float(v); };
  DR_MARK("leaf_begin");
  if (COUNT) c.V++;
  float mn[3] = {f(r.A.x), f(r.A.y), f(r.A.z)}, mx[3] = {f(r.B.x), f(r.B.y), f(r.B.z)};
  float dist;
#ifndef DR_HOST_BUILD
  if (!COUNT) {
    // The candidate is accepted when its primitive is hit AND the leaf's box is entered no farther than the best t (hit() K:484-488: box, then primitive):
    // a conjunction of pure tests, so the order is free.  The primitive first, for every lane (98 % of the lanes pass the box anyway); the box only for the
    // lanes whose primitive is a candidate -- 5 % of them --, and not at all when no lane of the wave has one: 0.5721 against 0.5777 ms/frame
    // (profiles/r4_phase_budget.txt).  The counting build and the host build keep the reference's order: L counts primitives tested behind a passed box.
    const int info = (int)r.A.w;
    const float t = prim_hit_kind((info >> WALK_SLOT_BITS) & 3, mk(f(r.B.w), f(r.C.x), f(r.C.y)), mk(f(r.C.z), f(r.C.w), f(r.D.x)), mk(f(r.D.y), f(r.D.z), f(r.D.w)), o, d);
    const int slot = info & ((1 << WALK_SLOT_BITS) - 1);
    const bool cand = t > 0.0f && (t < tr.best_t || (t == tr.best_t && (unsigned)slot < (unsigned)tr.best_slot));
    if (__builtin_amdgcn_ballot_w64(cand) != 0ull) {
      if (cand && slab_sel(o, inv, sg, mn, mx, dist) && dist <= tr.best_t) { tr.best_t = t; tr.best_slot = slot; }
    }
  } else
#endif
  if (slab_sel(o, inv, sg, mn, mx, dist) && dist <= tr.best_t) {      // <=: a box entered exactly at the best t may hold a tie with a lower slot
    if (COUNT) c.L++;
    const int info = (int)r.A.w;
    const float t = prim_hit_kind((info >> WALK_SLOT_BITS) & 3, mk(f(r.B.w), f(r.C.x), f(r.C.y)), mk(f(r.C.z), f(r.C.w), f(r.D.x)), mk(f(r.D.y), f(r.D.z), f(r.D.w)), o, d);
    const int slot = info & ((1 << WALK_SLOT_BITS) - 1);
    if (t > 0.0f && (t < tr.best_t || (t == tr.best_t && (unsigned)slot < (unsigned)tr.best_slot))) { tr.best_t = t; tr.best_slot = slot; }
  }
  wide_pop(tr, ws, stack);
  DR_MARK("leaf_end");
}

template <bool COUNT>
__device__ __forceinline__ void wide_node_compute(const WideRec& r, V3 inv, const WideRay& wr, Trav& tr, WideStack& ws, int* __restrict__ stack, Ctr& c) {
  SignCmp sg; sg.inv = inv;
  wide_node_compute<COUNT>(r, wr, sg, tr, ws, stack, c);
}
template <bool COUNT>
__device__ __forceinline__ void wide_leaf_compute(const WideRec& r, V3 o, V3 d, V3 inv, Trav& tr, WideStack& ws, const int* __restrict__ stack, Ctr& c) {
  SignCmp sg; sg.inv = inv;
  wide_leaf_compute<COUNT>(r, o, d, inv, sg, tr, ws, stack, c);
}
template <bool COUNT>
__device__ __forceinline__ Hit closest_hit_wide(WalkRsrc wide, float pmax, float mu_e, float mu_l, float mu_v, V3 o, V3 d, Ctr& c, int* __restrict__ stack /* [word * 64] */) {
  Trav tr;
  trav_begin(tr);
  WideStack ws; ws.top = 0u; ws.sp = 0; ws.sb = 0;
  const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const WideRay wr = wide_ray(o, d, inv, pmax, mu_e, mu_l, mu_v);
  if (COUNT) c.rays++;
  while (tr.node >= 0) {
    const WideRec r = wide_fetch(wide, tr.node);
    if (tr.node & 1) wide_leaf_compute<COUNT>(r, o, d, inv, tr, ws, stack, c);
    else wide_node_compute<COUNT>(r, inv, wr, tr, ws, stack, c);
  }
  Hit best; best.t = tr.best_slot < 0 ? -1.0f : tr.best_t; best.slot = tr.best_slot;
  return best;
}

#ifdef DR_HOST_BUILD
__device__ __forceinline__ bool first_active_lane() { return true; }      // a "wave" of one lane
#else
__device__ __forceinline__ bool first_active_lane() { return __lane_id() == (unsigned)__ffsll((long long)__ballot(1)) - 1u; }
#endif

template <bool COUNT>
__device__ __forceinline__ Hit closest_hit_threaded(WalkRsrc walk, V3 o, V3 d, Ctr& c) {
  Trav tr;
  trav_begin(tr);
  V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  if (COUNT) { c.rays++; if (first_active_lane()) c.ray_slots += 64; }
  while (tr.node >= 0) {
    if (COUNT) { if (first_active_lane()) c.trav_slots += 64; }
    trav_step<COUNT>(walk, o, d, inv, tr, c);
  }
  Hit best; best.t = tr.best_slot < 0 ? -1.0f : tr.best_t; best.slot = tr.best_slot;
  return best;
}

// hit() K:468-512 with the children visited near-first instead of child-0-first.
//
// What the reference returns is the primitive with the smallest t among all leaves whose
// boxes (and ancestors' boxes) the ray enters, and, among equal t, the one its child-0-first
// walk reaches first -- which is the one with the lowest leaf rank = slot (device_layout.h).
// So any visiting order gives the same answer as long as (1) a subtree is skipped only when
// its box is missed or entered no nearer than the best t so far (the reference's own pruning
// test, K:484) and (2) ties on t go to the lower slot.  One 64-byte record holds both child
// boxes; the far child waits on a per-lane stack in LDS (one dword per level, lane-major, so
// a wave's pushes and pops never bank-conflict).  The tree is a median split, so its depth is
// ceil(log2 N) <= 24 for N <= 2^24 (the host refuses deeper trees for this mode).

template <bool COUNT>
__device__ __forceinline__ Hit closest_hit_ordered(const DevPair* __restrict__ pairs, const DevPrim* __restrict__ prims,
                                                   V3 o, V3 d, Ctr& c, int* __restrict__ stack /* [level * 64] */) {
  Hit best; best.t = 10000000.0f; best.slot = 0x7fffffff;
  V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  int sp = 0;
  int cur = 0;
  if (COUNT) c.rays++;
  for (;;) {
    const float4* pp = reinterpret_cast<const float4*>(pairs + cur);
    float4 A = pp[0], B = pp[1], C = pp[2], D = pp[3];
    int c0 = __float_as_int(A.w), c1 = __float_as_int(B.w);
    float mn0[3] = {A.x, A.y, A.z}, mx0[3] = {B.x, B.y, B.z};
    float mn1[3] = {C.x, C.y, C.z}, mx1[3] = {D.x, D.y, D.z};
    float d0, d1;
    if (COUNT) c.V += 2;
    bool h0 = slab(o, inv, mn0, mx0, d0) && d0 <= best.t;     // <=: a box entered exactly at the best t may hold a tie with a lower slot
    bool h1 = slab(o, inv, mn1, mx1, d1) && d1 <= best.t;
    if (h0 && c0 < 0) {
      int slot = ~c0;
      if (COUNT) c.L++;
      float t = prim_hit(prims, slot, o, d);
      if (t > 0.0f && (t < best.t || (t == best.t && slot < best.slot))) { best.t = t; best.slot = slot; }
      h0 = false;
    }
    if (h1 && c1 < 0) {
      if (d1 <= best.t) {
        int slot = ~c1;
        if (COUNT) c.L++;
        float t = prim_hit(prims, slot, o, d);
        if (t > 0.0f && (t < best.t || (t == best.t && slot < best.slot))) { best.t = t; best.slot = slot; }
      }
      h1 = false;
    }
    h0 = h0 && d0 <= best.t;
    h1 = h1 && d1 <= best.t;
    if (h0 && h1) {
      bool first0 = d0 <= d1;
      int far_c = first0 ? c1 : c0;
      cur = first0 ? c0 : c1;
      if (sp < ORDERED_STACK) { stack[sp * 64] = far_c; sp++; }
    } else if (h0) {
      cur = c0;
    } else if (h1) {
      cur = c1;
    } else {
      if (sp == 0) break;
      sp--;
      cur = stack[sp * 64];
    }
  }
  if (best.slot == 0x7fffffff) { best.t = -1.0f; best.slot = -1; }
  return best;
}

// ------------------------------------------------------------------ shading helpers
__device__ __forceinline__ V3 reflect(V3 v, V3 n) {              // K:667 (2.0*dot narrows exactly)
  float k = 2.0f * dot(v, n);
  return v - splat(k) * n;
}
__device__ __forceinline__ V3 refract(V3 uv, V3 n, float eta) {  // K:678-683
  float cos_theta = (float)fmin((double)dot(uv * splat(-1.0f), n), 1.0);
  V3 perp = splat(eta) * (uv + splat(cos_theta) * n);
  float lp = length(perp);
  float par = (float)(-__builtin_sqrt(__builtin_fabs(1.0 - (double)(lp * lp))));
  return perp + splat(par) * n;
}
__device__ __forceinline__ float reflectance(float cosine, float ref_idx) {   // K:686-691
  float r0 = (1 - ref_idx) / (1 + ref_idx);
  r0 = r0 * r0;
  float a = 1 - cosine;
  float r = a;             // pow(float, int 5): CUDA powif = square and multiply in float
  a = a * a;
  a = a * a;
  r = r * a;
  return r0 + (1 - r0) * r;
}

// tex2D<uchar4>, point / wrap / normalised (K:830,841,962; descriptor K:1959-1964)
template <bool COUNT>
__device__ __forceinline__ uint32_t tex_fetch(const DevTex* __restrict__ tex, const uint32_t* __restrict__ texels, int id,
                                              float u, float v, Ctr& c) {
  DevTex t = tex[id];
  if (COUNT) c.T++;
  float fu = u - __builtin_floorf(u), fv = v - __builtin_floorf(v);
  int i = f2i(__builtin_floorf(fu * (float)t.w)), j = f2i(__builtin_floorf(fv * (float)t.h));
  if (i > t.w - 1) i = t.w - 1;
  if (j > t.h - 1) j = t.h - 1;
  if (i < 0) i = 0;
  if (j < 0) j = 0;
  return texels[(size_t)t.offset + (size_t)j * (size_t)t.w + (size_t)i];
}
// b / 255 for a texel byte b, as the IEEE division rounds it, in three fast instructions instead of the division's eleven: with c = RN(1 / 255),
// q0 = RN(b c), the residual b - 255 q0 is exact in an fma and q = RN(q0 + residual * c) is the correctly rounded quotient (all 256 bytes are compared
// with the division in tools/host_kernel.cpp hk_check_uniform).
__device__ __forceinline__ float byte_over_255(uint32_t b) {
  const float x = float(b), c = 0x1.010102p-8f;
  const float q0 = x * c;
  return __builtin_fmaf(__builtin_fmaf(-255.0f, q0, x), c, q0);
}
__device__ __forceinline__ V3 rgb_of(uint32_t px) {
  return mk(byte_over_255(px & 255u), byte_over_255((px >> 8) & 255u), byte_over_255((px >> 16) & 255u));
}

// getnormal K:703-773 on the device records of one primitive (prims: pA..pC, shade: s0..s4, s6): the unflipped normal and the
// interpolated texture coordinate.  Barycentrics are recomputed from the ray origin, not from the hit point (K:728-745);
// the file's face normal is used unless its z is the -20 sentinel (then the cross product), vertex normals only when the
// smooth flag is set and n1.z is not the sentinel; spheres divide by the radius, other types normalise (hit - pos).
__device__ __forceinline__ V3 surface_normal(float4 pA, float4 pB, float4 pC, float4 s0, float4 s1, float4 s2, float4 s3, float4 s4, float4 s6,
                                             V3 rayo, V3 raydir, V3 hitpoint, V3& texco) {
  int type = __float_as_int(pC.y);
  texco = mk(0, 0, 0);
  V3 N;
  if (type == 0) {
    N = (hitpoint - mk(pA.x, pA.y, pA.z)) / splat(pA.w);
  } else if (type == 2) {
    V3 v0 = mk(pA.x, pA.y, pA.z), v0v1 = mk(pA.w, pB.x, pB.y), v0v2 = mk(pB.z, pB.w, pC.x);
    N = cross(v0v1, v0v2);
    V3 pvec = cross(raydir, v0v2);
    float det = dot(v0v1, pvec);
    float invDet = 1 / det;
    V3 tvec = rayo - v0;
    float ux = dot(tvec, pvec) * invDet;
    V3 qvec = cross(tvec, v0v1);
    float uy = dot(raydir, qvec) * invDet;
    float uz = 1 - ux - uy;
    // texco = uz*t1 + ux*t2 + uy*t3 (the z components of t1..t3 are never read again)
    texco.x = uz * s3.x + ux * s3.z + uy * s4.x;
    texco.y = uz * s3.y + ux * s3.w + uy * s4.y;
    V3 fn = mk(s0.x, s0.y, s0.z);
    if (fn.z != -20) {
      N = fn;
      int flags = __float_as_int(s6.z);
      if (s1.y != -20 && (flags & 1)) {     // n1.z
        V3 n1 = mk(s0.w, s1.x, s1.y), n2 = mk(s1.z, s1.w, s2.x), n3 = mk(s2.y, s2.z, s2.w);
        N = splat(uz) * n1 + splat(ux) * n2 + splat(uy) * n3;
      }
    }
    N = normalized(N);
  } else {
    N = normalized(hitpoint - mk(pA.x, pA.y, pA.z));
  }
  return N;
}

// raycolor K:787-982 is split at its natural seams so that the per-pixel kernel and the
// persistent kernel run the same arithmetic: shade_hit = the "hit > 0" branch of one bounce
// (K:807-950), shade_miss = the background branch (K:951-976).
struct Path { V3 rayo, raydir, atten; };

// shade_hit in three parts, so that the persistent kernel can run ONE rejection loop per phase for the lanes that scatter and the lanes that start a
// path (rand_points_merged): everything up to the draws (shade_prepare: K:807-848 and glossy's one uniform), the point in the unit sphere, and the
// scatter itself (shade_scatter: K:848-944).  A lane's draws keep the reference's order.
struct ShadeCtx { V3 hitpoint, N, ocolor; float add_x, rough, ir, r5; int mat; bool front; };
__device__ __forceinline__ bool shade_needs_sphere(const ShadeCtx& sc) { return sc.mat == 0 || sc.mat == 3 || sc.mat == 5; }

// Returns false when the path ends at an emissive surface, with `emitted` as its radiance (K:941-944); true: sc is filled
template <bool COUNT>
__device__ __forceinline__ bool shade_prepare(const RenderParams& P, const Path& path, float t, int slot, Xorwow& rng, Ctr& c, ShadeCtx& sc, V3& emitted) {
  const V3 rayo = path.rayo, raydir = path.raydir;
  if (COUNT) c.S++;
  V3 hitpoint = rayo + splat(t) * raydir;
  const float4* pp = reinterpret_cast<const float4*>(P.prims + slot);
  const float4* sp = reinterpret_cast<const float4*>(P.shade + slot);
  float4 pA = pp[0], pB = pp[1], pC = pp[2];
  float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4], s5 = sp[5], s6 = sp[6];
  V3 texco;
  V3 N = surface_normal(pA, pB, pC, s0, s1, s2, s3, s4, s6, rayo, raydir, hitpoint, texco);
  bool front = dot(raydir, N) < 0;
  N = front ? N : N * splat(-1.0f);
  // ---- material inputs K:826-844
  V3 col = mk(s4.z, s4.w, s5.x);
  float add_x = s5.y, rough = s5.z;
  int mat = __float_as_int(s5.w), texnum = __float_as_int(s6.x), rtexnum = __float_as_int(s6.y);
  int flags = __float_as_int(s6.z);
  V3 ocolor = col;
  if (texnum >= 0) {
    ocolor = rgb_of(tex_fetch<COUNT>(P.tex, P.texels, texnum, texco.x, -texco.y + 1, c));
  } else if (flags & 2) {                     // checker K:776-784
    float u2 = __builtin_floorf(texco.x * 10), v2 = __builtin_floorf(texco.y * 10);
    float yes = u2 + v2;
    ocolor = (__builtin_fmodf(yes, 2.0f) == 0) ? splat(0.8f) : col;
  }
  if (rtexnum >= 0) {
    uint32_t px = tex_fetch<COUNT>(P.tex, P.texels, rtexnum, texco.x, -texco.y + 1, c);
    rough = byte_over_255(px & 255u) / 2;
  }
  if (!(mat == 0 || mat == 2 || mat == 3 || mat == 4 || mat == 5)) { emitted = ocolor * path.atten; return false; }      // emissive (and every unknown material) K:941-944
  // ---- the random draws come first (K:848-944), ONE copy of each loop for every lane that needs it: diffuse (0), metal (3) and glossy (5) all draw
  // a point in the unit sphere, glossy after its one uniform -- the order of draws of each lane is the reference's, but a wave runs the rejection
  // loop once instead of once per material branch (the five copies of that loop were most of the shade phase)
  sc.r5 = 0.0f;
  if (mat == 5) sc.r5 = randy(rng);
  sc.hitpoint = hitpoint; sc.N = N; sc.ocolor = ocolor; sc.add_x = add_x; sc.rough = rough; sc.ir = s5.z; sc.mat = mat; sc.front = front;      // ir: b[g].addional.y itself, not the textured roughness (K:917)
  return true;
}

// the scatter of one bounce (rs: the point in the unit sphere, for the materials that draw one): rayo / raydir / atten updated
__device__ __forceinline__ void shade_scatter(Path& path, const ShadeCtx& sc, V3 rs, Xorwow& rng) {
  V3& rayo = path.rayo; V3& raydir = path.raydir; V3& atten = path.atten;
  const V3 hitpoint = sc.hitpoint, N = sc.N, ocolor = sc.ocolor;
  const int mat = sc.mat;
  const bool like_metal = mat == 3 || (mat == 5 && sc.r5 > 0.8);      // float compared with the double 0.8
  if (mat == 0 || (mat == 5 && !like_metal)) {
    V3 target = hitpoint + N;
    if (mat == 5 || sc.add_x == 0) target = target + rs;              // glossy never normalises (K:899)
    else target = target + normalized(rs);
    atten = atten * ocolor;
    rayo = hitpoint;
    raydir = normalized(target - hitpoint);
  } else if (mat == 2) {
    atten = atten * ocolor;
    rayo = hitpoint;
    raydir = reflect(normalized(raydir), N);
  } else if (like_metal) {
    V3 refl = reflect(normalized(raydir), N);
    atten = atten * ocolor;
    rayo = hitpoint;
    raydir = refl + splat(sc.rough) * rs;
  } else {      // mat == 4
    float ir = sc.ir;
    float ratio = sc.front ? (1.0f / ir) : ir;
    float cos_theta = (float)fmin((double)dot(normalized(raydir) * splat(-1.0f), N), 1.0);
    float sin_theta = (float)__builtin_sqrt(1.0 - (double)(cos_theta * cos_theta));
    bool cannot = (ratio * sin_theta) > 1.0f;
    V3 out;
    if (cannot || reflectance(cos_theta, ratio) > randy(rng)) out = reflect(normalized(raydir), N);
    else out = refract(normalized(raydir), N, ratio);
    atten = atten * ocolor;
    rayo = hitpoint;
    raydir = out;
  }
}

// Returns true when the path continues (rayo/raydir/atten updated), false when it ends at an
// emissive surface with `emitted` as its radiance (K:941-944).
template <bool COUNT>
__device__ __forceinline__ bool shade_hit(const RenderParams& P, Path& path, float t, int slot, Xorwow& rng, Ctr& c, V3& emitted) {
  ShadeCtx sc;
  if (!shade_prepare<COUNT>(P, path, t, slot, rng, c, sc, emitted)) return false;
  V3 rs = mk(0, 0, 0);
  if (shade_needs_sphere(sc)) rs = rand_in_unit_sphere(rng);
  shade_scatter(path, sc, rs, rng);
  return true;
}

template <bool COUNT>
__device__ __forceinline__ V3 shade_miss(const RenderParams& P, const Path& path, Ctr& c) {
  const V3 raydir = path.raydir, atten = path.atten;
  V3 u = normalized(raydir);
  if (P.backtex > -1) {                       // K:953-966
    double ux = (double)u.x, uy = (double)u.y, uz = (double)u.z + 1.;
    float m = (float)(2. * __builtin_sqrt(ux * ux + uy * uy + uz * uz));
    V3 tt = u / splat(m) + splat(.5f);
    tt.y = -tt.y;
    V3 colr = rgb_of(tex_fetch<COUNT>(P.tex, P.texels, P.backtex, tt.x, -tt.y + 1, c));
    return atten * colr * splat(P.bgint);
  }
  float t2 = (float)(0.5 * ((double)u.y + 1.0));   // K:971-974
  float omt = (float)(1.0 - (double)t2);
  V3 sky = splat(omt) * mk(1.0f, 1.0f, 1.0f) + splat(t2) * mk(0.5f, 0.7f, 1.0f);
  return atten * sky * splat(P.bgint);
}

// One path: raycolor K:787-982.  `closest` is the traversal functor.
template <bool COUNT, class Closest>
__device__ __forceinline__ V3 trace_path(const RenderParams& P, const Closest& closest, V3 origin, V3 dir, Xorwow& rng, Ctr& c) {
  Path path; path.rayo = origin; path.raydir = dir; path.atten = splat(1.0f);
  for (int i = 0; i < P.max_depth; i++) {
    Hit h = closest(path.rayo, path.raydir, c);
    if (h.t > 0.0f) {
      V3 emitted;
      if (!shade_hit<COUNT>(P, path, h.t, h.slot, rng, c, emitted)) return emitted;
    } else {
      return shade_miss<COUNT>(P, path, c);
    }
  }
  return mk(0, 0, 0);
}

// Camera ray of one sample: K:1065-1073 (rng must be freshly seeded), in two parts around the point in the unit disk (the persistent kernel draws
// it in the phase's one rejection loop, rand_points_merged)
__device__ __forceinline__ void camera_prepare(const RenderParams& P, int x, int y, Xorwow& rng, float& nu, float& nv) {
  nu = (float)(((double)(float)x + rng.uniform_double()) / P.den_w);
  nv = (float)(((double)(float)y + rng.uniform_double()) / P.den_h);
}
__device__ __forceinline__ void camera_finish(const RenderParams& P, float nu, float nv, V3 disk, V3& origin, V3& dir) {
  V3 from = ld3(P.from), llc = ld3(P.llc), hor = ld3(P.hor), ver = ld3(P.ver), uu = ld3(P.uu), vu = ld3(P.vu);
  V3 rd = splat(P.lens_radius) * disk;
  V3 offset = uu * splat(rd.x) + vu * splat(rd.y);
  dir = llc + splat(nu) * hor + splat(nv) * ver - from - offset;
  origin = from + offset;
}
__device__ __forceinline__ void camera_ray(const RenderParams& P, int x, int y, Xorwow& rng, V3& origin, V3& dir) {
  float nu, nv;
  camera_prepare(P, x, y, rng, nu, nv);
  camera_finish(P, nu, nv, rand_in_unit_disk(rng), origin, dir);
}
__device__ __forceinline__ uint64_t sample_seed(const RenderParams& P, int x, int y, int s, int frame = 0) {   // K:1065 with clock() := frame seed
  return P.seed + (uint64_t)frame * P.batch_seed_stride + (uint64_t)s * 0x9E3779B97F4A7C15ull +
         (uint64_t)((uint32_t)x + (uint32_t)y * P.seed_stride);
}
// (frame_offset: launches that render every frame of their batch into a buffer of its own -- the grouped present pipeline -- pass frame * P.out_frame_stride)
__device__ __forceinline__ void store_pixel(const RenderParams& P, int x, int y, V3 color, size_t frame_offset = 0) {     // K:1081-1085
  int r = f2i(color.x * 255 * P.scale), g = f2i(color.y * 255 * P.scale), b = f2i(color.z * 255 * P.scale);
  int32_t* px = P.out + frame_offset + ((size_t)x * (size_t)P.H + (size_t)y) * 3;
  if (P.accumulate == 2) {        // frames of one batch may finish the same pixel concurrently; integer adds commute
    atomicAdd(px + 0, r); atomicAdd(px + 1, g); atomicAdd(px + 2, b);
  } else if (P.accumulate) { px[0] += r; px[1] += g; px[2] += b; }
  else { px[0] = r; px[1] = g; px[2] = b; }
}

// Kernel K:998-1093 for one pixel (the camera basis comes precomputed in P).
template <bool COUNT, class Closest>
__device__ __forceinline__ void render_pixel(const RenderParams& P, const Closest& closest, int x, int y, Ctr& c) {
  V3 color = mk(0, 0, 0);
  for (int s = 0; (float)s < P.spp_f; ++s) {
    Xorwow rng;
    rng.init(sample_seed(P, x, y, s));
    if (COUNT) c.samples++;
    V3 origin, dir;
    camera_ray(P, x, y, rng, origin, dir);
    color = color + trace_path<COUNT>(P, closest, origin, dir, rng, c);
  }
  store_pixel(P, x, y, color);
}

}  // namespace dr
