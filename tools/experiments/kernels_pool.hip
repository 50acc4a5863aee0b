// The pool kernel: the megakernel (Kernel / raycolor / hit, kernel.cu K:468-512, K:787-982, K:998-1093) with its three kinds of
// work -- node steps, leaf steps, shading -- each run at (nearly) full wave width.
//
// In the persistent kernel (kernels_render.hip) a path belongs to a LANE, and at any moment a wave's 64 lanes are spread over the
// three kinds of work: measured on the bench scene a node step executes for 30 of its 64 lanes, a leaf step for 23, a shade phase
// for 37 (lane use 0.36, profiles/r2_b).  Here a path belongs to a WAVE, which owns POOL_SLOTS = 128 of them -- twice its width --
// with the state the walk needs in LDS (ray, 1/direction, best hit, next record, stack top and a few stack words: 80-84 bytes each) and
// the rest (RNG, attenuation, pixel: 48 bytes, touched once per bounce) in a per-wave array in global memory.  Per iteration the
// wave reads the stage tag of its 128 paths, picks ONE kind of step, compacts up to 64 paths that wait for that kind onto its lanes
// (ranks from ballots, slot numbers through a 64-word exchange buffer), loads their state, does the step, stores the state and the
// new tags.  With two paths per lane there are nearly always 64 that want a node step, or 64 that want a leaf step (a Monte-Carlo
// model with the bench scene's transition rates, tools/sim_pool.py: fill 61 / 59 / 49 of 64 for node / leaf / shade at 128 paths, 37 /
// 31 / 25 at 64 -- the persistent kernel's situation -- and 1.8 x fewer wave-level instructions per ray).
// Which lane carries a path does not enter its arithmetic (same device functions, same order of draws; the seed is a function of
// x, y and the frame, K:1065): frames are bit-identical to the other kernels'.
//
// Waves never talk to each other: no queue, no atomic in LDS, no wait on another wave -- round 3's first version of this kernel kept
// ONE pool per workgroup with three shared queues; its batches were 63.9 lanes full and it was 2 x slower than the persistent
// kernel (profiles/r3_a_pool_kernel_lds_queues.txt: 56 % of a wave's life went into claiming and pushing, about 15 dependent LDS
// round trips per batch).  Here a batch costs three: tags, exchange buffer, state.
// The second version kept ALL of a path's state in LDS (144 bytes): 8 waves per CU, and although its lanes were 68 % busy and it needed
// 104 instead of 158 wave-level VALU instructions per ray, two waves per SIMD cannot hide a step's latency (VALU issuing 28 % of the
// time; 0.99 ms/frame against 0.62, profiles/r3_b_pool_kernel_wave_private_8waves.txt).  Hence the split above: 15-16 waves per CU.
//
// New pixels come from the same per-XCD queues, tile order and frame batching as the persistent kernel's, counted in PIXELS: a shade
// batch takes as many positions of its region's pixel sequence as it has paths without a pixel with one atomic add.  A path
// that finds no pixel left dies; the wave leaves when all of its paths are dead.
//
// RESULT (profiles/r3_c_pool_kernel_more_waves.txt): batches 54-64 lanes full, 104 instead of 158 wave-level VALU instructions per ray, and
// still 16 % SLOWER than the persistent kernel (0.717 against 0.616 ms/frame; C2 and C5 likewise): a batch's step is a chain of about 4 800
// cycles for about 260 VALU instructions, so the 4 waves per SIMD that LDS allows keep the VALU 40 % busy, against 65-69 % for the six
// register-resident waves of the persistent kernel.  Built only with -DDOGERAY_EXPERIMENTAL (tools/exp_variant.sh).
//
// The stack of a path is LSTACK (3-4) words in LDS besides the word in `top`; deeper words (the host bounds the depth at WIDE_STACK) go to
// the wave's scratch in global memory.
#include <hip/hip_runtime.h>

#include "device_core.hpp"
#include "kernels.hpp"
#include "../../include/dogeray_amd.h"

namespace dr {

#ifdef DOGERAY_EXPERIMENTAL      // MEASURED SLOWER than the persistent kernel on every scene (profiles/r3_c_pool_kernel_more_waves.txt): not in the product library

constexpr int POOL_SLOTS = 128;                 // paths per wave: slot s belongs to lane s & 63
constexpr int POOL_UNITS = 4;                   // 16-byte state units per path in LDS
constexpr int POOL_GUNITS = 3;                  // ... and in the wave's global array
enum { PT_NODE = 0, PT_LEAF = 1, PT_SHADE = 2, PT_DEAD = 3 };
// LDS:    U0 {o.xyz, best t} U1 {d.xyz, best slot} U2 {1/d.xyz, next record} U3 {stack top, stack pointer, steps, pixel code}
// global: G0 {rng v0..v3} G1 {rng v4, rng d, atten.xy} G2 {atten.z, (x + 1) << 16 | y, frame | bounce << 8, -}
constexpr unsigned POOL_META_NEW = 0x80000000u; // the slot holds no path yet (or its path has ended): it needs a pixel
constexpr int pool_wave_lds(int lstack) { return POOL_UNITS * POOL_SLOTS * 16 + lstack * POOL_SLOTS * 4 + POOL_SLOTS + 64 * 4; }      // bytes per wave
constexpr int pool_wave_scratch(int lstack) { return (WIDE_STACK - lstack) * POOL_SLOTS + POOL_GUNITS * 4 * POOL_SLOTS; }               // 32-bit words per wave

// the per-path stack: words [0, LSTACK) in LDS, deeper ones in the wave's global scratch
template <int LSTACK>
struct PoolStack {
  int* lds; unsigned* glob;   // both already offset by the slot
  __device__ __forceinline__ int ld(int k) const {
    if (k < LSTACK) return lds[k * POOL_SLOTS];
    return (int)__hip_atomic_load(glob + (k - LSTACK) * POOL_SLOTS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void st(int k, int v) const {
    if (k < LSTACK) lds[k * POOL_SLOTS] = v;
    else __hip_atomic_store(glob + (k - LSTACK) * POOL_SLOTS, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

// wide_pop (device_core.hpp) on a PoolStack
template <class PoolStack>
__device__ __forceinline__ void pool_pop(int& node, unsigned& top, int& sp, const PoolStack& stk) {
  if (top == 0u) {
    if (sp == 0) { node = -1; return; }
    sp--;
    top = (unsigned)stk.ld(sp);
  }
  const int j = __builtin_ctz(top);
  node = (int)((((top >> 8) + (unsigned)j) << 1) | ((top >> (4 + j)) & 1u));
  top &= top - 1u;
  top = (top & 15u) ? top : 0u;
}

__device__ __forceinline__ unsigned pool_lane_rank(unsigned long long m) {
  return (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ int pool_next_tag(int node) { return node < 0 ? PT_SHADE : ((node & 1) ? PT_LEAF : PT_NODE); }

// LSTACK stack words per path in LDS, WG_WAVES waves per workgroup (independent of each other), WPS waves per SIMD the registers allow
template <bool DIAG, int LSTACK, int WG_WAVES, int WPS>
__global__ __launch_bounds__(WG_WAVES * 64, WPS) void render_pool_kernel(RenderParams P, unsigned* __restrict__ tile_counter, const int* __restrict__ tile_order,
                                                                         const int* __restrict__ region_start, unsigned* __restrict__ pixel_cost,
                                                                         unsigned* __restrict__ scratch, int shade_min) {
  __shared__ __attribute__((aligned(16))) float4 lds_units[WG_WAVES * POOL_UNITS * POOL_SLOTS];
  __shared__ int lds_stack[WG_WAVES * LSTACK * POOL_SLOTS];
  __shared__ unsigned char lds_tags[WG_WAVES * POOL_SLOTS];
  __shared__ int lds_xch[WG_WAVES * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4* const units = lds_units + wave * (POOL_UNITS * POOL_SLOTS);      // unit u of slot s at units[u * POOL_SLOTS + s]
  int* const stackw = lds_stack + wave * (LSTACK * POOL_SLOTS);            // word k of slot s at stackw[k * POOL_SLOTS + s]
  unsigned char* const tags = lds_tags + wave * POOL_SLOTS;
  int* const xch = lds_xch + wave * 64;
  unsigned* const my_scratch = scratch + ((size_t)blockIdx.x * WG_WAVES + (size_t)wave) * (size_t)pool_wave_scratch(LSTACK);
  float4* const gunits = reinterpret_cast<float4*>(my_scratch + (WIDE_STACK - LSTACK) * POOL_SLOTS);      // unit k of slot s at gunits[s * POOL_GUNITS + k]
  const WalkRsrc walk = wide_rsrc(P);
  const unsigned long long t_begin = __builtin_readcyclecounter(), r_begin = __builtin_amdgcn_s_memrealtime();

  // ---- every slot starts without a path and waits for a pixel
  tags[lane] = PT_SHADE; tags[lane + 64] = PT_SHADE;
  gunits[lane * POOL_GUNITS + 2] = make_float4(0.0f, 0.0f, __uint_as_float(POOL_META_NEW), 0.0f);
  gunits[(lane + 64) * POOL_GUNITS + 2] = make_float4(0.0f, 0.0f, __uint_as_float(POOL_META_NEW), 0.0f);

  // work queues as in the persistent kernel (tile order of the cost feedback, frames of a batch interleaved, one queue per region =
  // XCD; a wave helps the next region once its own is empty), but counted in pixels
  int region = 0, regions_left = P.regions;
  if (P.regions > 1) region = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) % (unsigned)P.regions);   // HW_REG_XCC_ID
  Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
  // DIAG build (option pool_diag): batches and paths per stage, shader cycles spent choosing / inside the step
  unsigned long long d_t_sel = 0, d_t_node = 0, d_t_leaf = 0, d_t_shade = 0, d_t_nstate = 0, d_t_nfetch = 0;
  unsigned d_b_node = 0, d_b_leaf = 0, d_b_shade = 0, d_l_node = 0, d_l_leaf = 0, d_l_shade = 0;

  for (;;) {
    unsigned long long d_t0 = 0;
    if (DIAG) d_t0 = __builtin_readcyclecounter();
    // ---- what do my 128 paths wait for?
    const int t0 = tags[lane], t1 = tags[lane + 64];
    const int cn = (int)__popcll(__ballot(t0 == PT_NODE)) + (int)__popcll(__ballot(t1 == PT_NODE));
    const int cl = (int)__popcll(__ballot(t0 == PT_LEAF)) + (int)__popcll(__ballot(t1 == PT_LEAF));
    const int cs = (int)__popcll(__ballot(t0 == PT_SHADE)) + (int)__popcll(__ballot(t1 == PT_SHADE));
    if (cn + cl + cs == 0) break;                                // all dead
    // shading makes the rays the walk needs, but a shade batch is the most expensive one: once shade_min paths wait for it (or nothing
    // else can be done); otherwise the kind more paths wait for
    const int stage = (cs >= shade_min || cn + cl == 0) ? PT_SHADE : (cn >= cl ? PT_NODE : PT_LEAF);
    // ---- compact: the k-th path of that kind goes to lane k (the first 64 of them)
    const unsigned long long m0 = __ballot(t0 == stage), m1 = __ballot(t1 == stage);
    const int n0 = (int)__popcll(m0);
    const int total = n0 + (int)__popcll(m1);
    const int n = total < 64 ? total : 64;
    if (t0 == stage) xch[pool_lane_rank(m0)] = lane;
    const int r1 = n0 + (int)pool_lane_rank(m1);
    if (t1 == stage && r1 < 64) xch[r1] = lane + 64;
    const int slot = lane < n ? xch[lane] : -1;
    unsigned long long d_t1 = 0;
    if (DIAG) {
      d_t1 = __builtin_readcyclecounter(); d_t_sel += d_t1 - d_t0;
      if (stage == PT_NODE) { d_b_node++; d_l_node += (unsigned)n; } else if (stage == PT_LEAF) { d_b_leaf++; d_l_leaf += (unsigned)n; } else { d_b_shade++; d_l_shade += (unsigned)n; }
    }
    const int sidx = slot < 0 ? 0 : slot;                        // inactive lanes address slot 0 and touch nothing
    float* const u0 = reinterpret_cast<float*>(units + 0 * POOL_SLOTS + sidx);
    float* const u1 = reinterpret_cast<float*>(units + 1 * POOL_SLOTS + sidx);
    float* const u2 = reinterpret_cast<float*>(units + 2 * POOL_SLOTS + sidx);
    float* const u3 = reinterpret_cast<float*>(units + 3 * POOL_SLOTS + sidx);
    PoolStack<LSTACK> stk; stk.lds = stackw + sidx; stk.glob = my_scratch + sidx;

    unsigned long long d_ta = 0, d_tb = 0;
    if (stage == PT_NODE) {
      // ================= node step: wide_node_compute (device_core.hpp) on the path's state
      if (slot >= 0) {
        const float4 A0 = *reinterpret_cast<const float4*>(u0), A2 = *reinterpret_cast<const float4*>(u2);
        const float4 A3 = *reinterpret_cast<const float4*>(u3);
        int node = __float_as_int(A2.w);
        if (DIAG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); d_ta = __builtin_readcyclecounter(); }
        const WideRec r = wide_fetch(walk, node);
        if (DIAG) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); d_tb = __builtin_readcyclecounter(); }
        const V3 o = mk(A0.x, A0.y, A0.z), inv = mk(A2.x, A2.y, A2.z);
        const WideRay wr = wide_ray(o, inv, P.wide_pmax);
        unsigned top = __float_as_uint(A3.x); int sp = __float_as_int(A3.y);
        unsigned key;
        const unsigned mask = wide_node_test(r.A, r.B, r.C, r.D, o, inv, wr, A0.w, key);
        if (mask != 0u) {
          int near = (int)(key & 3u);
          near = ((mask >> near) & 1u) ? near : __builtin_ctz(mask);
          const unsigned cbase = r.A.w & 0xffffffu, leafmask = (r.B.w >> 4) & 15u;
          const unsigned rest = mask & ~(1u << near);
          if (rest != 0u) {
            if (top != 0u) { if (sp < WIDE_STACK) { stk.st(sp, (int)top); sp++; } }
            top = (cbase << 8) | (leafmask << 4) | rest;
          }
          node = (int)(((cbase + (unsigned)near) << 1) | ((leafmask >> near) & 1u));
        } else {
          pool_pop(node, top, sp, stk);
        }
        u2[3] = __int_as_float(node);
        u3[0] = __uint_as_float(top); u3[1] = __int_as_float(sp); u3[2] = __uint_as_float(__float_as_uint(A3.z) + 1u);
        tags[slot] = pool_next_tag(node);
      }
    } else if (stage == PT_LEAF) {
      // ================= leaf step: wide_leaf_compute -- the reference's exact box, then the primitive (hit() K:484-497)
      if (slot >= 0) {
        const float4 A0 = *reinterpret_cast<const float4*>(u0), A1 = *reinterpret_cast<const float4*>(u1), A2 = *reinterpret_cast<const float4*>(u2);
        const float4 A3 = *reinterpret_cast<const float4*>(u3);
        int node = __float_as_int(A2.w);
        const WideRec r = wide_fetch(walk, node);
        const V3 o = mk(A0.x, A0.y, A0.z), d = mk(A1.x, A1.y, A1.z), inv = mk(A2.x, A2.y, A2.z);
        float best_t = A0.w; int best_slot = __float_as_int(A1.w);
        unsigned top = __float_as_uint(A3.x); int sp = __float_as_int(A3.y);
        auto f = [](unsigned v) { return __uint_as_float(v); };
        float mn[3] = {f(r.A.x), f(r.A.y), f(r.A.z)}, mx[3] = {f(r.B.x), f(r.B.y), f(r.B.z)};
        float dist;
        if (slab(o, inv, mn, mx, dist) && dist <= best_t) {      // <=: a box entered exactly at the best t may hold a tie with a lower slot
          const int info = (int)r.A.w;
          const float t = prim_hit_kind((info >> WALK_SLOT_BITS) & 3, mk(f(r.B.w), f(r.C.x), f(r.C.y)), mk(f(r.C.z), f(r.C.w), f(r.D.x)), mk(f(r.D.y), f(r.D.z), f(r.D.w)), o, d);
          const int ps = info & ((1 << WALK_SLOT_BITS) - 1);
          if (t > 0.0f && (t < best_t || (t == best_t && (unsigned)ps < (unsigned)best_slot))) { best_t = t; best_slot = ps; }
        }
        pool_pop(node, top, sp, stk);
        u0[3] = best_t; u1[3] = __int_as_float(best_slot);
        u2[3] = __int_as_float(node);
        u3[0] = __uint_as_float(top); u3[1] = __int_as_float(sp); u3[2] = __uint_as_float(__float_as_uint(A3.z) + 1u);
        tags[slot] = pool_next_tag(node);
      }
    } else {
      // ================= shade: one bounce of raycolor (K:807-976) for every path of the batch whose walk has ended; paths that end
      // store their pixel (K:1081-1085) and the slot takes the next pixel (camera ray K:1065-1073)
      float4* const g = gunits + sidx * POOL_GUNITS;               // RNG, attenuation, pixel: this wave's array in global memory
      Path path; path.rayo = mk(0, 0, 0); path.raydir = mk(0, 0, 0); path.atten = mk(0, 0, 0);
      Xorwow rng; rng.v0 = rng.v1 = rng.v2 = rng.v3 = rng.v4 = rng.d = 0;
      int px = -1, py = 0, pcode = 0, frame = 0, bounce = 0;
      unsigned steps = 0;
      bool want_pixel = false, alive = false;
      if (slot >= 0) {
        const float4 A4 = g[0], A5 = g[1], A6 = g[2];
        const unsigned meta = __float_as_uint(A6.z);
        if (meta & POOL_META_NEW) {
          want_pixel = true;
        } else {
          const float4 A0 = *reinterpret_cast<const float4*>(u0), A1 = *reinterpret_cast<const float4*>(u1), A3 = *reinterpret_cast<const float4*>(u3);
          path.rayo = mk(A0.x, A0.y, A0.z); path.raydir = mk(A1.x, A1.y, A1.z); path.atten = mk(A5.z, A5.w, A6.x);
          const float best_t = A0.w; const int best_slot = __float_as_int(A1.w);
          rng.v0 = __float_as_uint(A4.x); rng.v1 = __float_as_uint(A4.y); rng.v2 = __float_as_uint(A4.z); rng.v3 = __float_as_uint(A4.w);
          rng.v4 = __float_as_uint(A5.x); rng.d = __float_as_uint(A5.y);
          steps = __float_as_uint(A3.z); pcode = __float_as_int(A3.w);
          { const int xy = __float_as_int(A6.y); px = (int)((unsigned)xy >> 16) - 1; py = xy & 0xffff; }
          frame = (int)(meta & 255u); bounce = (int)(meta >> 8);
          bool ended = false;
          V3 radiance = mk(0, 0, 0);
          if (best_slot >= 0 && best_t > 0.0f) {
            ended = !shade_hit<false>(P, path, best_t, best_slot, rng, c, radiance);
            if (!ended) {
              bounce++;
              if (bounce >= P.max_depth) ended = true;        // depth exhausted: black (K:981)
            }
          } else {
            radiance = shade_miss<false>(P, path, c);
            ended = true;
          }
          if (ended) {
            store_pixel(P, px, py, mk(0, 0, 0) + radiance);   // one sample per pixel (pool_kernel_can_render): colour = 0 + radiance (K:1059-1062)
            if (pixel_cost) pixel_cost[pcode] = steps;
            px = -1;
            want_pixel = true;
          } else {
            alive = true;
          }
        }
      }
      // ---- next pixels.  A region's work is one sequence of pixel positions, 64 per (tile, frame) chunk: position p is pixel p & 63 of
      // tile order[(p >> 6) / batch] in frame (p >> 6) % batch.  A batch takes as many positions as it has paths without a pixel with
      // ONE atomic add
      unsigned long long need = __ballot(want_pixel);
      while (need != 0ull && regions_left > 0) {
        const int r0 = region_start ? region_start[region] : P.region_start[region];
        const int r1e = region_start ? region_start[region + 1] : P.region_start[region + 1];
        const unsigned limit = (unsigned)(r1e - r0) * (unsigned)P.batch * 64u;
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(tile_counter + region, (unsigned)__popcll(need));
        const unsigned p = (unsigned)__builtin_amdgcn_readfirstlane((int)t) + pool_lane_rank(need);
        if (want_pixel && p < limit) {
          const unsigned q = p >> 6, l = p & 63u;
          const unsigned tt = q / (unsigned)P.batch;
          frame = (int)(q - tt * (unsigned)P.batch);
          const int tile = tile_order ? tile_order[r0 + (int)tt] : r0 + (int)tt;
          const int col = tile / P.gy, by = tile - col * P.gy;
          px = (P.stripe_rem + col * P.stripe_mod) * 8 + (int)(l >> 3);
          py = by * 8 + (int)(l & 7u);
          pcode = tile * 64 + (int)l;
          want_pixel = false;
        }
        need = __ballot(want_pixel);
        if (need != 0ull) {                                      // this band is done: help with the next one
          region = region + 1 == P.regions ? 0 : region + 1;
          regions_left--;
        }
      }
      if (slot >= 0 && want_pixel) tags[slot] = PT_DEAD;         // no pixels left
      if (slot >= 0 && !alive && !want_pixel) {
        // a new pixel: its one sample (K:1059-1073)
        rng.init(sample_seed(P, px, py, 0, frame));
        camera_ray(P, px, py, rng, path.rayo, path.raydir);
        path.atten = splat(1.0f);
        bounce = 0; steps = 0;
        alive = true;
      }
      if (slot >= 0 && alive) {
        const V3 inv = mk(1.0f / path.raydir.x, 1.0f / path.raydir.y, 1.0f / path.raydir.z);
        *reinterpret_cast<float4*>(u0) = make_float4(path.rayo.x, path.rayo.y, path.rayo.z, 10000.0f);                 // trav_begin
        *reinterpret_cast<float4*>(u1) = make_float4(path.raydir.x, path.raydir.y, path.raydir.z, __int_as_float(-1));
        *reinterpret_cast<float4*>(u2) = make_float4(inv.x, inv.y, inv.z, __int_as_float(0));
        *reinterpret_cast<float4*>(u3) = make_float4(__uint_as_float(0u), __int_as_float(0), __uint_as_float(steps), __int_as_float(pcode));
        g[0] = make_float4(__uint_as_float(rng.v0), __uint_as_float(rng.v1), __uint_as_float(rng.v2), __uint_as_float(rng.v3));
        g[1] = make_float4(__uint_as_float(rng.v4), __uint_as_float(rng.d), path.atten.x, path.atten.y);
        g[2] = make_float4(path.atten.z, __int_as_float(((px + 1) << 16) | py), __uint_as_float((unsigned)frame | ((unsigned)bounce << 8)), 0.0f);
        tags[slot] = PT_NODE;
      }
    }
    if (DIAG) {
      const unsigned long long d_t2 = __builtin_readcyclecounter();
      if (stage == PT_NODE) { d_t_node += d_t2 - d_t1; d_t_nstate += d_ta - d_t1; d_t_nfetch += d_tb - d_ta; } else if (stage == PT_LEAF) d_t_leaf += d_t2 - d_t1; else d_t_shade += d_t2 - d_t1;
    }
  }
  if (DIAG && lane == 0) {
    unsigned long long* const d = P.counters + 16;
    atomicAdd(&d[0], (unsigned long long)d_b_node); atomicAdd(&d[1], (unsigned long long)d_b_leaf); atomicAdd(&d[2], (unsigned long long)d_b_shade);
    atomicAdd(&d[3], (unsigned long long)d_l_node); atomicAdd(&d[4], (unsigned long long)d_l_leaf); atomicAdd(&d[5], (unsigned long long)d_l_shade);
    atomicAdd(&d[6], d_t_sel); atomicAdd(&d[7], d_t_node); atomicAdd(&d[8], d_t_leaf); atomicAdd(&d[9], d_t_shade);
    atomicAdd(&d[10], d_t_nstate); atomicAdd(&d[11], d_t_nfetch);
  }
  if (lane == 0) {
    atomicAdd(&P.counters[8], __builtin_readcyclecounter() - t_begin);
    atomicAdd(&P.counters[15], __builtin_amdgcn_s_memrealtime() - r_begin);      // 100 MHz ticks: wave cycles / this = shader clock / 100 MHz
  }
}

// ------------------------------------------------------------------ launcher
bool pool_kernel_can_render(const RenderParams& P) {
  // one sample per pixel and frame (the progressive loop's setting; more samples per pixel go to the persistent kernel), something to trace
  return P.wide != nullptr && P.max_depth > 0 && P.max_depth < (1 << 22) && P.spp_f > 0.0f && P.spp_f <= 1.0f && P.batch >= 1 && P.batch <= 256;
}

// the shapes built: stack words in LDS / waves per workgroup / workgroups per CU (the LDS of a CU: 160 KiB) / waves per SIMD
struct PoolShape { int lstack, wg_waves, wgs_per_cu; };
static const PoolShape pool_shapes[] = {{4, 5, 3}, {3, 4, 4}, {8, 4, 3}};
static_assert(3 * 5 * pool_wave_lds(4) <= 160 * 1024 && 4 * 4 * pool_wave_lds(3) <= 160 * 1024 && 3 * 4 * pool_wave_lds(8) <= 160 * 1024, "LDS of a CU");
static inline const PoolShape& pool_shape(int k) { return pool_shapes[k < 0 || k > 2 ? 0 : k]; }

size_t pool_scratch_words(int num_cus) {      // enough for every shape
  size_t most = 0;
  for (const PoolShape& sh : pool_shapes) {
    const size_t w = (size_t)num_cus * sh.wgs_per_cu * sh.wg_waves * (size_t)pool_wave_scratch(sh.lstack);
    if (w > most) most = w;
  }
  return most;
}

void launch_pool_kernel(hipStream_t stream, const RenderParams& P, const PoolCfg& cfg, unsigned* tile_counter, const int* order, const int* region_start,
                        unsigned* pixel_cost, unsigned* scratch) {
  const PoolShape& sh = pool_shape(cfg.shape);
  const long long work = (long long)P.ncols * P.gy * P.batch;      // chunks of 64 pixels
  const long long per_block = sh.wg_waves * (POOL_SLOTS / 64);
  long long blocks = (long long)cfg.num_cus * sh.wgs_per_cu;
  if (blocks * per_block > work) blocks = (work + per_block - 1) / per_block;
  if (blocks < 1) blocks = 1;
#define DR_POOL_LAUNCH(DIAG, LSTACK, WGW, WPS)                                                                                                   \
  hipLaunchKernelGGL((render_pool_kernel<DIAG, LSTACK, WGW, WPS>), dim3((unsigned)blocks), dim3(WGW * 64), 0, stream, P, tile_counter, order, region_start, \
                     pixel_cost, scratch, cfg.shade_min)
  if (cfg.shape == 1) { if (cfg.diag) DR_POOL_LAUNCH(true, 3, 4, 4); else DR_POOL_LAUNCH(false, 3, 4, 4); }
  else if (cfg.shape == 2) { if (cfg.diag) DR_POOL_LAUNCH(true, 8, 4, 3); else DR_POOL_LAUNCH(false, 8, 4, 3); }
  else { if (cfg.diag) DR_POOL_LAUNCH(true, 4, 5, 4); else DR_POOL_LAUNCH(false, 4, 5, 4); }
#undef DR_POOL_LAUNCH
}

#else   // product build

bool pool_kernel_can_render(const RenderParams&) { return false; }
size_t pool_scratch_words(int) { return 0; }
void launch_pool_kernel(hipStream_t, const RenderParams&, const PoolCfg&, unsigned*, const int*, const int*, unsigned*, unsigned*) {}

#endif
}  // namespace dr
