// The pool kernel: the megakernel (Kernel / raycolor / hit, kernel.cu K:468-512, K:787-982, K:998-1093) with its three kinds of
// work -- node steps, leaf steps, shading -- each run at (nearly) full wave width.
//
// In the persistent kernel (kernels_render.hip) a path belongs to a lane, and a wave's lanes are spread over the three kinds:
// measured on the bench scene a node step executes for 30 of 64 lanes, a leaf step for 23, a shade phase for 37 (lane use 0.36,
// profiles/r2_b).  Here a path belongs to nobody.  One workgroup per CU owns POOL_SLOTS paths whose whole state lives in LDS
// (ray, best hit, next record, stack, RNG, attenuation, pixel: 144 bytes each), and three queues in LDS -- NODE, LEAF, SHADE -- hold
// the slot numbers of the paths that wait for that kind of step.  A wave takes up to 64 slots of ONE queue, loads their state, does
// that one step for all of them, stores the state and pushes every slot to the queue of its NEXT step.  Which wave or lane
// carries a path does not enter its arithmetic (same device functions, same order of draws; the seed is a function of x, y
// and the frame, K:1065): frames are bit-identical to the other kernels'.
//
// Queues.  Per queue a ring of POOL_RING 16-bit entries and two counters that only grow: `wr` (entries reserved by producers)
// and `rd` (entries claimed by consumers).  A producer wave reserves n positions with one atomic add and then writes its slot
// numbers; a consumer wave claims min(64, wr - rd) positions with one compare-and-swap on `rd` and then reads them.  An entry
// is EMPTY until it is written and is set back to EMPTY by the lane that takes it, so a consumer that claimed a position whose
// producer has not written yet spins on that entry (a handful of cycles: the write follows the reservation at once), and a
// producer never overwrites an entry that has not been taken (at most POOL_SLOTS < POOL_RING slot numbers exist).  The state of
// a path is written before its slot number (LDS operations of one wave execute in order; release / acquire fences keep the
// compiler from reordering).  Every wait is bounded: a wave that spins too long raises the abort flag, everybody leaves, and the
// host fails the call (dr_context: check_abort) -- a protocol bug must not hang the GPU.
//
// A wave picks the queue with the most waiting paths and takes a batch once POOL min_fill of them wait (or, after a few tries,
// whatever is there: progress is guaranteed); with 12 waves holding at most 768 of the 1 024 paths, a quarter of the pool waits
// in the queues and batches are full.  New pixels come from the same per-XCD tile queues, tile order and frame batching as the
// persistent kernel's (a SHADE batch whose paths ended takes the next tiles); a path that finds no pixel left dies, and the
// workgroup leaves when none lives.
//
// The stack of a path is POOL_LSTACK words in LDS; deeper words (the host bounds the depth at WIDE_STACK) go to a scratch array in
// global memory -- rare, and it lets 1 024 paths fit the CU's 160 KiB.
#include <hip/hip_runtime.h>

#include "device_core.hpp"
#include "kernels.hpp"
#include "../../include/dogeray_amd.h"

namespace dr {

constexpr int POOL_WAVES = 12;                  // waves per workgroup (3 per SIMD)
constexpr int POOL_SLOTS = 1024;                // paths per workgroup
constexpr int POOL_LSTACK = 8;                  // stack words per path kept in LDS
constexpr int POOL_RING = 2048;                 // entries per queue ring
constexpr int POOL_UNITS = 7;                   // 16-byte state units per path
constexpr unsigned POOL_EMPTY = 0xffffu;
constexpr unsigned POOL_SPIN_LIMIT = 1u << 22;  // bounded waits (a correct run spins a few times)
constexpr int POOL_TRIES = 6;                   // polls of the queues before a wave settles for a batch below min_fill
enum { PQ_NODE = 0, PQ_LEAF = 1, PQ_SHADE = 2 };
// counters in LDS
enum { PC_WR = 0, PC_RD = 4, PC_LIVE = 8, PC_DRAIN = 9, PC_ABORT = 10, PC_WORDS = 16 };
// state units: U0 {o.xyz, best t} U1 {d.xyz, best slot} U2 {1/d.xyz, next record} U3 {stack top, stack pointer, steps, pixel code}
//              U4 {rng v0..v3} U5 {rng v4, rng d, atten.xy} U6 {atten.z, (x + 1) << 16 | y, frame | bounce << 8, -}
constexpr unsigned POOL_META_NEW = 0x80000000u; // the slot holds no path yet (or its path has ended): it needs a pixel

static_assert(POOL_UNITS * POOL_SLOTS * 16 + POOL_LSTACK * POOL_SLOTS * 4 + 3 * POOL_RING * 2 + PC_WORDS * 4 <= 160 * 1024, "LDS of one CU");
static_assert(POOL_RING > POOL_SLOTS && (POOL_RING & (POOL_RING - 1)) == 0, "ring size");
static_assert(POOL_LSTACK <= WIDE_STACK, "stack split");

struct PoolLds {
  float4* units;              // unit u of slot s at units[u * POOL_SLOTS + s]
  int* stackw;                // word k of slot s at stackw[k * POOL_SLOTS + s]
  unsigned short* ring;       // queue q at ring[q * POOL_RING ...]
  unsigned* ctr;
};

// the per-path stack: words [0, POOL_LSTACK) in LDS, deeper ones in the workgroup's global scratch (read and written past the L1:
// another wave of the workgroup may continue the path)
struct PoolStack {
  int* lds; unsigned* glob;   // both already offset by the slot
  __device__ __forceinline__ int ld(int k) const {
    if (k < POOL_LSTACK) return lds[k * POOL_SLOTS];
    return (int)__hip_atomic_load(glob + (k - POOL_LSTACK) * POOL_SLOTS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void st(int k, int v) const {
    if (k < POOL_LSTACK) lds[k * POOL_SLOTS] = v;
    else __hip_atomic_store(glob + (k - POOL_LSTACK) * POOL_SLOTS, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

// wide_pop (device_core.hpp) on a PoolStack
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

// pushes the slots of the lanes with `flag` to queue q; false if a wait ran out (protocol failure)
__device__ __forceinline__ bool pool_push(const PoolLds& L, int q, bool flag, int slot) {
  const unsigned long long m = __ballot(flag);
  if (m == 0ull) return true;
  const unsigned n = (unsigned)__popcll(m);
  unsigned base = 0;
  if (__lane_id() == (unsigned)(__ffsll((long long)m) - 1)) base = __hip_atomic_fetch_add(&L.ctr[PC_WR + q], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  base = (unsigned)__builtin_amdgcn_readlane((int)base, __ffsll((long long)m) - 1);
  bool ok = true;
  if (flag) {
    volatile unsigned short* e = L.ring + q * POOL_RING + ((base + pool_lane_rank(m)) & (POOL_RING - 1));
    unsigned spins = 0;
    while (*e != POOL_EMPTY) { if (++spins > POOL_SPIN_LIMIT) { ok = false; break; } __builtin_amdgcn_s_sleep(1); }
    *e = (unsigned short)slot;
  }
  return __ballot(!ok) == 0ull;
}

template <bool DIAG>
__global__ __launch_bounds__(POOL_WAVES * 64) void render_pool_kernel(RenderParams P, unsigned* __restrict__ tile_counter, const int* __restrict__ tile_order,
                                                                      const int* __restrict__ region_start, unsigned* __restrict__ pixel_cost,
                                                                      unsigned* __restrict__ scratch, unsigned* __restrict__ abort_flag, int min_fill) {
  __shared__ __attribute__((aligned(16))) float4 lds_units[POOL_UNITS * POOL_SLOTS];
  __shared__ int lds_stack[POOL_LSTACK * POOL_SLOTS];
  __shared__ unsigned short lds_ring[3 * POOL_RING];
  __shared__ unsigned lds_ctr[PC_WORDS];
  PoolLds L; L.units = lds_units; L.stackw = lds_stack; L.ring = lds_ring; L.ctr = lds_ctr;
  const int lane = threadIdx.x & 63;
  unsigned* const my_scratch = scratch + (size_t)blockIdx.x * (size_t)((WIDE_STACK - POOL_LSTACK) * POOL_SLOTS);
  const WalkRsrc walk = wide_rsrc(P);
  const unsigned long long t_begin = __builtin_readcyclecounter(), r_begin = __builtin_amdgcn_s_memrealtime();

  // ---- every slot starts without a path, waiting in the SHADE queue for a pixel
  for (int i = threadIdx.x; i < 3 * POOL_RING; i += POOL_WAVES * 64) lds_ring[i] = (unsigned short)((i >= PQ_SHADE * POOL_RING && i < PQ_SHADE * POOL_RING + POOL_SLOTS) ? i - PQ_SHADE * POOL_RING : POOL_EMPTY);
  for (int i = threadIdx.x; i < POOL_SLOTS; i += POOL_WAVES * 64) {
    lds_units[3 * POOL_SLOTS + i] = make_float4(0, 0, 0, 0);
    lds_units[6 * POOL_SLOTS + i] = make_float4(0.0f, 0.0f, __uint_as_float(POOL_META_NEW), 0.0f);
  }
  if (threadIdx.x < PC_WORDS) lds_ctr[threadIdx.x] = threadIdx.x == PC_WR + PQ_SHADE ? (unsigned)POOL_SLOTS : (threadIdx.x == PC_LIVE ? (unsigned)POOL_SLOTS : 0u);
  __syncthreads();

  // work queues as in the persistent kernel (tile order of the cost feedback, frames of a batch interleaved, one queue per region =
  // XCD; a wave helps the next region once its own is empty), but counted in pixels
  int region = 0, regions_left = P.regions;
  if (P.regions > 1) region = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) % (unsigned)P.regions);   // HW_REG_XCC_ID
  Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
  bool failed = false;
  // DIAG build (option pool_diag): batches and paths per stage, shader cycles spent choosing / in the step / pushing, retries
  unsigned long long d_t_sel = 0, d_t_node = 0, d_t_leaf = 0, d_t_shade = 0, d_t_push = 0;
  unsigned d_b_node = 0, d_b_leaf = 0, d_b_shade = 0, d_l_node = 0, d_l_leaf = 0, d_l_shade = 0, d_tries = 0, d_casfail = 0, d_polls = 0;

  for (;;) {
    unsigned long long d_t0 = 0;
    if (DIAG) d_t0 = __builtin_readcyclecounter();
    // ---- pick a queue and claim a batch
    int stage = -1, n = 0; unsigned base = 0;
    int tries = 0; unsigned polls = 0;
    for (;;) {
      unsigned v = 0;
      if (lane < PC_WORDS) v = __hip_atomic_load(&L.ctr[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // (rd is read by the same instruction as wr: a claim in between only makes `avail` an under-estimate... the compare-and-swap decides)
      const unsigned rd0 = (unsigned)__builtin_amdgcn_readlane((int)v, PC_RD + 0), rd1 = (unsigned)__builtin_amdgcn_readlane((int)v, PC_RD + 1), rd2 = (unsigned)__builtin_amdgcn_readlane((int)v, PC_RD + 2);
      const int a0 = (int)((unsigned)__builtin_amdgcn_readlane((int)v, PC_WR + 0) - rd0), a1 = (int)((unsigned)__builtin_amdgcn_readlane((int)v, PC_WR + 1) - rd1),
                a2 = (int)((unsigned)__builtin_amdgcn_readlane((int)v, PC_WR + 2) - rd2);
      const unsigned live = (unsigned)__builtin_amdgcn_readlane((int)v, PC_LIVE);
      const bool drain = __builtin_amdgcn_readlane((int)v, PC_DRAIN) != 0;
      if (__builtin_amdgcn_readlane((int)v, PC_ABORT) != 0) { failed = true; break; }
      int s = a0 >= a1 ? 0 : 1; int best = a0 >= a1 ? a0 : a1; unsigned rds = a0 >= a1 ? rd0 : rd1;
      if (a2 > best) { s = 2; best = a2; rds = rd2; }
      if (best <= 0) {
        if (live == 0u) break;                                   // nothing waits and nothing lives: done
        if (++polls > POOL_SPIN_LIMIT) { failed = true; break; }
        if (DIAG) d_polls++;
        __builtin_amdgcn_s_sleep(8);
        continue;
      }
      if (best < min_fill && !drain && tries < POOL_TRIES) { tries++; if (DIAG) d_tries++; __builtin_amdgcn_s_sleep(16); continue; }
      const int take = best < 64 ? best : 64;
      unsigned got = 0;
      if (lane == 0) {
        unsigned expect = rds;
        got = __hip_atomic_compare_exchange_strong(&L.ctr[PC_RD + s], &expect, rds + (unsigned)take, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
      }
      if (__builtin_amdgcn_readfirstlane((int)got) != 0) { stage = s; n = take; base = rds; break; }
      if (DIAG) d_casfail++;
    }
    if (stage < 0) break;
    int slot = -1;
    if (lane < n) {
      volatile unsigned short* e = L.ring + stage * POOL_RING + ((base + (unsigned)lane) & (POOL_RING - 1));
      unsigned spins = 0, got = POOL_EMPTY;
      while ((got = *e) == POOL_EMPTY) { if (++spins > POOL_SPIN_LIMIT) break; __builtin_amdgcn_s_sleep(1); }
      *e = (unsigned short)POOL_EMPTY;
      slot = got == POOL_EMPTY ? -2 : (int)got;
    }
    if (__ballot(slot == -2) != 0ull) { failed = true; break; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    unsigned long long d_t1 = 0;
    if (DIAG) {
      d_t1 = __builtin_readcyclecounter(); d_t_sel += d_t1 - d_t0;
      if (stage == PQ_NODE) { d_b_node++; d_l_node += (unsigned)n; } else if (stage == PQ_LEAF) { d_b_leaf++; d_l_leaf += (unsigned)n; } else { d_b_shade++; d_l_shade += (unsigned)n; }
    }
    const int sidx = slot < 0 ? 0 : slot;                        // inactive lanes address slot 0 and touch nothing
    float* const u0 = reinterpret_cast<float*>(L.units + 0 * POOL_SLOTS + sidx);
    float* const u1 = reinterpret_cast<float*>(L.units + 1 * POOL_SLOTS + sidx);
    float* const u2 = reinterpret_cast<float*>(L.units + 2 * POOL_SLOTS + sidx);
    float* const u3 = reinterpret_cast<float*>(L.units + 3 * POOL_SLOTS + sidx);
    PoolStack stk; stk.lds = L.stackw + sidx; stk.glob = my_scratch + sidx;
    bool to_node = false, to_leaf = false, to_shade = false;
    bool deep = false;                                           // this lane may have written stack words to the global scratch

    if (stage == PQ_NODE) {
      // ================= node step: wide_node_compute (device_core.hpp) on the path's state
      if (slot >= 0) {
        const float4 A0 = *reinterpret_cast<const float4*>(u0), A2 = *reinterpret_cast<const float4*>(u2);
        const float4 A3 = *reinterpret_cast<const float4*>(u3);
        int node = __float_as_int(A2.w);
        const WideRec r = wide_fetch(walk, node);
        const V3 o = mk(A0.x, A0.y, A0.z), inv = mk(A2.x, A2.y, A2.z);
        const WideRay wr = wide_ray(o, inv, P.wide_pmax);
        unsigned top = __float_as_uint(A3.x); int sp = __float_as_int(A3.y);
        unsigned key;
        const unsigned mask = wide_node_test(r.A, r.B, r.C, r.D, o, DR_WIDE_FOLD ? wr.inv : inv, wr.marg, A0.w, key);
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
        to_node = node >= 0 && !(node & 1); to_leaf = node >= 0 && (node & 1); to_shade = node < 0;
        deep = sp > POOL_LSTACK;
      }
    } else if (stage == PQ_LEAF) {
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
        to_node = node >= 0 && !(node & 1); to_leaf = node >= 0 && (node & 1); to_shade = node < 0;
      }
    } else {
      // ================= shade: one bounce of raycolor (K:807-976) for every path of the batch whose walk has ended; paths that end
      // store their pixel (K:1081-1085) and the slot takes the next pixel (camera ray K:1065-1073)
      float* const u4 = reinterpret_cast<float*>(L.units + 4 * POOL_SLOTS + sidx);
      float* const u5 = reinterpret_cast<float*>(L.units + 5 * POOL_SLOTS + sidx);
      float* const u6 = reinterpret_cast<float*>(L.units + 6 * POOL_SLOTS + sidx);
      Path path; path.rayo = mk(0, 0, 0); path.raydir = mk(0, 0, 0); path.atten = mk(0, 0, 0);
      Xorwow rng; rng.v0 = rng.v1 = rng.v2 = rng.v3 = rng.v4 = rng.d = 0;
      int px = -1, py = 0, pcode = 0, frame = 0, bounce = 0;
      unsigned steps = 0;
      bool want_pixel = false, alive = false;
      if (slot >= 0) {
        const float4 A6 = *reinterpret_cast<const float4*>(u6);
        const unsigned meta = __float_as_uint(A6.z);
        if (meta & POOL_META_NEW) {
          want_pixel = true;
        } else {
          const float4 A0 = *reinterpret_cast<const float4*>(u0), A1 = *reinterpret_cast<const float4*>(u1), A3 = *reinterpret_cast<const float4*>(u3);
          const float4 A4 = *reinterpret_cast<const float4*>(u4), A5 = *reinterpret_cast<const float4*>(u5);
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
      // ONE atomic add (the persistent kernel hands out whole tiles to a wave, whose lanes come back to it; here no wave owns anything,
      // so nothing may be left over in a wave)
      unsigned long long need = __ballot(want_pixel);
      while (need != 0ull && regions_left > 0) {
        const int r0 = region_start ? region_start[region] : P.region_start[region];
        const int r1 = region_start ? region_start[region + 1] : P.region_start[region + 1];
        const unsigned limit = (unsigned)(r1 - r0) * (unsigned)P.batch * 64u;
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
      if (need != 0ull && lane == 0) __hip_atomic_store(&L.ctr[PC_DRAIN], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // no pixels left: the slots still asking die
      const unsigned long long dying = __ballot(slot >= 0 && want_pixel);      // asked and got nothing
      if (dying != 0ull && lane == 0) __hip_atomic_fetch_sub(&L.ctr[PC_LIVE], (unsigned)__popcll(dying), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
        *reinterpret_cast<float4*>(u0) = make_float4(path.rayo.x, path.rayo.y, path.rayo.z, 10000000.0f);                 // trav_begin
        *reinterpret_cast<float4*>(u1) = make_float4(path.raydir.x, path.raydir.y, path.raydir.z, __int_as_float(-1));
        *reinterpret_cast<float4*>(u2) = make_float4(inv.x, inv.y, inv.z, __int_as_float(0));
        *reinterpret_cast<float4*>(u3) = make_float4(__uint_as_float(0u), __int_as_float(0), __uint_as_float(steps), __int_as_float(pcode));
        *reinterpret_cast<float4*>(u4) = make_float4(__uint_as_float(rng.v0), __uint_as_float(rng.v1), __uint_as_float(rng.v2), __uint_as_float(rng.v3));
        *reinterpret_cast<float4*>(u5) = make_float4(__uint_as_float(rng.v4), __uint_as_float(rng.d), path.atten.x, path.atten.y);
        *reinterpret_cast<float4*>(u6) = make_float4(path.atten.z, __int_as_float(((px + 1) << 16) | py), __uint_as_float((unsigned)frame | ((unsigned)bounce << 8)), 0.0f);
        to_node = true;
      }
    }
    // ---- every path of the batch goes to the queue of its next step (its state first)
    // (stack words that went to global memory must have arrived before another wave can take the path: rare, so the wider fence --
    // it waits for every outstanding store and pixel atomic of the wave -- is paid only then)
    if (__ballot(deep) != 0ull) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    unsigned long long d_t2 = 0;
    if (DIAG) {
      d_t2 = __builtin_readcyclecounter();
      if (stage == PQ_NODE) d_t_node += d_t2 - d_t1; else if (stage == PQ_LEAF) d_t_leaf += d_t2 - d_t1; else d_t_shade += d_t2 - d_t1;
    }
    bool ok = pool_push(L, PQ_NODE, to_node, slot);
    ok = pool_push(L, PQ_LEAF, to_leaf, slot) && ok;
    ok = pool_push(L, PQ_SHADE, to_shade, slot) && ok;
    if (DIAG) d_t_push += __builtin_readcyclecounter() - d_t2;
    if (!ok) { failed = true; break; }
  }
  if (DIAG && lane == 0) {
    unsigned long long* const d = P.counters + 16;
    atomicAdd(&d[0], (unsigned long long)d_b_node); atomicAdd(&d[1], (unsigned long long)d_b_leaf); atomicAdd(&d[2], (unsigned long long)d_b_shade);
    atomicAdd(&d[3], (unsigned long long)d_l_node); atomicAdd(&d[4], (unsigned long long)d_l_leaf); atomicAdd(&d[5], (unsigned long long)d_l_shade);
    atomicAdd(&d[6], d_t_sel); atomicAdd(&d[7], d_t_node); atomicAdd(&d[8], d_t_leaf); atomicAdd(&d[9], d_t_shade); atomicAdd(&d[10], d_t_push);
    atomicAdd(&d[11], (unsigned long long)d_tries); atomicAdd(&d[12], (unsigned long long)d_casfail); atomicAdd(&d[13], (unsigned long long)d_polls);
  }
  if (failed) {
    if (lane == 0) { __hip_atomic_store(&L.ctr[PC_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicExch(abort_flag, 1u); }
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

size_t pool_scratch_words(int num_cus) { return (size_t)num_cus * (size_t)((WIDE_STACK - POOL_LSTACK) * POOL_SLOTS); }

void launch_pool_kernel(hipStream_t stream, const RenderParams& P, const PoolCfg& cfg, unsigned* tile_counter, const int* order, const int* region_start,
                        unsigned* pixel_cost, unsigned* scratch, unsigned* abort_flag) {
  const long long work = (long long)P.ncols * P.gy * P.batch;
  long long blocks = cfg.num_cus;                                   // one workgroup per CU (its LDS)
  if (blocks * (POOL_SLOTS / 64) > work) blocks = (work + POOL_SLOTS / 64 - 1) / (POOL_SLOTS / 64);
  if (blocks < 1) blocks = 1;
  if (cfg.diag) hipLaunchKernelGGL((render_pool_kernel<true>), dim3((unsigned)blocks), dim3(POOL_WAVES * 64), 0, stream, P, tile_counter, order, region_start, pixel_cost,
                                   scratch, abort_flag, cfg.min_fill);
  else hipLaunchKernelGGL((render_pool_kernel<false>), dim3((unsigned)blocks), dim3(POOL_WAVES * 64), 0, stream, P, tile_counter, order, region_start, pixel_cost,
                          scratch, abort_flag, cfg.min_fill);
}

}  // namespace dr
