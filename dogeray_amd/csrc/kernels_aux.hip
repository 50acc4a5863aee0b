// Everything on the device that is not a render kernel: tile-order feedback of the persistent kernels, the display divide,
// the stripe copies of the multi-GPU gather, the gather-ceiling probe and the known-answer kernels behind dr_kat_*.
#include <hip/hip_runtime.h>

#include "device_core.hpp"
#include "kernels.hpp"
#include "../../include/dogeray_amd.h"

namespace dr {

// Cost feedback for the persistent kernel: per-tile cost = the most node steps any of its pixels
// took (the critical path of the tile), then tiles sorted by cost, most expensive first.
__global__ __launch_bounds__(256) void tile_cost_kernel(const unsigned* __restrict__ pixel_cost, unsigned* __restrict__ tile_cost, int ntiles) {
  int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  unsigned v = pixel_cost[(size_t)tile * 64 + (threadIdx.x & 63)];
  for (int off = 32; off > 0; off >>= 1) { unsigned o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
  if ((threadIdx.x & 63) == 0) tile_cost[tile] = v;
}
// One workgroup builds the next launch's tile order.  Per region (band of the tile numbering): first the
// EXPENSIVE tiles (cost above `heavy_factor` x the mean), most expensive first -- they set the length of a
// launch, so they start first; then all other tiles in their natural order, so that the waves of an XCD walk
// their band coherently (neighbouring tiles see neighbouring parts of the scene).
constexpr int ORDER_BUCKETS = 256;
// exclusive prefix sum over the threads of a 1 024-thread block (all of them call it); total = the sum over the block
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned& total) {
  __shared__ unsigned wave_sum[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned x = v;
  for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
  if (lane == 63) wave_sum[w] = x;
  __syncthreads();
  if (w == 0) {
    unsigned sw = lane < 16 ? wave_sum[lane] : 0u;
    for (int off = 1; off < 16; off <<= 1) { const unsigned y = __shfl_up(sw, off, 64); if (lane >= off) sw += y; }
    if (lane < 16) wave_sum[lane] = sw;                  // inclusive over the waves
  }
  __syncthreads();
  const unsigned before = w > 0 ? wave_sum[w - 1] : 0u;
  total = wave_sum[15];
  __syncthreads();                                       // wave_sum may be reused by the next call
  return before + x - v;
}
__global__ __launch_bounds__(1024) void tile_order_kernel(const unsigned* __restrict__ tile_cost, int* __restrict__ order,
                                                           int* __restrict__ region_start, int ntiles, int regions, int heavy_factor, int split_steps, int split_limit) {
  constexpr int WAVES = 16, GROUPS = 8;                    // 1 024 threads; a wave reads GROUPS x 64 costs per round trip
  __shared__ unsigned hist[MAX_REGIONS * ORDER_BUCKETS];   // expensive tiles per (region, cost class)
  __shared__ unsigned base[MAX_REGIONS * ORDER_BUCKETS];
  __shared__ unsigned light_cnt[WAVES][MAX_REGIONS];       // light tiles of a wave's range per region, then: of the waves before it
  __shared__ unsigned light_in_region[MAX_REGIONS], light_before[MAX_REGIONS], light_start[MAX_REGIONS], split_tiles[MAX_REGIONS];
  __shared__ int rb[MAX_REGIONS + 1];                      // first tile of region r (region_of(t) = t * regions / ntiles)
  __shared__ unsigned long long total_cost;
  const int nb = regions * ORDER_BUCKETS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < nb; i += blockDim.x) hist[i] = 0;
  if (tid < WAVES * MAX_REGIONS) (&light_cnt[0][0])[tid] = 0;
  if (tid < MAX_REGIONS) split_tiles[tid] = 0;
  if (tid <= regions) rb[tid] = (int)(((long long)tid * ntiles + regions - 1) / regions);
  if (tid == 0) total_cost = 0;
  __syncthreads();
  // Wave w owns the contiguous range [w0, w1) of the tile numbering and walks it 64 tiles at a time, lane l on tile g + l: the light
  // tiles keep their natural order through ballots and prefix counts (a range at a time, a region at a time), the expensive ones are
  // binned by cost class.  (One thread per 32-tile chunk with everything per tile took 70-100 us on the one CU this block has.)
  const int per = ((ntiles + WAVES - 1) / WAVES + 63) & ~63;
  const int w0 = w * per < ntiles ? w * per : ntiles, w1 = w0 + per < ntiles ? w0 + per : ntiles;
  auto mbcnt = [](unsigned long long m) { return (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
  // body(t, cost, valid, rcur): rcur = region of the 64-tile group's first tile (wave-uniform); a lane's own region is rcur or rcur + 1
  auto for_range = [&](auto&& body) {
    int rcur = 0;
    while (rcur + 1 < regions && w0 >= rb[rcur + 1]) rcur++;
    for (int g0 = w0; g0 < w1; g0 += 64 * GROUPS) {
      unsigned v[GROUPS];
#pragma unroll
      for (int k = 0; k < GROUPS; k++) { const int t = g0 + 64 * k + lane; v[k] = t < w1 ? tile_cost[t] : 0u; }
#pragma unroll
      for (int k = 0; k < GROUPS; k++) {
        const int g = g0 + 64 * k;
        if (g < w1) {
          while (rcur + 1 < regions && g >= rb[rcur + 1]) rcur++;
          body(g + lane, v[k], g + lane < w1, rcur);
        }
      }
    }
  };
  {
    unsigned long long sum = 0;
    for_range([&](int, unsigned cost, bool, int) { sum += cost; });
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) atomicAdd(&total_cost, sum);
  }
  __syncthreads();
  const unsigned long long threshold = heavy_factor > 0 ? (total_cost * (unsigned long long)heavy_factor) / (unsigned long long)(ntiles > 0 ? ntiles : 1)
                                                       : (heavy_factor < 0 ? 0ull : ~0ull);       // -1: every tile by cost, 0: all natural
  auto key_of = [&](unsigned cost, int r) {
    unsigned b = cost >> 4; if (b > ORDER_BUCKETS - 1) b = ORDER_BUCKETS - 1;
    return r * ORDER_BUCKETS + (ORDER_BUCKETS - 1 - (int)b);      // ascending key = region, then most expensive first
  };
  // count: expensive tiles per class; tiles whose longest pixel took at least split_steps node steps (a whole cost class: they are a
  // prefix of their region's order; launches of one frame hand them out in parts, render_persistent_kernel nsplit); light tiles per
  // (wave, region)
  for_range([&](int t, unsigned cost, bool valid, int rcur) {
    const int r = rcur + ((rcur + 1 < regions && t >= rb[rcur + 1]) ? 1 : 0);
    const bool heavy = valid && (unsigned long long)cost > threshold;
    if (heavy) {
      atomicAdd(&hist[key_of(cost, r)], 1u);
      if (split_steps > 0 && (cost >> 4) >= (unsigned)(split_steps >> 4)) atomicAdd(&split_tiles[r], 1u);
    }
    const bool light = valid && !heavy;
    const unsigned long long b0 = __ballot(light && r == rcur), b1 = __ballot(light && r != rcur);
    if (lane == 0) { light_cnt[w][rcur] += (unsigned)__popcll(b0); if (b1) light_cnt[w][rcur + 1] += (unsigned)__popcll(b1); }
  });
  __syncthreads();
  // positions: region r's expensive tiles by class, then its light tiles
  if (tid < regions) {
    unsigned run = 0;
    for (int k = 0; k < WAVES; k++) { const unsigned n = light_cnt[k][tid]; light_cnt[k][tid] = run; run += n; }      // now: light tiles of region tid in the waves before k
    light_in_region[tid] = run;
  }
  unsigned total_heavy;
  const int i0 = tid * 2;                                                            // nb <= 2 * blockDim.x
  const unsigned h0 = i0 < nb ? hist[i0] : 0u, h1 = i0 + 1 < nb ? hist[i0 + 1] : 0u;
  const unsigned heavy_before = block_exclusive_scan(h0 + h1, total_heavy);         // expensive tiles in the classes before i0 (syncs the block)
  if (tid == 0) {
    unsigned lights = 0;
    for (int r = 0; r < regions; r++) { light_before[r] = lights; lights += light_in_region[r]; }      // light tiles in the regions before r
  }
  __syncthreads();
  if (i0 < nb) base[i0] = heavy_before + light_before[i0 / ORDER_BUCKETS];
  if (i0 + 1 < nb) base[i0 + 1] = heavy_before + h0 + light_before[(i0 + 1) / ORDER_BUCKETS];
  __syncthreads();
  if (tid < regions) {
    region_start[tid] = (int)base[tid * ORDER_BUCKETS];
    const unsigned heavy_through = tid + 1 < regions ? base[(tid + 1) * ORDER_BUCKETS] - light_before[tid + 1] : total_heavy;
    light_start[tid] = heavy_through + light_before[tid];                            // where region r's light tiles begin
    const int limit = split_limit / regions;                                         // (any prefix of a region's order will do)
    region_start[MAX_REGIONS + 1 + tid] = (int)split_tiles[tid] < limit ? (int)split_tiles[tid] : limit;
  }
  if (tid == 0) region_start[regions] = ntiles;
  __syncthreads();
  // place
  int placed_r = -1; unsigned placed0 = 0, placed1 = 0;      // light tiles of this wave already placed in regions placed_r and placed_r + 1
  for_range([&](int t, unsigned cost, bool valid, int rcur) {
    if (rcur != placed_r) { placed0 = rcur == placed_r + 1 ? placed1 : 0u; placed1 = 0u; placed_r = rcur; }
    const int r = rcur + ((rcur + 1 < regions && t >= rb[rcur + 1]) ? 1 : 0);
    const bool heavy = valid && (unsigned long long)cost > threshold;
    if (heavy) order[atomicAdd(&base[key_of(cost, r)], 1u)] = t;
    const bool light = valid && !heavy;
    const unsigned long long b0 = __ballot(light && r == rcur), b1 = __ballot(light && r != rcur);
    if (light) {
      if (r == rcur) order[light_start[r] + light_cnt[w][r] + placed0 + mbcnt(b0)] = t;
      else order[light_start[r] + light_cnt[w][r] + placed1 + mbcnt(b1)] = t;
    }
    placed0 += (unsigned)__popcll(b0); placed1 += (unsigned)__popcll(b1);
  });
}

// clamp(acc / divide_by, 0, 255) into row-major RGB8 (draw loop K:2281-2287)
__global__ void present_kernel(const int32_t* acc, uint8_t* rgb, int W, int H, int div) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= W * H) return;
  int x = idx / H, y = idx - x * H;          // consecutive threads walk a column (coalesced read)
  const int32_t* p = acc + (size_t)idx * 3;
  uint8_t* q = rgb + ((size_t)y * W + x) * 3;
  for (int k = 0; k < 3; k++) {
    int v = p[k] / div;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    q[k] = (uint8_t)v;
  }
}

// Multi-GPU gather (K:1006: the framebuffer is column-major, so an 8-pixel block column is ONE contiguous run of
// 8*H*3 int32): copies `ncols` such runs between a strided position in a frame and a packed buffer, 16 bytes per lane.
//   pack    frame column rem + j*mod  ->  packed column j          (dr_accum_pack_stripe)
//   unpack  packed column j of rank r ->  frame column r + j*R     (dr_accum_unpack_stripes, on rank 0)
__global__ __launch_bounds__(256) void stripe_copy_kernel(int4* __restrict__ dst, const int4* __restrict__ src, int ncols, int run4,
                                                          long long dst_first4, long long dst_stride4, long long src_first4, long long src_stride4) {
  const int col = blockIdx.y;
  if (col >= ncols) return;
  int4* d = dst + dst_first4 + (long long)col * dst_stride4;
  const int4* sp = src + src_first4 + (long long)col * src_stride4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < run4; i += gridDim.x * blockDim.x) d[i] = sp[i];
}

// Ceiling probe for the measurement harness (bench.py `roofline.gather`): every lane fetches 64-byte records of the
// resident wide array at addresses that depend on what it fetched before -- the walk's memory behaviour without its
// arithmetic.  `nrec` restricts the walk to the first records of the array (a set that fits the L2s, or all of it).
__global__ __launch_bounds__(256, 5) void gather_probe_kernel(RenderParams P, unsigned nrec, int iters, unsigned* out) {
  const WalkRsrc r = wide_rsrc(P);
  unsigned x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  unsigned acc = 0;
  for (int i = 0; i < iters; i++) {
    const unsigned off = (x % nrec) << 6;
    const u32x4 a = ld_unit_raw(r, off), b = ld_unit_raw(r, off + 16), c = ld_unit_raw(r, off + 32), d = ld_unit_raw(r, off + 48);
    acc += a.x ^ b.y ^ c.z ^ d.w;
    x = x * 1664525u + 1013904223u + (acc & 1u);
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// ---- measurement aid: a TRACE-ONLY kernel over a list of rays (dr_context_probe_trace).  What would the walk cost if it were a kernel of its own -- a wave's
// lanes refilling from a global ray queue in a few instructions, no generator / attenuation / pixel state in its registers, shading done elsewhere?  Same node and
// leaf steps as the render kernel (exclusive steps, leaf steps once PARK lanes stand at a leaf), results {t bits, slot} per ray ({10000, -1}: no hit).
template <int OCC, int REFILL_MIN, int PARK>
__global__ __launch_bounds__(256, OCC) void trace_probe_kernel(RenderParams P, const float4* __restrict__ rays, unsigned n, unsigned* __restrict__ cursor, uint2* __restrict__ out) {
  __shared__ int lds[4 * WIDE_STACK * 64];
  const int lane = threadIdx.x & 63;
  int* const my_stack = lds + (threadIdx.x >> 6) * (WIDE_STACK * 64) + lane;
  const WalkRsrc walk = wide_rsrc(P);
  Trav tr; tr.node = -2; tr.best_t = 0; tr.best_slot = -1;      // -2: wants a ray; -1: done, result not written yet; -3: retired
  WideStack ws; ws.top = 0u; ws.sp = 0; ws.sb = 0;
  V3 o = mk(0, 0, 0), d = mk(0, 0, 0), inv = mk(0, 0, 0);
  WideRay wr = wide_ray_none();
  SignMask sg = sign_mask(inv);
  Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned idx = 0;
  unsigned chunk_next = 0, chunk_end = 0;      // this wave's share of the ray list (wave-uniform)
  bool exhausted = false;
  for (;;) {
    const unsigned long long want = __ballot(tr.node == -1 || tr.node == -2);
    const unsigned long long walking = __ballot(tr.node >= 0);
    if (want != 0ull && ((int)__popcll(want) >= REFILL_MIN || walking == 0ull)) {
      if (tr.node == -1) out[idx] = make_uint2(__float_as_uint(tr.best_t), (unsigned)tr.best_slot);
      const bool mine = tr.node == -1 || tr.node == -2;
      if (!exhausted) {
        const int cnt = (int)__popcll(want);
        // rays come in chunks of 128 per wave (one atomic on the shared cursor per chunk: a cursor touched at every refill is the bottleneck, 0.35 Grays/s)
        if (chunk_next + (unsigned)cnt > chunk_end && chunk_next >= chunk_end) {
          unsigned b = 0;
          if (lane == 0) b = atomicAdd(cursor, 128u);
          chunk_next = (unsigned)__builtin_amdgcn_readfirstlane((int)b); chunk_end = chunk_next + 128u;
        }
        const unsigned base = chunk_next;
        const unsigned my = base + (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
        chunk_next = base + (unsigned)cnt < chunk_end ? base + (unsigned)cnt : chunk_end;
        if (mine) {
          if (my < n && my < chunk_end) {
            const float4 a = rays[2 * (size_t)my], b = rays[2 * (size_t)my + 1];
            idx = my; o = mk(a.x, a.y, a.z); d = mk(b.x, b.y, b.z);
            inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            wr = wide_ray(o, d, inv, P.wide_pmax, P.wide_mu.e, P.wide_mu.l, P.wide_mu.v); sg = sign_mask(inv);
            trav_begin(tr); ws.top = 0u; ws.sp = 0; ws.sb = 0;
          } else tr.node = my >= n ? -3 : -2;      // (-2: the chunk ran out in the middle of the wave's request: next refill)
        }
        exhausted = chunk_next >= n;
      } else if (mine) tr.node = -3;
      if (__ballot(tr.node != -3) == 0ull) break;
    }
    for (int u = 0; u < 2; u++) {
      const bool at_leaf = tr.node >= 0 && (tr.node & 1);
      const unsigned long long leaves = __ballot(at_leaf), nodes = __ballot(tr.node >= 0 && !(tr.node & 1));
      const bool do_leaves = leaves != 0ull && ((int)__popcll(leaves) >= PARK || nodes == 0ull || exhausted);
      const bool do_nodes = !do_leaves || exhausted;
      if (tr.node >= 0 && (at_leaf ? do_leaves : do_nodes)) {
        const WideRec r = wide_fetch(walk, tr.node);
        if (at_leaf) wide_leaf_compute<false>(r, o, d, inv, sg, tr, ws, my_stack, c);
        else wide_node_compute<false>(r, wr, sg, tr, ws, my_stack, c);
      }
    }
  }
}
// the same rays, one per lane, each wave waiting for its slowest (the reference the probe's results are checked against)
__global__ __launch_bounds__(256) void trace_plain_kernel(RenderParams P, const float4* __restrict__ rays, unsigned n, uint2* __restrict__ out) {
  __shared__ int lds[4 * WIDE_STACK * 64];
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  int* const stack = lds + (threadIdx.x >> 6) * (WIDE_STACK * 64) + (threadIdx.x & 63);
  Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
  const float4 a = rays[2 * (size_t)i], b = rays[2 * (size_t)i + 1];
  const Hit h = closest_hit_wide<false>(wide_rsrc(P), P.wide_pmax, P.wide_mu.e, P.wide_mu.l, P.wide_mu.v, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), c, stack);
  out[i] = h.slot < 0 ? make_uint2(__float_as_uint(10000.0f), 0xffffffffu) : make_uint2(__float_as_uint(h.t), (unsigned)h.slot);
}

// ---- known-answer kernels
__global__ void kat_rng_kernel(uint64_t seed, int n, double* out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    Xorwow r; r.init(seed);
    for (int i = 0; i < n; i++) out[i] = r.uniform_double();
  }
}
__global__ void kat_aabb_kernel(int n, const float* o, const float* d, const float* mn, const float* mx, int32_t* hit, float* dist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 dd = ld3(d + 3 * i);
  V3 inv = mk(1.0f / dd.x, 1.0f / dd.y, 1.0f / dd.z);
  float t;
  bool h = slab(ld3(o + 3 * i), inv, mn + 3 * i, mx + 3 * i, t);
  hit[i] = h; dist[i] = h ? t : 0;
}
__global__ void kat_node_planes_kernel(int n, const uint32_t* w, const float* a, const float* b, float* t_mix, float* t_cvt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const PlanePairs p = plane_pairs(w[i]);
  const float a24 = a[i] * 0x1p24f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    t_mix[4 * i + k] = plane_t(p, k, a24, b[i]);                                                         // the kernel's instruction (v_fma_mix_f32 on the f16 denormal)
    t_cvt[4 * i + k] = __builtin_fmaf((float)((w[i] >> (8 * k)) & 255u), a[i], b[i]);                    // conversion + fma
  }
}
__global__ void kat_tri_kernel(int n, const float* o, const float* d, const float* v0, const float* v1, const float* v2, float* t) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 a = ld3(v0 + 3 * i);
  t[i] = tri_hit(ld3(o + 3 * i), ld3(d + 3 * i), a, ld3(v1 + 3 * i) - a, ld3(v2 + 3 * i) - a);
}
__global__ void kat_sphere_kernel(int n, const float* o, const float* d, const float* c, const float* r, float* t) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  t[i] = sphere_hit(ld3(c + 3 * i), r[i], ld3(o + 3 * i), ld3(d + 3 * i));
}
__global__ void kat_optics_kernel(int n, const float* v, const float* nrm, const float* eta, float* refl, float* refr, float* sch) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3 a = ld3(v + 3 * i), b = ld3(nrm + 3 * i);
  V3 r1 = reflect(a, b), r2 = refract(a, b, eta[i]);
  refl[3 * i] = r1.x; refl[3 * i + 1] = r1.y; refl[3 * i + 2] = r1.z;
  refr[3 * i] = r2.x; refr[3 * i + 1] = r2.y; refr[3 * i + 2] = r2.z;
  sch[i] = reflectance(a.x, eta[i]);
}
__global__ void kat_normal_kernel(RenderParams P, int n, const int32_t* slot, const float* o, const float* d, const float* t, float* nrm, float* texco) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4* pp = reinterpret_cast<const float4*>(P.prims + slot[i]);
  const float4* sp = reinterpret_cast<const float4*>(P.shade + slot[i]);
  const V3 ro = ld3(o + 3 * i), rd = ld3(d + 3 * i);
  const V3 hitpoint = ro + splat(t[i]) * rd;                           // K:806
  V3 tc;
  const V3 N = surface_normal(pp[0], pp[1], pp[2], sp[0], sp[1], sp[2], sp[3], sp[4], sp[6], ro, rd, hitpoint, tc);
  nrm[3 * i] = N.x; nrm[3 * i + 1] = N.y; nrm[3 * i + 2] = N.z;
  texco[3 * i] = tc.x; texco[3 * i + 1] = tc.y; texco[3 * i + 2] = tc.z;
}
template <int MODE>
__global__ __launch_bounds__(256) void kat_hit_kernel(RenderParams P, int n, const float* o, const float* d, float* t, int32_t* slot, int32_t* visits) {
  __shared__ int lds_stack[MODE == DR_TRAVERSAL_ORDERED ? ORDERED_STACK * 256 : (MODE == DR_TRAVERSAL_WIDE ? WIDE_STACK * 256 : 1)];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
  Hit h;
  if (MODE == DR_TRAVERSAL_ORDERED) {
    int* stack = lds_stack + (threadIdx.x >> 6) * (ORDERED_STACK * 64) + (threadIdx.x & 63);
    h = closest_hit_ordered<true>(P.pairs, P.prims, ld3(o + 3 * i), ld3(d + 3 * i), c, stack);
  } else if (MODE == DR_TRAVERSAL_WIDE) {
    int* stack = lds_stack + (threadIdx.x >> 6) * (WIDE_STACK * 64) + (threadIdx.x & 63);
    h = closest_hit_wide<true>(wide_rsrc(P), P.wide_pmax, P.wide_mu.e, P.wide_mu.l, P.wide_mu.v, ld3(o + 3 * i), ld3(d + 3 * i), c, stack);
  } else {
    h = closest_hit_threaded<true>(walk_rsrc(P), ld3(o + 3 * i), ld3(d + 3 * i), c);
  }
  t[i] = h.t; slot[i] = h.slot;
  if (visits) visits[i] = (int32_t)c.V;
}

// acc += frame, 16 bytes per lane (pipelined single frames: every frame renders into a buffer of its own and is folded into
// the accumulator in frame order, K:2213-2218)
__global__ __launch_bounds__(256) void frame_add_kernel(int4* __restrict__ acc, const int4* __restrict__ frame, size_t n4, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    int4 a = acc[i];
    const int4 f = frame[i];
    a.x += f.x; a.y += f.y; a.z += f.z; a.w += f.w;
    acc[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3))       // W * H * 3 need not be a multiple of 4
    reinterpret_cast<int*>(acc)[n4 * 4 + threadIdx.x] += reinterpret_cast<const int*>(frame)[n4 * 4 + threadIdx.x];
}

// ------------------------------------------------------------------ launchers
void launch_tile_feedback(hipStream_t stream, const unsigned* pixel_cost, unsigned* tile_cost, int* tile_order, int* region_start, int tiles, int regions,
                          int heavy_factor, int split_steps, int split_limit) {
  hipLaunchKernelGGL(tile_cost_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, pixel_cost, tile_cost, tiles);
  hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, tile_cost, tile_order, region_start, tiles, regions, heavy_factor, split_steps, split_limit);
}
void launch_present(hipStream_t stream, const int32_t* acc, uint8_t* rgb, int W, int H, int div) {
  const int n = W * H;
  hipLaunchKernelGGL(present_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, acc, rgb, W, H, div);
}
void launch_frame_add(hipStream_t stream, int32_t* acc, const int32_t* frame, size_t n) {      // n int32
  const size_t n4 = n / 4;
  size_t blocks = (n4 + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(frame_add_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<int4*>(acc), reinterpret_cast<const int4*>(frame), n4, n);
}
void launch_stripe_copy(hipStream_t stream, int32_t* dst, const int32_t* src, int ncols, int run4, long long dst_first4, long long dst_stride4,
                        long long src_first4, long long src_stride4) {
  int bx = (run4 + 255) / 256; if (bx > 64) bx = 64;
  hipLaunchKernelGGL(stripe_copy_kernel, dim3((unsigned)bx, (unsigned)ncols), dim3(256), 0, stream, reinterpret_cast<int4*>(dst), reinterpret_cast<const int4*>(src),
                     ncols, run4, dst_first4, dst_stride4, src_first4, src_stride4);
}
void launch_gather_probe(hipStream_t stream, const RenderParams& P, int blocks, unsigned nrec, int iters, unsigned* out) {
  hipLaunchKernelGGL(gather_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, P, nrec, iters, out);
}
void launch_trace_probe(hipStream_t stream, const RenderParams& P, int num_cus, int variant, const float* rays, unsigned n, unsigned* cursor, unsigned* out) {
  const float4* r4 = reinterpret_cast<const float4*>(rays);
  uint2* o2 = reinterpret_cast<uint2*>(out);
  switch (variant) {
    case 0: hipLaunchKernelGGL(trace_plain_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, P, r4, n, o2); break;
    case 1: hipLaunchKernelGGL((trace_probe_kernel<6, 1, 20>), dim3((unsigned)(num_cus * 6)), dim3(256), 0, stream, P, r4, n, cursor, o2); break;
    case 2: hipLaunchKernelGGL((trace_probe_kernel<6, 8, 20>), dim3((unsigned)(num_cus * 6)), dim3(256), 0, stream, P, r4, n, cursor, o2); break;
    case 3: hipLaunchKernelGGL((trace_probe_kernel<6, 16, 20>), dim3((unsigned)(num_cus * 6)), dim3(256), 0, stream, P, r4, n, cursor, o2); break;
    case 4: hipLaunchKernelGGL((trace_probe_kernel<8, 8, 20>), dim3((unsigned)(num_cus * 8)), dim3(256), 0, stream, P, r4, n, cursor, o2); break;
    case 5: hipLaunchKernelGGL((trace_probe_kernel<8, 8, 28>), dim3((unsigned)(num_cus * 8)), dim3(256), 0, stream, P, r4, n, cursor, o2); break;
    case 6: hipLaunchKernelGGL((trace_probe_kernel<6, 8, 28>), dim3((unsigned)(num_cus * 6)), dim3(256), 0, stream, P, r4, n, cursor, o2); break;
    default: hipLaunchKernelGGL((trace_probe_kernel<8, 4, 32>), dim3((unsigned)(num_cus * 8)), dim3(256), 0, stream, P, r4, n, cursor, o2); break;
  }
}
void launch_kat_rng(hipStream_t stream, uint64_t seed, int n, double* out) { hipLaunchKernelGGL(kat_rng_kernel, dim3(1), dim3(64), 0, stream, seed, n, out); }
void launch_kat_aabb(hipStream_t stream, int n, const float* o, const float* d, const float* mn, const float* mx, int32_t* hit, float* dist) {
  hipLaunchKernelGGL(kat_aabb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, o, d, mn, mx, hit, dist);
}
void launch_kat_node_planes(hipStream_t stream, int n, const uint32_t* w, const float* a, const float* b, float* t_mix, float* t_cvt) {
  hipLaunchKernelGGL(kat_node_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, w, a, b, t_mix, t_cvt);
}
void launch_kat_tri(hipStream_t stream, int n, const float* o, const float* d, const float* v0, const float* v1, const float* v2, float* t) {
  hipLaunchKernelGGL(kat_tri_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, o, d, v0, v1, v2, t);
}
void launch_kat_sphere(hipStream_t stream, int n, const float* o, const float* d, const float* c, const float* r, float* t) {
  hipLaunchKernelGGL(kat_sphere_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, o, d, c, r, t);
}
void launch_kat_optics(hipStream_t stream, int n, const float* v, const float* nrm, const float* eta, float* refl, float* refr, float* sch) {
  hipLaunchKernelGGL(kat_optics_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, v, nrm, eta, refl, refr, sch);
}
void launch_kat_normal(hipStream_t stream, const RenderParams& P, int n, const int32_t* slot, const float* o, const float* d, const float* t, float* nrm, float* texco) {
  hipLaunchKernelGGL(kat_normal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P, n, slot, o, d, t, nrm, texco);
}
void launch_kat_hit(hipStream_t stream, const RenderParams& P, int traversal, int n, const float* o, const float* d, float* t, int32_t* slot, int32_t* visits) {
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (traversal == DR_TRAVERSAL_WIDE) hipLaunchKernelGGL((kat_hit_kernel<DR_TRAVERSAL_WIDE>), grid, block, 0, stream, P, n, o, d, t, slot, visits);
  else if (traversal == DR_TRAVERSAL_ORDERED) hipLaunchKernelGGL((kat_hit_kernel<DR_TRAVERSAL_ORDERED>), grid, block, 0, stream, P, n, o, d, t, slot, visits);
  else hipLaunchKernelGGL((kat_hit_kernel<DR_TRAVERSAL_THREADED>), grid, block, 0, stream, P, n, o, d, t, slot, visits);
}

}  // namespace dr
