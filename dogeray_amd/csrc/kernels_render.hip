// The render kernels: the per-tile kernel (the reference's launch shape, kernel.cu K:2634-2640) and the persistent kernel (a wave is a
// pool of 64 path slots that refill from tile queues).  Launchers at the end; context.cpp picks options, this file picks the instantiation.
#include <hip/hip_runtime.h>

#include "device_core.hpp"
#include "kernels.hpp"
#include "persistent_kernel.hpp"
#include "../../include/dogeray_amd.h"

namespace dr {

// ------------------------------------------------------------------ kernels
// One wave = one 8x8 pixel tile, as in the reference launch (K:2634-2640: block (8,8)); four
// tiles per 256-thread workgroup.  Tiles are numbered column-major over the block columns this
// context owns, so neighbouring waves work on vertically adjacent tiles (coherent rays, and the
// column-major framebuffer gives each wave eight 96-byte runs).
template <bool COUNT, int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void render_kernel(RenderParams P) {
  __shared__ int lds_stack[MODE == DR_TRAVERSAL_ORDERED ? ORDERED_STACK * 256 : (MODE == DR_TRAVERSAL_WIDE ? WIDE_STACK * 256 : 1)];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
  if (tile < P.ncols * P.gy) {
    const int col = tile / P.gy, by = tile - col * P.gy;
    const int bx = P.stripe_rem + col * P.stripe_mod;
    const int x = bx * 8 + (lane >> 3), y = by * 8 + (lane & 7);
    if (MODE == DR_TRAVERSAL_ORDERED) {
      int* stack = lds_stack + wave * (ORDERED_STACK * 64) + lane;
      auto closest = [&](V3 o, V3 d, Ctr& cc) { return closest_hit_ordered<COUNT>(P.pairs, P.prims, o, d, cc, stack); };
      render_pixel<COUNT>(P, closest, x, y, c);
    } else if (MODE == DR_TRAVERSAL_WIDE) {
      const WalkRsrc wide = wide_rsrc(P);
      int* stack = lds_stack + wave * (WIDE_STACK * 64) + lane;
      auto closest = [&](V3 o, V3 d, Ctr& cc) { return closest_hit_wide<COUNT>(wide, P.wide_pmax, o, d, cc, stack); };
      render_pixel<COUNT>(P, closest, x, y, c);
    } else {
      const WalkRsrc walk = walk_rsrc(P);
      auto closest = [&](V3 o, V3 d, Ctr& cc) { return closest_hit_threaded<COUNT>(walk, o, d, cc); };
      render_pixel<COUNT>(P, closest, x, y, c);
    }
  }
  if (COUNT) {
    unsigned v[8] = {c.rays, c.V, c.L, c.S, c.T, c.samples, c.trav_slots, c.ray_slots};
    for (int k = 0; k < 8; k++) {
      unsigned long long s = v[k];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lane == 0 && s) atomicAdd(&P.counters[k], s);
    }
  }
}

// ------------------------------------------------------------------ launchers
void launch_tile_kernel(hipStream_t stream, const RenderParams& P, int traversal, bool count, int occupancy) {
  const int tiles = P.ncols * P.gy;
  dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
#define DR_TILE(COUNT, MODE, OCC) hipLaunchKernelGGL((render_kernel<COUNT, MODE, OCC>), grid, block, 0, stream, P)
#define DR_TILE_OCC(OCC)                                                                   \
  do {                                                                                     \
    if (count) {                                                                           \
      if (traversal == DR_TRAVERSAL_ORDERED) DR_TILE(true, DR_TRAVERSAL_ORDERED, OCC);     \
      else if (traversal == DR_TRAVERSAL_WIDE) DR_TILE(true, DR_TRAVERSAL_WIDE, OCC);      \
      else DR_TILE(true, DR_TRAVERSAL_THREADED, OCC);                                      \
    } else {                                                                               \
      if (traversal == DR_TRAVERSAL_ORDERED) DR_TILE(false, DR_TRAVERSAL_ORDERED, OCC);    \
      else if (traversal == DR_TRAVERSAL_WIDE) DR_TILE(false, DR_TRAVERSAL_WIDE, OCC);     \
      else DR_TILE(false, DR_TRAVERSAL_THREADED, OCC);                                     \
    }                                                                                      \
  } while (0)
  if (occupancy >= 6) DR_TILE_OCC(6);
  else DR_TILE_OCC(4);
#undef DR_TILE_OCC
#undef DR_TILE
}

namespace {

template <int OCC, int TRAV_MIN, int PARK_MIN, int P_UNROLL = 1>
int launch_persistent(hipStream_t stream, const RenderParams& P_in, const PersistentCfg& cfg, unsigned* counter, const int* order, const int* rstart, unsigned* pixel_cost) {
  RenderParams P = P_in;
  int work = P.ncols * P.gy * P.batch;
  int blocks = cfg.num_cus * (cfg.wgs_per_cu > 0 && cfg.wgs_per_cu < OCC ? cfg.wgs_per_cu : OCC);      // OCC waves per SIMD on every CU
  if (blocks * 4 > work) blocks = (work + 3) / 4;
  if (P.wave_log && blocks * 4 > WAVE_LOG_WAVES) P.wave_log = nullptr;
  int log_waves = P.wave_log ? blocks * 4 : 0;
  dim3 grid((unsigned)blocks), block(256);
  if (cfg.traversal == DR_TRAVERSAL_WIDE) {
    // the cooperative drain shortens a launch's tail; with many tiles per wave the tail does not show and the leaner build is faster
    const bool coop = P.coop_steps > 0 && ((long long)work < (long long)cfg.coop_tiles_per_wave * blocks * 4 || P.duo == 2);
    if (!coop) log_waves = 0;      // only the work-sharing build writes the log
    if (cfg.count) hipLaunchKernelGGL((render_persistent_kernel<true, OCC, TRAV_MIN, PARK_MIN, P_UNROLL, true, false>), grid, block, 0, stream, P, counter, order, rstart, pixel_cost);
    else if (coop) hipLaunchKernelGGL((render_persistent_kernel<false, OCC, TRAV_MIN, PARK_MIN, P_UNROLL, true, true>), grid, block, 0, stream, P, counter, order, rstart, pixel_cost);
    else hipLaunchKernelGGL((render_persistent_kernel<false, OCC, TRAV_MIN, PARK_MIN, P_UNROLL, true, false>), grid, block, 0, stream, P, counter, order, rstart, pixel_cost);
    return log_waves;
  }
  if (cfg.count) hipLaunchKernelGGL((render_persistent_kernel<true, OCC, TRAV_MIN, PARK_MIN, P_UNROLL, false>), grid, block, 0, stream, P, counter, order, rstart, pixel_cost);
  else hipLaunchKernelGGL((render_persistent_kernel<false, OCC, TRAV_MIN, PARK_MIN, P_UNROLL, false>), grid, block, 0, stream, P, counter, order, rstart, pixel_cost);
  return log_waves;
}

// Six waves per SIMD (occupancy 6): only the wide walk's lean build fits -- 80 VGPRs and 26 KiB of LDS per workgroup -- and only with the
// default schedule; every other launch (counting build, work-sharing build of short launches, other schedules) runs five.
bool launch_wide_lean6(hipStream_t stream, const RenderParams& P_in, const PersistentCfg& cfg, unsigned* counter, const int* order, const int* rstart, unsigned* pixel_cost, int& log_waves) {
  if (cfg.traversal != DR_TRAVERSAL_WIDE || cfg.count || cfg.schedule != 0) return false;
  RenderParams P = P_in;
  const long long work = (long long)P.ncols * P.gy * P.batch;
  if (P.coop_steps > 0 && work < (long long)cfg.coop_tiles_per_wave * cfg.num_cus * 5 * 4) return false;      // a short launch: work-sharing build
  int blocks = cfg.num_cus * 6;
  if ((long long)blocks * 4 > work) blocks = (int)((work + 3) / 4);
  P.wave_log = nullptr;      // (the lean kernel does not log its waves)
  log_waves = 0;
  hipLaunchKernelGGL((render_persistent_kernel<false, 6, 32, 20, 2, true, false>), dim3((unsigned)blocks), dim3(256), 0, stream, P, counter, order, rstart, pixel_cost);
  return true;
}

// The instantiated schedules (option "schedule"): 0 = the tuned one -- shade / refill below 32 walking lanes, leaf steps for 20 lanes, two steps per
// loop iteration --; 1 and 2 keep the other paths of the loop alive in the tests (leaf steps for 8 lanes, one step per iteration; shade / refill
// below 48 lanes, leaves tested on the spot).  Every other combination rounds 2 and 3 measured is in profiles/r2_*, r3_p_*.
template <int OCC>
int launch_persistent_occ(hipStream_t stream, const RenderParams& P, const PersistentCfg& cfg, unsigned* counter, const int* order, const int* rstart, unsigned* pcost) {
  switch (cfg.schedule) {
    case 0:  return launch_persistent<OCC, 32, 20, 2>(stream, P, cfg, counter, order, rstart, pcost);
    case 1:  return launch_persistent<OCC, 32, 8, 1>(stream, P, cfg, counter, order, rstart, pcost);
    default: return launch_persistent<OCC, 48, 0, 1>(stream, P, cfg, counter, order, rstart, pcost);
  }
}

}  // namespace

bool persistent_launch_is_short(const RenderParams& P, const PersistentCfg& cfg) {
  if (cfg.traversal != DR_TRAVERSAL_WIDE || cfg.count || P.coop_steps <= 0) return false;
  const long long work = (long long)P.ncols * P.gy * P.batch;
  return work < (long long)cfg.coop_tiles_per_wave * cfg.num_cus * 5 * 4;
}

int launch_persistent_kernel(hipStream_t stream, const RenderParams& P, const PersistentCfg& cfg, unsigned* tile_counter, const int* order,
                             const int* region_start, unsigned* pixel_cost) {
  const int* rstart = order ? region_start : nullptr;        // identity order: the split travels in P.region_start
  int log_waves = 0;
  if (cfg.occupancy >= 6 && launch_wide_lean6(stream, P, cfg, tile_counter, order, rstart, pixel_cost, log_waves)) return log_waves;
  if (cfg.occupancy >= 5) return launch_persistent_occ<5>(stream, P, cfg, tile_counter, order, rstart, pixel_cost);
  return launch_persistent_occ<4>(stream, P, cfg, tile_counter, order, rstart, pixel_cost);
}

}  // namespace dr
