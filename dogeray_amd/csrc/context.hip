// Device context: resident scene, render launches, on-device accumulation, stats, KAT hooks.
// Replaces CudaStarter (kernel.cu K:2562-2669), which mallocs, uploads the whole scene,
// launches, synchronises, downloads and frees on every call.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "device_core.hpp"
#include "linearise.hpp"
#include "scene_host.hpp"

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

// Persistent variant of the same megakernel.
//
// In the per-tile kernel above a wave's 64 lanes walk the BVH in lock step until the LAST of
// them is done, then shade, then walk again: on the 1M-triangle bench scene only 36 % of the
// lane-slots of the node loop and 77 % of the bounce loop do useful work.  Here a wave is a
// pool of 64 path slots.  Every loop iteration advances each walking lane by one node; as soon
// as fewer than TRAV_MIN lanes are still walking, the finished lanes are shaded (hit or miss),
// scatter into their next ray, or -- when their path has ended -- store their pixel and take the
// next unrendered pixel, so the node loop stays full until the frame runs out of pixels.
// Pixels are handed out in 8x8 tiles from per-XCD queues (one atomicAdd per tile), lane l of a tile is
// pixel (l >> 3, l & 7) of that tile: which lane renders a pixel does not enter its arithmetic (the RNG
// seed is a function of x, y and the frame, K:1065), so the frame is identical to the per-tile kernel's.
//
// lane states (kept in `tr.node`): >= 0 walking; -1 walk finished, needs shading; -2 needs a new
// sample or pixel; -3 retired.
// WIDE: the lanes walk the 4-way tree (wide_node_step / wide_leaf_step) instead of the threaded links.  A lane whose
// next record is a leaf waits until PARK_MIN lanes have one (the leaf step -- exact box + triangle -- is the long
// block, as the parked triangle test is in the threaded walk); its stack lives in the first WIDE_STACK words of
// the wave's LDS region, the phase stash behind it.
// COOP (wide walk): build with work sharing (lanes without a pixel take over subtrees of the rays still walking: once the queue is
// empty, and from the start in the waves that hold a part of a split tile).  It costs registers (96 VGPRs: five waves per SIMD) and
// code in the loop, so launches whose queue is long enough to hide their tail use the lean build (80 VGPRs and 26 KiB of LDS at
// OCC = 6: six waves per SIMD; launch_persistent / launch_wide_lean6 pick).
constexpr int MAX_REGIONS = 8;
constexpr int WAVE_LOG_WAVES = 16384;    // waves the wave log (option wave_log) has room for
constexpr size_t PIXEL_LOG_WORDS = DR_WAVE_LOG_DETAIL ? (size_t)2 * 4096 * 4096 : 0;   // experiment builds: start and end stamp of every pixel behind the wave log

template <bool COUNT, int OCC, int TRAV_MIN, int PARK_MIN, int P_UNROLL, bool WIDE, bool COOP = true>
__global__ __launch_bounds__(256, OCC) void render_persistent_kernel(RenderParams P, unsigned* __restrict__ tile_counter,
                                                                     const int* __restrict__ tile_order, const int* __restrict__ region_start,
                                                                     unsigned* __restrict__ pixel_cost) {
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  // LDS per wave (threaded walk: 6 KiB; wide walk: 6.5 KiB, 7.25 with work sharing), used for two things that never overlap in time
  // within a wave:
  //  * during the shade/refill phase, the state that phase does not need (threaded walk: the parked leaf, 1/direction; and while a
  //    hit is shaded also the pixel bookkeeping) waits here -- the shading code is where register pressure peaks, and this keeps
  //    the kernel within the VGPRs of its occupancy;
  //  * outside the phase, the node stack of the threaded walk's cooperative drain (COOP_STACK entries) / the exchange words of
  //    the wide walk's work sharing.  The wide walk's own stack (WIDE_STACK words per lane) sits in front of the stash.
  // WIDE && COOP: three more words per lane behind the stash -- the shared best hit (64-bit key) and the number of helper lanes of
  // a ray whose subtrees have been handed out (drain phase, below)
  constexpr int SHARE_OFF = (WIDE_STACK + WIDE_STASH) * 64;
  constexpr int REGION = WIDE ? (WIDE_STACK + WIDE_STASH + (COOP ? 3 : 0)) * 64 : WAVE_LDS_DWORDS;
  __shared__ __attribute__((aligned(16))) int wave_lds[4 * REGION];
  int* const my_lds = wave_lds + (threadIdx.x >> 6) * REGION;
  int* const my_stack = my_lds + lane;                             // WIDE: word k of this lane's stack at my_stack[k * 64]
  unsigned long long* const share_key = reinterpret_cast<unsigned long long*>(my_lds + (WIDE ? SHARE_OFF : 0));     // [64], WIDE && COOP only
  unsigned* const share_pend = reinterpret_cast<unsigned*>(my_lds + (WIDE ? SHARE_OFF : 0) + 128);                   // [64]
  int share = -1;                  // -1: this lane walks a ray of its own, alone; 0..63: it helps that lane's ray; 64: its ray has helpers
  WideStack ws; ws.top = 0u; ws.sp = 0; ws.sb = 0;
  const int ntiles = P.ncols * P.gy;
  const int nwork = ntiles * P.batch;          // queue length: every tile of every frame of the batch
  const WalkRsrc walk = WIDE ? wide_rsrc(P) : walk_rsrc(P);
  Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
  // wave-uniform work cursor
  // The queue hands out positions q = 0, 1, 2, ...; tile_order (when present) maps a position to
  // a tile so that the tiles that were most expensive in the previous frame of this view start
  // first (longest-processing-time-first: the kernel's duration is otherwise set by whichever
  // expensive tile happens to start last).
  // A launch may cover a batch of frames (same view, consecutive seeds): position q is tile
  // order[q / batch] of frame q % batch, so the expensive tiles of ALL frames start first and the
  // tail of one frame (its longest paths) overlaps the bulk of the others.
  // The queue is split into P.regions contiguous parts (tiles are numbered column by column, so a part
  // is a band of the image).  With 8 regions every XCD drains "its" band first -- its 4 MiB L2 then holds
  // the part of the scene that band sees instead of competing for all of it -- and helps the next band
  // once its own is empty.  Region r owns positions [region_start[r], region_start[r+1]) of the order.
  int cur_tile = ntiles, cur_frame = 0;        // chunk being handed out; ntiles = none
  int cur_next = 64;               // next unassigned lane-in-tile of cur_tile (64 = exhausted: fetch first)
  int cur_limit = 64;              // ... and where this wave's share of cur_tile ends (split tiles: a part of the tile)
  bool cur_split = false;
  // tiles at the head of each region's order that are handed out in P.split_parts parts (work-sharing build, order from feedback)
  // (launches of ONE frame: with several frames in the queue the long pixels of one overlap the bulk of the others anyway, and waves that hold cost throughput)
  const bool splitting = WIDE && COOP && region_start && P.split_parts > 1 && P.batch == 1;      // region r's count: region_start[MAX_REGIONS + 1 + r]
  int region = 0, regions_left = P.regions;
  if (P.regions > 1) region = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) % (unsigned)P.regions);   // HW_REG_XCC_ID, 4 bits
  (void)nwork;
  // per-lane path slot
  Trav tr; tr.node = -2; tr.best_t = 0; tr.best_slot = -1;
  Path path; path.rayo = mk(0, 0, 0); path.raydir = mk(0, 0, 0); path.atten = mk(0, 0, 0);
  V3 inv = mk(0, 0, 0), color = mk(0, 0, 0);
  WideRay wr; wr.inv = wr.marg = mk(0, 0, 0);   // WIDE: clamped 1/direction and margins of the folded node test, a function of the lane's ray
  Xorwow rng; rng.v0 = rng.v1 = rng.v2 = rng.v3 = rng.v4 = rng.d = 0;
  int px = -1, py = 0, sample = 0, bounce = 0;
  int frame = 0;                   // frame of the batch the pixel in this slot belongs to
  int pcode = 0;                   // tile * 64 + lane-in-tile of the pixel in this slot
  unsigned steps = 0;              // node steps spent on the pixel in this slot (the cost fed back)
  unsigned rstart = 0;             // value of `steps` when the current ray started
  const bool degenerate = P.max_depth <= 0 || !(P.spp_f > 0.0f);

  // diagnostic stamps (counting build only): wave lifetime, cycles inside the shade/refill phase
  unsigned long long t_begin = 0, t_phase = 0, n_iter = 0, n_phase = 0, n_nodestep = 0, n_leafstep = 0, n_shaded = 0;
  unsigned long long r_begin = 0;
  // wave lifetime in shader cycles and in 100 MHz ticks, every build: two clock reads per wave, written to the statistics buffer only
  t_begin = __builtin_readcyclecounter(); r_begin = __builtin_amdgcn_s_memrealtime();
  bool held = false;               // COOP: this wave has pixels of a split tile and fetches no new tiles while they live
  // wave log (option wave_log): when this wave first found the queue empty, loop iterations since.  Only in the build that short launches
  // use (their timeline is what the log is for): the two scalar instructions per iteration cost the long launches 0.5 %
  constexpr bool WAVE_LOG = (WIDE && COOP) || DR_WAVE_LOG_DETAIL;
  unsigned long long r_empty = 0, n_after = 0;
  unsigned long long d_iters = 0;      // loop iterations of the wave
  unsigned long long d_home_tiles = 0, d_stolen_tiles = 0, d_left_home = 0;      // (tools/exp_regions.py) tiles from the region the wave began in / from others, when it left the first
  const int d_home = region;
  unsigned long long d_want_give = 0, d_idle = 0, d_owner_walk = 0, d_share_iters = 0, d_wait_owner = 0, d_pending = 0;   // ... of the iterations in which the sharing block ran: givers, idle lanes, walking owners, owners waiting for helpers, lanes waiting for a phase
  unsigned long long d_phases = 0, d_given = 0, d_walking = 0, d_phase_ticks = 0;      // -DDR_WAVE_LOG_DETAIL builds: phases, hand-overs, walking lanes summed, ticks inside phases, all after the queue ran empty
  ParkedLeaf pk; pk.v0x = 0; pk.C = pk.D = u32x4{0, 0, 0, 0}; pk.info = 0; pk.parked = false;   // PARK_MIN > 0 only
  for (;;) {
    const unsigned long long walking = __ballot(tr.node >= 0 || (PARK_MIN > 0 && pk.parked));
    if (COUNT) n_iter++;
    if (WAVE_LOG && r_empty != 0ull) n_after++;
    if (DR_WAVE_LOG_DETAIL) d_iters++;
    if (DR_WAVE_LOG_DETAIL && r_empty != 0ull) d_walking += (unsigned long long)__popcll(__ballot(tr.node >= 0));
    // (a wave that only drains -- queue empty, nobody waiting to be shaded or refilled -- skips the phase: its stash/restore would be
    // paid on every iteration of the launch's tail)
    if (WIDE && COOP) {
      // a helper whose subtree is done reports (its best is in the key already) and is idle again; an owner whose own part is
      // done takes the shared result once its last helper has reported
      // (the pending word: helpers still out in the low 8 bits; above them the node steps the helpers have spent on this pixel's rays,
      // which the owner adds to its own when it takes the result -- the cost fed back for a pixel is all the work it caused)
      if (tr.node == -1 && share >= 0 && share < 64) {
        atomicMin(&share_key[share], hit_key(tr.best_t, tr.best_slot));
        atomicAdd(&share_pend[share], (((steps - rstart - (unsigned)P.coop_steps) & 0xffffu) << 8) - 1u);
        tr.node = -3; share = -1;
      }
      if (tr.node == -1 && share == 64) {
        atomicMin(&share_key[lane], hit_key(tr.best_t, tr.best_slot));      // its own last improvement may be newer than the key
        if ((share_pend[lane] & 0xffu) == 0u) {
          steps += share_pend[lane] >> 8;
          const unsigned long long k = share_key[lane];
          tr.best_t = __uint_as_float((unsigned)(k >> 32)); tr.best_slot = (int)(unsigned)k;
          share = -1;
        }
      }
    }
    const bool waits_for_helpers = WIDE && COOP && share == 64;      // (only ever true with tr.node == -1 here or while still walking)
    // (a holding wave -- long pixels, helpers walking for them -- shades as soon as a ray is finished: its pixels' latency is the point)
    if ((__popcll(walking) < TRAV_MIN || (WIDE && COOP && held)) && (walking == 0ull || __ballot((tr.node == -1 && !waits_for_helpers) || tr.node == -2) != 0ull)) {
      unsigned long long t0 = 0;
      if (COUNT) { t0 = __builtin_readcyclecounter(); n_phase++; }
      unsigned long long d_t0 = 0;
      if (DR_WAVE_LOG_DETAIL && r_empty != 0ull) { d_phases++; d_t0 = __builtin_amdgcn_s_memrealtime(); }
      const bool shade_me = tr.node == -1 && !(PARK_MIN > 0 && pk.parked) && !waits_for_helpers;
      if (COUNT) n_shaded += __popcll(__ballot(shade_me));
      bool fresh_ray = false;                  // this lane starts a new ray in this phase: 1/direction is recomputed after the phase
      float* const st = reinterpret_cast<float*>(my_lds) + (WIDE ? WIDE_STACK * 64 : 0) + lane;      // slot k of this lane: st[k * 64]
      if (WIDE) {
        // ten words (26 KiB of LDS per workgroup with the stack: six workgroups per CU): the stack pointer shares a word with the frame
        // of the batch, x + 1 (0: no pixel) with y -- make_params bounds the frame size
        st[0 * 64] = __uint_as_float(ws.top); st[1 * 64] = __int_as_float(ws.sp | (frame << 8));
        st[2 * 64] = color.x; st[3 * 64] = color.y; st[4 * 64] = color.z;
        st[5 * 64] = __int_as_float(((px + 1) << 16) | py); st[6 * 64] = __int_as_float(pcode);
        st[7 * 64] = __int_as_float(sample);
        st[8 * 64] = __uint_as_float(steps); st[9 * 64] = __uint_as_float(rstart);
        asm volatile("" ::: "memory");
      } else {
        st[0 * 64] = pk.v0x; st[1 * 64] = __uint_as_float(pk.C.x); st[2 * 64] = __uint_as_float(pk.C.y); st[3 * 64] = __uint_as_float(pk.C.z); st[4 * 64] = __uint_as_float(pk.C.w);
        st[5 * 64] = __uint_as_float(pk.D.x); st[6 * 64] = __uint_as_float(pk.D.y); st[7 * 64] = __uint_as_float(pk.D.z); st[8 * 64] = __uint_as_float(pk.D.w);
        st[9 * 64] = __int_as_float(pk.info); st[10 * 64] = pk.parked ? 1.0f : 0.0f;
        st[11 * 64] = inv.x; st[12 * 64] = inv.y; st[13 * 64] = inv.z;
        // pixel bookkeeping: not needed while the hit is shaded, back right after
        st[14 * 64] = color.x; st[15 * 64] = color.y; st[16 * 64] = color.z;
        st[17 * 64] = __int_as_float(px); st[18 * 64] = __int_as_float(py); st[19 * 64] = __int_as_float(pcode);
        st[20 * 64] = __int_as_float(frame); st[21 * 64] = __int_as_float(sample);
        st[22 * 64] = __uint_as_float(steps); st[23 * 64] = __uint_as_float(rstart);
        asm volatile("" ::: "memory");           // the values must really travel through LDS (no forwarding in registers)
      }
      // ---- shade the lanes whose walk has finished
      bool ended = false;
      V3 radiance = mk(0, 0, 0);
      if (shade_me) {
        if (tr.best_slot >= 0 && tr.best_t > 0.0f) {
          ended = !shade_hit<COUNT>(P, path, tr.best_t, tr.best_slot, rng, c, radiance);
          if (!ended) {
            bounce++;
            if (bounce >= P.max_depth) ended = true;        // depth exhausted: black (K:981)
          }
        } else {
          radiance = shade_miss<COUNT>(P, path, c);
          ended = true;
        }
      }
      if (WIDE) {
        asm volatile("" ::: "memory");
        color = mk(st[2 * 64], st[3 * 64], st[4 * 64]);
        { const int xy = __float_as_int(st[5 * 64]); px = (int)((unsigned)xy >> 16) - 1; py = xy & 0xffff; }
        pcode = __float_as_int(st[6 * 64]);
        frame = __float_as_int(st[1 * 64]) >> 8; sample = __float_as_int(st[7 * 64]);
        steps = __float_as_uint(st[8 * 64]); rstart = __float_as_uint(st[9 * 64]);
      } else {
        asm volatile("" ::: "memory");
        color = mk(st[14 * 64], st[15 * 64], st[16 * 64]);
        px = __float_as_int(st[17 * 64]); py = __float_as_int(st[18 * 64]); pcode = __float_as_int(st[19 * 64]);
        frame = __float_as_int(st[20 * 64]); sample = __float_as_int(st[21 * 64]);
        steps = __float_as_uint(st[22 * 64]); rstart = __float_as_uint(st[23 * 64]);
      }
      if (shade_me) {
        if (ended) {
          color = color + radiance;
          sample++;
          tr.node = -2;
        } else {
          trav_begin(tr);
          rstart = steps;
          fresh_ray = true;
          if (COUNT) c.rays++;
        }
      }
      // ---- lanes between paths: next sample of the same pixel, or store and take a new pixel
      bool want_pixel = false;
      if (tr.node == -2) {
        if (px >= 0 && (float)sample < P.spp_f && !degenerate) {
          // same pixel, next sample (K:1059)
        } else {
          if (px >= 0) {
            store_pixel(P, px, py, color);
            if (pixel_cost) pixel_cost[pcode & 0x7fffffff] = steps;
            if (DR_WAVE_LOG_DETAIL && P.wave_log)      // experiment builds: when the pixel was finished (100 MHz ticks since the wave began, i.e. since the launch)
              reinterpret_cast<unsigned*>(P.wave_log + (size_t)WAVE_LOG_WAVES * 16)[(size_t)ntiles * 64 + (pcode & 0x7fffffff)] = (unsigned)(__builtin_amdgcn_s_memrealtime() - r_begin);
          }
          px = -1;
          want_pixel = true;
        }
      }
      if (WIDE && COOP && splitting) {
        // A wave that took a part of a split tile holds -- fetches no further tile -- while one of those pixels lives: its lanes, as
        // their own pixels end, are helpers that take over subtrees of the long pixels' rays (the work-sharing block below); once
        // the pixels are done the lanes go back to fetching.
        held = __ballot(px >= 0 && pcode < 0) != 0ull;
        if (!held && tr.node == -3 && share < 0 && !(cur_tile >= ntiles && regions_left == 0)) { tr.node = -2; want_pixel = true; }
      }
      unsigned long long need = __ballot(want_pixel);
      while (need != 0ull) {
        if (cur_next >= cur_limit) {               // wave-uniform: fetch the next chunk (tile, frame)
          if (WIDE && COOP && held) {              // holding: no new tile, the lanes asking become helpers
            if (want_pixel) { tr.node = -3; want_pixel = false; }
            break;
          }
          cur_tile = ntiles;
          cur_next = 0; cur_limit = 64; cur_split = false;
          while (regions_left > 0) {
            const int r0 = region_start ? region_start[region] : P.region_start[region];
            const int r1 = region_start ? region_start[region + 1] : P.region_start[region + 1];
            unsigned t = 0;
            if (lane == 0) t = atomicAdd(tile_counter + region, 1u);
            const int q = (int)__builtin_amdgcn_readfirstlane(t);
            const int nsplit = splitting ? region_start[MAX_REGIONS + 1 + region] : 0;
            if (q < (r1 - r0 + nsplit * (P.split_parts - 1)) * P.batch) {
              int tt = q / P.batch;
              cur_frame = q - tt * P.batch;
              if (WIDE && COOP && tt < nsplit * P.split_parts) {
                // one of the tiles whose pixels were the longest of the last frame (the head of the order): this wave takes
                // 64 / split_parts of its pixels and holds (no further tile) until they are done -- its other lanes help with
                // their rays from the start
                const int part = tt % P.split_parts, width = 64 / P.split_parts;
                tt /= P.split_parts;
                cur_next = part * width; cur_limit = cur_next + width; cur_split = true; held = true;
              } else {
                tt -= nsplit * (P.split_parts - 1);
              }
              cur_tile = tile_order ? tile_order[r0 + tt] : r0 + tt;
              if (DR_WAVE_LOG_DETAIL) { if (region == d_home) d_home_tiles++; else d_stolen_tiles++; }
              break;
            }
            if (DR_WAVE_LOG_DETAIL && region == d_home && d_left_home == 0ull) d_left_home = __builtin_amdgcn_s_memrealtime();
            region = region + 1 == P.regions ? 0 : region + 1;   // this band is done: help with the next one
            regions_left--;
          }
        }
        if (cur_tile >= ntiles) {                  // frame exhausted: retire the lanes still asking
          if (want_pixel) { tr.node = -3; want_pixel = false; }
          if (WAVE_LOG && r_empty == 0ull) r_empty = __builtin_amdgcn_s_memrealtime();
          break;
        }
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
        const int avail = cur_limit - cur_next;
        if (want_pixel && rank < avail) {
          const int l = cur_next + rank;
          const int col = cur_tile / P.gy, by = cur_tile - col * P.gy;
          px = (P.stripe_rem + col * P.stripe_mod) * 8 + (l >> 3);
          py = by * 8 + (l & 7);
          pcode = (cur_tile * 64 + l) | (cur_split ? (int)0x80000000 : 0);      // sign bit: pixel of a split tile (its wave holds while it lives)
          frame = cur_frame;
          steps = 0;
          sample = 0;
          color = mk(0, 0, 0);
          want_pixel = false;
          if (DR_WAVE_LOG_DETAIL && P.wave_log)        // ... and when it was started
            reinterpret_cast<unsigned*>(P.wave_log + (size_t)WAVE_LOG_WAVES * 16)[pcode & 0x7fffffff] = (unsigned)(__builtin_amdgcn_s_memrealtime() - r_begin);
        }
        const int n = __popcll(need);
        cur_next += n < avail ? n : avail;
        need = __ballot(want_pixel);
      }
      // ---- start the next path of every lane that has a pixel and no path
      if (tr.node == -2 && px >= 0) {
        if (degenerate) {
          sample = 0x7fffffff;                     // nothing to trace: the pixel is stored as 0 next round
          if (COUNT) c.samples++;
        } else {
          rng.init(sample_seed(P, px, py, sample, frame));
          if (COUNT) { c.samples++; c.rays++; }
          camera_ray(P, px, py, rng, path.rayo, path.raydir);
          path.atten = splat(1.0f);
          bounce = 0;
          trav_begin(tr);
          rstart = steps;
          fresh_ray = true;
        }
      }
      if (WIDE) {
        asm volatile("" ::: "memory");
        ws.top = __float_as_uint(st[0 * 64]); ws.sp = __float_as_int(st[1 * 64]) & 0xff;
        if (fresh_ray) { ws.top = 0u; ws.sp = 0; ws.sb = 0; }
        inv = mk(1.0f / path.raydir.x, 1.0f / path.raydir.y, 1.0f / path.raydir.z);      // 1/direction and the folded test's margins are
        wr = wide_ray(path.rayo, inv, P.wide_pmax);                                        // recomputed for every lane rather than stashed
      } else {
        asm volatile("" ::: "memory");
        pk.v0x = st[0 * 64]; pk.C = u32x4{__float_as_uint(st[1 * 64]), __float_as_uint(st[2 * 64]), __float_as_uint(st[3 * 64]), __float_as_uint(st[4 * 64])};
        pk.D = u32x4{__float_as_uint(st[5 * 64]), __float_as_uint(st[6 * 64]), __float_as_uint(st[7 * 64]), __float_as_uint(st[8 * 64])};
        pk.info = __float_as_int(st[9 * 64]); pk.parked = st[10 * 64] != 0.0f;
        inv = mk(st[11 * 64], st[12 * 64], st[13 * 64]);
        if (fresh_ray) inv = mk(1.0f / path.raydir.x, 1.0f / path.raydir.y, 1.0f / path.raydir.z);
      }
      if (COUNT) t_phase += __builtin_readcyclecounter() - t0;
      if (DR_WAVE_LOG_DETAIL && d_t0 != 0ull) d_phase_ticks += __builtin_amdgcn_s_memrealtime() - d_t0;
      if (__ballot(tr.node != -3) == 0ull) break;
    }
    if (WIDE && COOP && !COUNT && P.coop_steps > 0 && (cur_tile >= ntiles || held)) {
      // ---- draining (the queue is empty), or holding (pixels of a split tile, see the refill above): lanes without a pixel take over subtrees of the rays that still walk.  A walking lane
      // hands the OLDEST word of its stack (the children of a node near the root that it entered but has not visited: the
      // largest piece of work it owns) to an idle lane, which walks it with the same ray.  All lanes of one ray keep the best
      // hit in one LDS word (ds_min_u64 on the (t, slot) key: the lexicographic minimum whatever the order) and prune
      // against it; the owner shades when its own part and every helper's is done.  A ray that would cost one lane hundreds
      // of dependent steps is spread over the idle lanes at the cost of one hand-over per piece -- every leaf is still
      // tested by exactly one lane, with the reference's box and arithmetic, against a bound no smaller than the final t.
      for (int round = 0; round < P.coop_rounds; round++) {      // (a lane that has just taken a word over may hand part of it on in the next round)
      const unsigned long long idle = __ballot(tr.node == -3);
      const bool can_give = tr.node >= 0 && (ws.sp > ws.sb || ws.top != 0u) && (int)(steps - rstart) >= P.coop_steps;
      const unsigned long long givers = __ballot(can_give);
      const int n_idle = (int)__popcll(idle), n_give = (int)__popcll(givers);
      const int n = n_idle < n_give ? n_idle : n_give;
      if (DR_WAVE_LOG_DETAIL && round == 0) {
        d_share_iters++; d_want_give += (unsigned long long)n_give; d_idle += (unsigned long long)n_idle;
        d_owner_walk += (unsigned long long)__popcll(__ballot(tr.node >= 0 && px >= 0));
        d_wait_owner += (unsigned long long)__popcll(__ballot(tr.node == -1 && share == 64));
        d_pending += (unsigned long long)__popcll(__ballot((tr.node == -1 && share != 64) || tr.node == -2));
      }
      if (n == 0) break;
      {
        if (DR_WAVE_LOG_DETAIL) d_given += (unsigned long long)n;
        int* const xch = my_lds + WIDE_STACK * 64;                 // the phase stash is free between phases: 8 words per hand-over
        const int rank_g = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(givers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)givers, 0u));
        const int rank_i = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
        if (can_give && rank_g < n) {
          const int root = share >= 0 && share < 64 ? share : lane;               // the lane whose pixel this ray belongs to
          if (share < 0) { share_key[lane] = hit_key(tr.best_t, tr.best_slot); share_pend[lane] = 0u; share = 64; }
          atomicAdd(&share_pend[root], 1u);
          unsigned word;
          if (ws.sp > ws.sb) { word = (unsigned)my_stack[ws.sb * 64]; ws.sb++; }
          else { word = ws.top; ws.top = 0u; }
          xch[rank_g + 0 * 64] = __float_as_int(path.rayo.x); xch[rank_g + 1 * 64] = __float_as_int(path.rayo.y); xch[rank_g + 2 * 64] = __float_as_int(path.rayo.z);
          xch[rank_g + 3 * 64] = __float_as_int(path.raydir.x); xch[rank_g + 4 * 64] = __float_as_int(path.raydir.y); xch[rank_g + 5 * 64] = __float_as_int(path.raydir.z);
          xch[rank_g + 6 * 64] = (int)word; xch[rank_g + 7 * 64] = root;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tr.node == -3 && rank_i < n) {
          path.rayo = mk(__int_as_float(xch[rank_i + 0 * 64]), __int_as_float(xch[rank_i + 1 * 64]), __int_as_float(xch[rank_i + 2 * 64]));
          path.raydir = mk(__int_as_float(xch[rank_i + 3 * 64]), __int_as_float(xch[rank_i + 4 * 64]), __int_as_float(xch[rank_i + 5 * 64]));
          ws.top = (unsigned)xch[rank_i + 6 * 64]; ws.sp = 0; ws.sb = 0;
          share = xch[rank_i + 7 * 64];
          inv = mk(1.0f / path.raydir.x, 1.0f / path.raydir.y, 1.0f / path.raydir.z);
          wr = wide_ray(path.rayo, inv, P.wide_pmax);
          const unsigned long long k = share_key[share];
          tr.best_t = __uint_as_float((unsigned)(k >> 32)); tr.best_slot = (int)(unsigned)k;
          wide_pop(tr, ws, my_stack);                              // the first pending child of the word
          rstart = steps - (unsigned)P.coop_steps;                 // a helper may hand on at once
        }
        __builtin_amdgcn_wave_barrier();                           // the exchange words are read before a phase (or the next round) may overwrite them
      }
      }
      // lanes of a shared ray: publish an improvement, take over a better bound
      if (tr.node >= 0 && share >= 0) {
        const int root = share < 64 ? share : lane;
        const unsigned long long mine = hit_key(tr.best_t, tr.best_slot), k = share_key[root];
        if (mine < k) atomicMin(&share_key[root], mine);
        else if (k < mine) { tr.best_t = __uint_as_float((unsigned)(k >> 32)); tr.best_slot = (int)(unsigned)k; }
      }
    }
    if (!WIDE && COOP && !COUNT && P.coop_steps > 0 && cur_tile >= ntiles && (int)__popcll(walking) <= P.coop_lanes) {
      // ---- threaded walk, draining: a ray that is already old is finished by the whole wave at once (coop_closest_hit)
      unsigned long long cand = __ballot(tr.node >= 0 && !(PARK_MIN > 0 && pk.parked) && (int)(steps - rstart) >= P.coop_steps);
      while (cand != 0ull) {
        const int L = __ffsll((long long)cand) - 1;
        cand &= cand - 1ull;
        auto bcast = [L](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L)); };
        const V3 uo = mk(bcast(path.rayo.x), bcast(path.rayo.y), bcast(path.rayo.z));
        const V3 ud = mk(bcast(path.raydir.x), bcast(path.raydir.y), bcast(path.raydir.z));
        const V3 ui = mk(bcast(inv.x), bcast(inv.y), bcast(inv.z));
        Hit r;
        const bool done = coop_closest_hit(P.pairs, P.prims, uo, ud, ui, bcast(tr.best_t), __builtin_amdgcn_readlane(tr.best_slot, L), my_lds, r);
        if (lane == L) {
          if (done) { tr.best_t = r.t; tr.best_slot = r.slot; tr.node = -1; }
          else rstart = 0x80000000u;              // stack overflow: never ask again for this ray ((int)(steps - rstart) is negative from now on)
        }
      }
    }
    if (WIDE) {
      // ---- one record per walking lane.  Lanes at a leaf (exact box + primitive: the long block) wait until enough of
      // them stand at one, or nobody can take a node step; when they go, they fetch together with the lanes at nodes, so
      // the wave waits for one round trip, not two.
      const bool at_leaf = tr.node >= 0 && (tr.node & 1);
      const unsigned long long leaves = __ballot(at_leaf);
      const unsigned long long nodes = __ballot(tr.node >= 0 && !(tr.node & 1));
      const bool do_leaves = leaves != 0ull && ((int)__popcll(leaves) >= (PARK_MIN > 0 ? PARK_MIN : 1) || nodes == 0ull || cur_tile >= ntiles || held);
      if (COUNT) { n_leafstep += do_leaves; n_nodestep += nodes != 0ull; }
      if (tr.node >= 0 && (!at_leaf || do_leaves)) {
        if (COUNT) { if (first_active_lane()) c.trav_slots += 64; }
        const WideRec r = wide_fetch(walk, tr.node);
        if (at_leaf) wide_leaf_compute<COUNT>(r, path.rayo, path.raydir, inv, tr, ws, my_stack, c);
        else wide_node_compute<COUNT>(r, path.rayo, inv, wr, tr, ws, my_stack, c);
        steps++;
      }
#if DR_MERGED_STEPS
      for (int u = 1; u < P_UNROLL; u++) {         // the further steps of an iteration take leaf lanes along too
        const bool at_leaf2 = tr.node >= 0 && (tr.node & 1);
        const unsigned long long leaves2 = __ballot(at_leaf2);
        const unsigned long long nodes2 = __ballot(tr.node >= 0 && !(tr.node & 1));
        const bool do_leaves2 = leaves2 != 0ull && ((int)__popcll(leaves2) >= (PARK_MIN > 0 ? PARK_MIN : 1) || nodes2 == 0ull || cur_tile >= ntiles || held);
        if (COUNT) { n_leafstep += do_leaves2; n_nodestep += nodes2 != 0ull; }
        if (tr.node >= 0 && (!at_leaf2 || do_leaves2)) {
          if (COUNT) { if (first_active_lane()) c.trav_slots += 64; }
          const WideRec r = wide_fetch(walk, tr.node);
          if (at_leaf2) wide_leaf_compute<COUNT>(r, path.rayo, path.raydir, inv, tr, ws, my_stack, c);
          else wide_node_compute<COUNT>(r, path.rayo, inv, wr, tr, ws, my_stack, c);
          steps++;
        }
      }
#else
      for (int u = 1; u < P_UNROLL; u++) {
        if (COUNT) n_nodestep += __ballot(tr.node >= 0 && !(tr.node & 1)) != 0ull;
        if (tr.node >= 0 && !(tr.node & 1)) {
          if (COUNT) { if (first_active_lane()) c.trav_slots += 64; }
          wide_node_step<COUNT>(walk, path.rayo, inv, wr, tr, ws, my_stack, c);
          steps++;
        }
      }
#endif
    } else if (PARK_MIN > 0) {
      // ---- test the parked leaves once enough lanes hold one (or nobody could step anyway)
      const unsigned long long parked = __ballot(pk.parked);
      const unsigned long long steppers = __ballot(tr.node >= 0 && !pk.parked);
      // (once the queue is empty the wave only drains: waiting for company just lengthens the tail)
      if (parked != 0ull && (__popcll(parked) >= PARK_MIN || steppers == 0ull || cur_tile >= ntiles)) {
        if (pk.parked) parked_test<COUNT>(path.rayo, path.raydir, tr, pk, c);
      }
      // ---- UNROLL node steps for every lane that is walking and not parked (the bookkeeping above
      // is then paid once per UNROLL steps; a lane that parks or finishes sits out the rest)
      for (int u = 0; u < P_UNROLL; u++) {     // P_UNROLL is a template constant: fully unrolled by the optimizer
        if (tr.node >= 0 && !pk.parked) {
          if (COUNT) { if (first_active_lane()) c.trav_slots += 64; }
          trav_step_park<COUNT>(walk, path.rayo, inv, tr, pk, c);
          steps++;
        }
      }
    } else {
      // ---- one node step for every walking lane
      if (tr.node >= 0) {
        if (COUNT) { if (first_active_lane()) c.trav_slots += 64; }
        trav_step<COUNT>(walk, path.rayo, path.raydir, inv, tr, c);
        steps++;
      }
    }
  }
  if (lane == 0) {
    const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
    atomicAdd(&P.counters[8], __builtin_readcyclecounter() - t_begin);
    atomicAdd(&P.counters[15], r_end - r_begin);      // 100 MHz ticks: wave cycles / this = shader clock / 100 MHz
    if (WAVE_LOG && P.wave_log) {
      unsigned long long* const w = P.wave_log + (size_t)wave_id * 16;
      w[0] = r_begin; w[1] = r_empty; w[2] = r_end; w[3] = n_after; w[4] = d_phases; w[5] = d_given; w[6] = d_walking; w[7] = d_phase_ticks;
      w[8] = d_want_give; w[9] = d_idle; w[10] = d_owner_walk; w[11] = d_share_iters; w[12] = d_wait_owner; w[13] = d_pending; w[14] = d_iters;
      if (DR_WAVE_LOG_DETAIL) { w[4] = d_home_tiles; w[5] = d_stolen_tiles; w[6] = d_left_home; w[15] = (unsigned long long)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u); }
    }
  }
  if (COUNT && lane == 0) {
    atomicAdd(&P.counters[9], t_phase);
    atomicAdd(&P.counters[10], n_iter);
    atomicAdd(&P.counters[11], n_phase);
    atomicAdd(&P.counters[12], n_nodestep);
    atomicAdd(&P.counters[13], n_leafstep);
    atomicAdd(&P.counters[14], n_shaded);
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

// Two paths per lane (wide walk, long launches).
//
// In the kernel above a lane owns ONE path: when its walk ends it idles until the wave's next shade/refill phase, and a
// phase runs as soon as half the lanes idle -- measured on the bench scene, a node step executes for 26 of 64 lanes and a
// phase shades 34.  Here every lane owns TWO paths (records in a per-wave region of global memory, 112 bytes each): while
// one is walked (its ray, stack and best hit in registers, nothing else), the other waits to be shaded or holds the next
// ray, already made.  A walk that ends leaves (t, slot) in its record and the lane starts on its other path at once; the
// phase runs when THRESH lanes have a path to service (or nobody can walk) and shades ONE path of every such lane -- at
// 48-64 lanes instead of 34, and the node loop keeps nearly all lanes.  The arithmetic of a path is untouched (same
// functions, same order of draws), which lane or slot carries it does not enter it: frames are identical to the other
// kernels'.  Option "paired" = 1 selects it for launches with many tiles per wave.  MEASURED (bench scene, 32 frames per launch):
// node steps run for 34 lanes instead of 26 and a phase serves 41 paths instead of 34, but the lanes that wait at leaves
// are as many as before, the records travel through memory and the phase spills (96 VGPRs + 88 bytes of scratch):
// 0.86 ms/frame against 0.69.  Kept as an option, off by default.
constexpr int PATH_UNITS = 7;             // 16-byte units per path record
constexpr int PAIR_STASH = 12;
constexpr int PAIR_IDLE_MAX = 16;         // ... or as soon as this many lanes have nothing to walk            // dwords per lane stashed during a phase (behind the WIDE_STACK stack words)
enum { PS_EMPTY = 0, PS_READY = 1, PS_WALK = 2, PS_DONE = 3, PS_RETIRED = 4 };

template <int OCC, int THRESH, int PARK_MIN, int P_UNROLL>
__global__ __launch_bounds__(256, OCC) void render_paired_kernel(RenderParams P, unsigned* tile_counter, const int* __restrict__ tile_order,
                                                                 const int* __restrict__ region_start, unsigned* __restrict__ pixel_cost,
                                                                 float4* __restrict__ paths) {
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  constexpr int REGION = (WIDE_STACK + PAIR_STASH) * 64;
  __shared__ int wave_lds[4 * REGION];
  int* const my_lds = wave_lds + (threadIdx.x >> 6) * REGION;
  int* const my_stack = my_lds + lane;
  float* const st = reinterpret_cast<float*>(my_lds) + WIDE_STACK * 64 + lane;
  float4* const my_paths = paths + ((size_t)wave_id * 64 + (size_t)lane) * 2 * PATH_UNITS;   // this lane's two records
  const int ntiles = P.ncols * P.gy;
  const WalkRsrc walk = wide_rsrc(P);
  Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
  int cur_tile = ntiles, cur_frame = 0, cur_next = 64;
  int region = 0, regions_left = P.regions;
  if (P.regions > 1) region = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) % (unsigned)P.regions);   // HW_REG_XCC_ID
  // the walking path: ray, traversal state, steps of this ray
  Trav tr; tr.node = -1; tr.best_t = 0; tr.best_slot = -1;
  WideStack ws; ws.top = 0u; ws.sp = 0; ws.sb = 0;
  V3 wo = mk(0, 0, 0), wd = mk(0, 0, 0), inv = mk(0, 0, 0);
  WideRay wr; wr.inv = wr.marg = mk(0, 0, 0);
  unsigned wsteps = 0;
  int cur = -1;                           // slot being walked, -1 none
  int st0 = PS_EMPTY, st1 = PS_EMPTY;     // state of this lane's two paths
  V3 so = mk(0, 0, 0), sd = mk(0, 0, 0);  // the ray of the path that is PS_READY (at most one: a lane that is not walking starts on it at once)
  const unsigned long long t_begin = __builtin_readcyclecounter(), r_begin = __builtin_amdgcn_s_memrealtime();
  unsigned n_iter = 0, n_phase = 0, n_nodestep = 0, n_leafstep = 0, n_served = 0;      // wave-level tallies (scalar registers), written once at the end
  for (;;) {
    n_iter++;
    // ---- a walk that has ended leaves its hit in the record; the lane turns to its other path if that has a ray
    if (cur >= 0 && tr.node == -1) {
      float* rec = reinterpret_cast<float*>(my_paths + cur * PATH_UNITS);
      rec[6] = tr.best_slot < 0 ? -1.0f : tr.best_t;
      rec[7] = __int_as_float(tr.best_slot);
      rec[26] = __uint_as_float(wsteps);
      if (cur == 0) st0 = PS_DONE; else st1 = PS_DONE;
      cur = -1;
    }
    if (cur < 0 && (st0 == PS_READY || st1 == PS_READY)) {
      cur = st0 == PS_READY ? 0 : 1;
      if (cur == 0) st0 = PS_WALK; else st1 = PS_WALK;
      wo = so; wd = sd;
      inv = mk(1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z);
      wr = wide_ray(wo, inv, P.wide_pmax);
      trav_begin(tr);
      ws.top = 0u; ws.sp = 0;
      wsteps = 0;
    }
    const bool serviceable0 = st0 == PS_DONE || st0 == PS_EMPTY, serviceable1 = st1 == PS_DONE || st1 == PS_EMPTY;
    const unsigned long long need = __ballot(serviceable0 || serviceable1);
    const unsigned long long walking = __ballot(cur >= 0);
    // a phase when enough lanes have a path to service, or too many lanes have nothing to walk (both their paths wait)
    if (need != 0ull && ((int)__popcll(need) >= THRESH || (int)__popcll(~walking) >= PAIR_IDLE_MAX || walking == 0ull)) {
      // ================= phase: one path of every lane that has one to service
      const int s = serviceable0 ? 0 : (serviceable1 ? 1 : -1);
      n_phase++; n_served += (unsigned)__popcll(need);
      {
        st[0 * 64] = wo.x; st[1 * 64] = wo.y; st[2 * 64] = wo.z; st[3 * 64] = wd.x; st[4 * 64] = wd.y; st[5 * 64] = wd.z;
        st[6 * 64] = tr.best_t; st[7 * 64] = __int_as_float(tr.best_slot); st[8 * 64] = __int_as_float(tr.node);
        st[9 * 64] = __uint_as_float(ws.top); st[10 * 64] = __int_as_float(ws.sp); st[11 * 64] = __uint_as_float(wsteps);
        asm volatile("" ::: "memory");           // the walking path's state waits in LDS: the shading code is where register pressure peaks
      }
      float4* const rec = my_paths + (s < 0 ? 0 : s) * PATH_UNITS;
      const int sstate = s == 0 ? st0 : st1;
      Path path; path.rayo = mk(0, 0, 0); path.raydir = mk(0, 0, 0); path.atten = mk(0, 0, 0);
      V3 color = mk(0, 0, 0);
      Xorwow rng; rng.v0 = rng.v1 = rng.v2 = rng.v3 = rng.v4 = rng.d = 0;
      int px = -1, py = 0, pcode = 0, frame = 0, bounce = 0, sample = 0;
      unsigned psteps = 0;
      float hit_t = -1.0f; int hit_slot = -1;
      const bool shade_me = s >= 0 && sstate == PS_DONE;
      if (shade_me) {
        const float4 u0 = rec[0], u1 = rec[1], u2 = rec[2], u3 = rec[3], u4 = rec[4], u5 = rec[5], u6 = rec[6];
        path.rayo = mk(u0.x, u0.y, u0.z); path.raydir = mk(u0.w, u1.x, u1.y);
        hit_t = u1.z; hit_slot = __float_as_int(u1.w);
        path.atten = mk(u2.x, u2.y, u2.z); color = mk(u2.w, u3.x, u3.y);
        px = __float_as_int(u3.z); py = __float_as_int(u3.w);
        rng.v0 = __float_as_uint(u4.x); rng.v1 = __float_as_uint(u4.y); rng.v2 = __float_as_uint(u4.z); rng.v3 = __float_as_uint(u4.w);
        rng.v4 = __float_as_uint(u5.x); rng.d = __float_as_uint(u5.y); pcode = __float_as_int(u5.z);
        frame = __float_as_int(u5.w) & 0xffff; bounce = __float_as_int(u5.w) >> 16;
        sample = __float_as_int(u6.x); psteps = __float_as_uint(u6.y) + __float_as_uint(u6.z);
      }
      bool between = s >= 0 && sstate == PS_EMPTY;       // has no path: next sample or a new pixel
      bool has_ray = false;
      if (shade_me) {
        bool ended;
        V3 radiance = mk(0, 0, 0);
        if (hit_slot >= 0 && hit_t > 0.0f) {
          ended = !shade_hit<false>(P, path, hit_t, hit_slot, rng, c, radiance);
          if (!ended) { bounce++; if (bounce >= P.max_depth) ended = true; }        // depth exhausted: black (K:981)
        } else {
          radiance = shade_miss<false>(P, path, c);
          ended = true;
        }
        if (ended) { color = color + radiance; sample++; between = true; }
        else has_ray = true;
      }
      bool want_pixel = false;
      if (between) {
        if (px >= 0 && (float)sample < P.spp_f) {
          // same pixel, next sample (K:1059)
        } else {
          if (px >= 0) {
            store_pixel(P, px, py, color);
            if (pixel_cost) pixel_cost[pcode] = psteps;
          }
          px = -1;
          want_pixel = true;
        }
      }
      unsigned long long ask = __ballot(want_pixel);
      while (ask != 0ull) {
        if (cur_next >= 64) {                      // wave-uniform: fetch the next chunk (tile, frame)
          cur_tile = ntiles;
          while (regions_left > 0) {
            const int r0 = region_start ? region_start[region] : P.region_start[region];
            const int r1 = region_start ? region_start[region + 1] : P.region_start[region + 1];
            unsigned t = 0;
            if (lane == 0) t = atomicAdd(tile_counter + region, 1u);
            const int q = (int)__builtin_amdgcn_readfirstlane(t);
            if (q < (r1 - r0) * P.batch) {
              const int tt = q / P.batch;
              cur_frame = q - tt * P.batch;
              cur_tile = tile_order ? tile_order[r0 + tt] : r0 + tt;
              break;
            }
            region = region + 1 == P.regions ? 0 : region + 1;   // this band is done: help with the next one
            regions_left--;
          }
          cur_next = 0;
        }
        if (cur_tile >= ntiles) break;             // frame exhausted: the lanes still asking retire their path below
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ask, 0u));
        const int avail = 64 - cur_next;
        if (want_pixel && rank < avail) {
          const int l = cur_next + rank;
          const int col = cur_tile / P.gy, by = cur_tile - col * P.gy;
          px = (P.stripe_rem + col * P.stripe_mod) * 8 + (l >> 3);
          py = by * 8 + (l & 7);
          pcode = cur_tile * 64 + l;
          frame = cur_frame;
          psteps = 0;
          sample = 0;
          color = mk(0, 0, 0);
          want_pixel = false;
        }
        const int n = __popcll(ask);
        cur_next += n < avail ? n : avail;
        ask = __ballot(want_pixel);
      }
      if (between && px >= 0) {                    // the next path of this slot: K:1065-1073
        rng.init(sample_seed(P, px, py, sample, frame));
        camera_ray(P, px, py, rng, path.rayo, path.raydir);
        path.atten = splat(1.0f);
        bounce = 0;
        has_ray = true;
      }
      if (s >= 0) {
        const int ns = has_ray ? PS_READY : PS_RETIRED;
        if (s == 0) st0 = ns; else st1 = ns;
        if (has_ray) {
          rec[0] = make_float4(path.rayo.x, path.rayo.y, path.rayo.z, path.raydir.x);
          rec[1] = make_float4(path.raydir.y, path.raydir.z, -1.0f, __int_as_float(-1));
          rec[2] = make_float4(path.atten.x, path.atten.y, path.atten.z, color.x);
          rec[3] = make_float4(color.y, color.z, __int_as_float(px), __int_as_float(py));
          rec[4] = make_float4(__uint_as_float(rng.v0), __uint_as_float(rng.v1), __uint_as_float(rng.v2), __uint_as_float(rng.v3));
          rec[5] = make_float4(__uint_as_float(rng.v4), __uint_as_float(rng.d), __int_as_float(pcode), __int_as_float((frame & 0xffff) | (bounce << 16)));
          rec[6] = make_float4(__int_as_float(sample), __uint_as_float(psteps), __uint_as_float(0u), 0.0f);
          so = path.rayo; sd = path.raydir;
        }
      }
      {
        asm volatile("" ::: "memory");
        wo = mk(st[0 * 64], st[1 * 64], st[2 * 64]); wd = mk(st[3 * 64], st[4 * 64], st[5 * 64]);
        tr.best_t = st[6 * 64]; tr.best_slot = __float_as_int(st[7 * 64]); tr.node = __float_as_int(st[8 * 64]);
        ws.top = __float_as_uint(st[9 * 64]); ws.sp = __float_as_int(st[10 * 64]); wsteps = __float_as_uint(st[11 * 64]);
        inv = mk(1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z);
        wr = wide_ray(wo, inv, P.wide_pmax);
      }
      continue;                                    // lanes that now have a ray start on it at the top of the loop
    }
    if (walking == 0ull) break;                    // nothing walks and nothing can be serviced: every path has retired
    // ---- one record per walking lane (as in the kernel above): lanes at a leaf wait for company
    {
      const bool active = cur >= 0 && tr.node >= 0;
      const bool at_leaf = active && (tr.node & 1);
      const unsigned long long leaves = __ballot(at_leaf);
      const unsigned long long nodes = __ballot(active && !(tr.node & 1));
      const bool do_leaves = leaves != 0ull && ((int)__popcll(leaves) >= (PARK_MIN > 0 ? PARK_MIN : 1) || nodes == 0ull);
      n_leafstep += do_leaves; n_nodestep += nodes != 0ull;
      if (active && (!at_leaf || do_leaves)) {
        const WideRec r = wide_fetch(walk, tr.node);
        if (at_leaf) wide_leaf_compute<false>(r, wo, wd, inv, tr, ws, my_stack, c);
        else wide_node_compute<false>(r, wo, inv, wr, tr, ws, my_stack, c);
        wsteps++;
      }
      for (int u = 1; u < P_UNROLL; u++) {
        n_nodestep += __ballot(cur >= 0 && tr.node >= 0 && !(tr.node & 1)) != 0ull;
        if (cur >= 0 && tr.node >= 0 && !(tr.node & 1)) {
          wide_node_step<false>(walk, wo, inv, wr, tr, ws, my_stack, c);
          wsteps++;
        }
      }
    }
  }
  if (lane == 0) {
    atomicAdd(&P.counters[8], __builtin_readcyclecounter() - t_begin);
    atomicAdd(&P.counters[15], __builtin_amdgcn_s_memrealtime() - r_begin);
    atomicAdd(&P.counters[10], (unsigned long long)n_iter);
    atomicAdd(&P.counters[11], (unsigned long long)n_phase);
    atomicAdd(&P.counters[12], (unsigned long long)n_nodestep);
    atomicAdd(&P.counters[13], (unsigned long long)n_leafstep);
    atomicAdd(&P.counters[14], (unsigned long long)n_served);
  }
}

// Waves with roles (wide walk, long launches; option "roles").
//
// In the kernels above node steps, leaf steps and shading share a wave, and each runs for a fraction of its lanes (26 / 25 / 34
// of 64 on the bench scene).  Here a workgroup is NT trace waves and one shade wave that exchange work through rings in LDS:
//   ray ring   (one, written by the shade wave)        {path, origin, direction}: a trace lane that has finished takes the next ray
//   hit rings  (one per trace wave, read by the shade wave)   {path, t, slot}
// A trace wave is the node / leaf loop and nothing else; a lane refills the moment its walk ends, so the loop stays full.  The
// shade wave takes 64 hits at a time from the rings, loads each path's state (global memory, touched by this wave only),
// shades at full width, and puts the next ray -- or the camera ray of the next sample or pixel, tiles from the same per-XCD
// queues -- into the ray ring.  A workgroup owns RK_PATHS paths; the ray ring holds as many entries, so it never fills, and a
// trace wave waits when its hit ring is full (the shade wave always drains it).  Counters only grow (unsigned differences);
// a ring entry is written before the counter that publishes it (LDS operations of one wave execute in order).  Every
// wait is bounded: a wave that spins too long raises the workgroup's abort flag and everybody leaves (the frame is then
// incomplete: the host checks the flag and fails the call).  The arithmetic of a path is the same functions in the same
// order; which wave carries which part does not enter it.
constexpr unsigned RK_SPIN_LIMIT = 1u << 22;

// NT trace waves + NS shade waves per workgroup ((NT + NS) a multiple of 4).  A path belongs to ONE shade wave (its state is never
// touched by another wave): shade wave s owns paths [s * RS, (s + 1) * RS) of the workgroup, has its own ray ring of RS entries
// (never full) and reads the hit rings [trace wave][s]; a trace lane takes rays from either ray ring and returns the hit to
// the ring of the path's owner.
template <int NT, int NS, int PARK_MIN, int P_UNROLL>
__global__ __launch_bounds__((NT + NS) * 64, 4) void render_roles_kernel(RenderParams P, unsigned* tile_counter, const int* __restrict__ tile_order,
                                                                         const int* __restrict__ region_start, float4* __restrict__ paths,
                                                                         unsigned* __restrict__ abort_flag) {
  constexpr int RS = 512;                 // paths (= ray ring entries) per shade wave
  constexpr int HQ = NS == 1 ? 128 : 64;  // entries per hit ring
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  __shared__ int stacks[NT * WIDE_STACK * 64];
  __shared__ int ray_pid[NS][RS];
  __shared__ float ray_o[NS][3][RS], ray_d[NS][3][RS];
  __shared__ int hit_pid[NT][NS][HQ];
  __shared__ float hit_t[NT][NS][HQ];
  __shared__ int hit_slot[NT][NS][HQ];
  __shared__ unsigned ray_published[NS], ray_claimed[NS], hit_tail[NT][NS], hit_head[NT][NS], done_flag[NS], abort_lds;
  if (threadIdx.x == 0) abort_lds = 0u;
  if (threadIdx.x < NS) { ray_published[threadIdx.x] = 0u; ray_claimed[threadIdx.x] = 0u; done_flag[threadIdx.x] = 0u; }
  if (threadIdx.x < NT * NS) { (&hit_tail[0][0])[threadIdx.x] = 0u; (&hit_head[0][0])[threadIdx.x] = 0u; }
  __syncthreads();
  const unsigned long long t_begin = __builtin_readcyclecounter(), r_begin = __builtin_amdgcn_s_memrealtime();
  Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned n_iter = 0, n_nodestep = 0, n_leafstep = 0, n_phase = 0, n_served = 0;
  auto give_up = [&]() { if (lane == 0) { *(volatile unsigned*)&abort_lds = 1u; atomicExch(abort_flag, 1u); } };

  if (wave < NT) {
    // ================================================================ trace wave
    const WalkRsrc walk = wide_rsrc(P);
    int* const my_stack = stacks + wave * (WIDE_STACK * 64) + lane;
    Trav tr; tr.node = -1; tr.best_t = 0; tr.best_slot = -1;
    WideStack ws; ws.top = 0u; ws.sp = 0; ws.sb = 0;
    V3 wo = mk(0, 0, 0), wd = mk(0, 0, 0), inv = mk(0, 0, 0);
    WideRay wr; wr.inv = wr.marg = mk(0, 0, 0);
    int pid = -1;                       // the path this lane walks for (index within the workgroup), -1 none
    unsigned my_tail[NS];               // this wave's hit rings: entries written so far (wave-uniform)
#pragma unroll
    for (int q = 0; q < NS; q++) my_tail[q] = 0u;
    unsigned spins = 0u;
    bool leave = false;
    for (;;) {
      n_iter++;
      // ---- finished walks go to the hit ring of the path's shade wave
#pragma unroll
      for (int q = 0; q < NS; q++) {
        const bool mine = pid >= 0 && tr.node == -1 && (NS == 1 || pid / RS == q);
        const unsigned long long fin = __ballot(mine);
        if (fin == 0ull) continue;
        const unsigned n = (unsigned)__popcll(fin);
        unsigned head = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)&hit_head[wave][q]);
        while (my_tail[q] - head + n > (unsigned)HQ) {            // the shade wave drains the ring; wave-uniform wait
          __builtin_amdgcn_s_sleep(2);
          if (++spins > RK_SPIN_LIMIT) give_up();
          if (*(volatile unsigned*)&abort_lds) { leave = true; break; }
          head = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)&hit_head[wave][q]);
        }
        if (leave) break;
        if (mine) {
          const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(fin >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fin, 0u));
          const unsigned e = (my_tail[q] + rank) & (HQ - 1);
          hit_t[wave][q][e] = tr.best_slot < 0 ? -1.0f : tr.best_t;
          hit_slot[wave][q][e] = tr.best_slot;
          hit_pid[wave][q][e] = pid;
          pid = -1;
        }
        my_tail[q] += n;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) *(volatile unsigned*)&hit_tail[wave][q] = my_tail[q];       // published after the entries
      }
      if (leave) break;
      // ---- lanes without a ray take the next ones of the ray rings
#pragma unroll
      for (int k = 0; k < NS; k++) {
        const int q = NS == 1 ? 0 : (k ^ (int)((n_iter + (unsigned)wave) & 1u));   // alternate which ring is asked first
        const unsigned long long idle = __ballot(pid < 0);
        if (idle == 0ull) break;
        unsigned base = 0u, m = 0u;
        if (lane == 0) {
          const unsigned want = (unsigned)__popcll(idle);
          for (int tries = 0; tries < 64; tries++) {
            const unsigned cl = *(volatile unsigned*)&ray_claimed[q], pub = *(volatile unsigned*)&ray_published[q];
            const unsigned avail = pub - cl;
            if (avail == 0u) break;
            const unsigned take = want < avail ? want : avail;
            if (atomicCAS(&ray_claimed[q], cl, cl + take) == cl) { base = cl; m = take; break; }
          }
        }
        base = __builtin_amdgcn_readfirstlane(base); m = __builtin_amdgcn_readfirstlane(m);
        if (m > 0u) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
          if (pid < 0 && rank < m) {
            const unsigned e = (base + rank) & (RS - 1);
            pid = ray_pid[q][e];
            wo = mk(ray_o[q][0][e], ray_o[q][1][e], ray_o[q][2][e]); wd = mk(ray_d[q][0][e], ray_d[q][1][e], ray_d[q][2][e]);
            inv = mk(1.0f / wd.x, 1.0f / wd.y, 1.0f / wd.z);
            wr = wide_ray(wo, inv, P.wide_pmax);
            trav_begin(tr);
            ws.top = 0u; ws.sp = 0; ws.sb = 0;
          }
        }
      }
      const unsigned long long walking = __ballot(pid >= 0 && tr.node >= 0);
      if (walking == 0ull) {
        bool all_done = true;
#pragma unroll
        for (int q = 0; q < NS; q++) all_done = all_done && *(volatile unsigned*)&done_flag[q] != 0u;
        if (all_done || *(volatile unsigned*)&abort_lds) break;
        __builtin_amdgcn_s_sleep(4);
        if (++spins > RK_SPIN_LIMIT) { give_up(); break; }
        continue;
      }
      spins = 0u;
      // ---- one record per walking lane; lanes at a leaf wait for company (as in render_persistent_kernel)
      {
        const bool active = pid >= 0 && tr.node >= 0;
        const bool at_leaf = active && (tr.node & 1);
        const unsigned long long leaves = __ballot(at_leaf);
        const unsigned long long nodes = __ballot(active && !(tr.node & 1));
        const bool do_leaves = leaves != 0ull && ((int)__popcll(leaves) >= (PARK_MIN > 0 ? PARK_MIN : 1) || nodes == 0ull);
        n_leafstep += do_leaves; n_nodestep += nodes != 0ull;
        if (active && (!at_leaf || do_leaves)) {
          const WideRec r = wide_fetch(walk, tr.node);
          if (at_leaf) wide_leaf_compute<false>(r, wo, wd, inv, tr, ws, my_stack, c);
          else wide_node_compute<false>(r, wo, inv, wr, tr, ws, my_stack, c);
        }
        for (int u = 1; u < P_UNROLL; u++) {
          n_nodestep += __ballot(pid >= 0 && tr.node >= 0 && !(tr.node & 1)) != 0ull;
          if (pid >= 0 && tr.node >= 0 && !(tr.node & 1)) wide_node_step<false>(walk, wo, inv, wr, tr, ws, my_stack, c);
        }
      }
    }
  } else {
    // ================================================================ shade wave
    const int sq = wave - NT;            // which shade wave: owns paths [sq * RS, (sq + 1) * RS) and ray ring sq
    const int ntiles = P.ncols * P.gy;
    int cur_tile = ntiles, cur_frame = 0, cur_next = 64;
    int region = 0, regions_left = P.regions;
    if (P.regions > 1) region = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) % (unsigned)P.regions);   // HW_REG_XCC_ID
    float4* const my_paths = paths + ((size_t)blockIdx.x * NS + (size_t)sq) * RS * PATH_UNITS;
    unsigned my_head[NT];
#pragma unroll
    for (int w = 0; w < NT; w++) my_head[w] = 0u;
    unsigned pub = 0u;                  // rays published so far
    int retired = 0, started = 0;       // paths of this wave that have ended for good / that have been given their first pixel
    unsigned spins = 0u;
    for (;;) {
      n_iter++;
      // ---- up to 64 hits from the rings, or (first) paths that have never had a pixel
      int hpid = -1; float hit_tv = -1.0f; int hit_sv = -1;
      bool fresh = false;               // a path that starts: no hit to shade
      if (started < RS) {
        hpid = started + lane;
        fresh = hpid < RS;
        if (!fresh) hpid = -1;
        started += 64;
      } else {
        int taken = 0;
#pragma unroll
        for (int w = 0; w < NT; w++) {
          const unsigned tail = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)&hit_tail[w][sq]);
          const int avail = (int)(tail - my_head[w]);
          const int take = avail < 64 - taken ? avail : 64 - taken;
          if (take > 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lane >= taken && lane < taken + take) {
              const unsigned e = (my_head[w] + (unsigned)(lane - taken)) & (HQ - 1);
              hpid = hit_pid[w][sq][e] - sq * RS; hit_tv = hit_t[w][sq][e]; hit_sv = hit_slot[w][sq][e];
            }
            my_head[w] += (unsigned)take;
            taken += take;
          }
        }
        if (taken > 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the entries are read before their slots are handed back
#pragma unroll
          for (int w = 0; w < NT; w++) if (lane == w) *(volatile unsigned*)&hit_head[w][sq] = my_head[w];
        }
      }
      const unsigned long long have = __ballot(hpid >= 0);
      if (have == 0ull) {
        if (retired >= RS) { if (lane == 0) *(volatile unsigned*)&done_flag[sq] = 1u; break; }
        if (*(volatile unsigned*)&abort_lds) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > RK_SPIN_LIMIT) { give_up(); break; }
        continue;
      }
      spins = 0u;
      n_phase++; n_served += (unsigned)__popcll(have);
      float4* const rec = my_paths + (size_t)(hpid < 0 ? 0 : hpid) * PATH_UNITS;
      Path path; path.rayo = mk(0, 0, 0); path.raydir = mk(0, 0, 0); path.atten = mk(0, 0, 0);
      V3 color = mk(0, 0, 0);
      Xorwow rng; rng.v0 = rng.v1 = rng.v2 = rng.v3 = rng.v4 = rng.d = 0;
      int px = -1, py = 0, frame = 0, bounce = 0, sample = 0;
      const bool shade_me = hpid >= 0 && !fresh;
      if (shade_me) {
        const float4 u0 = rec[0], u1 = rec[1], u2 = rec[2], u3 = rec[3], u4 = rec[4], u5 = rec[5], u6 = rec[6];
        path.rayo = mk(u0.x, u0.y, u0.z); path.raydir = mk(u0.w, u1.x, u1.y);
        path.atten = mk(u2.x, u2.y, u2.z); color = mk(u2.w, u3.x, u3.y);
        px = __float_as_int(u3.z); py = __float_as_int(u3.w);
        rng.v0 = __float_as_uint(u4.x); rng.v1 = __float_as_uint(u4.y); rng.v2 = __float_as_uint(u4.z); rng.v3 = __float_as_uint(u4.w);
        rng.v4 = __float_as_uint(u5.x); rng.d = __float_as_uint(u5.y);
        frame = __float_as_int(u5.w) & 0xffff; bounce = __float_as_int(u5.w) >> 16;
        sample = __float_as_int(u6.x);
      }
      bool between = fresh;
      bool has_ray = false;
      if (shade_me) {
        bool ended;
        V3 radiance = mk(0, 0, 0);
        if (hit_sv >= 0 && hit_tv > 0.0f) {
          ended = !shade_hit<false>(P, path, hit_tv, hit_sv, rng, c, radiance);
          if (!ended) { bounce++; if (bounce >= P.max_depth) ended = true; }        // depth exhausted: black (K:981)
        } else {
          radiance = shade_miss<false>(P, path, c);
          ended = true;
        }
        if (ended) { color = color + radiance; sample++; between = true; }
        else has_ray = true;
      }
      bool want_pixel = false;
      if (between) {
        if (px >= 0 && (float)sample < P.spp_f) {
          // same pixel, next sample (K:1059)
        } else {
          if (px >= 0) store_pixel(P, px, py, color);
          px = -1;
          want_pixel = true;
        }
      }
      unsigned long long ask = __ballot(want_pixel);
      while (ask != 0ull) {
        if (cur_next >= 64) {                      // wave-uniform: fetch the next chunk (tile, frame)
          cur_tile = ntiles;
          while (regions_left > 0) {
            const int r0 = region_start ? region_start[region] : P.region_start[region];
            const int r1 = region_start ? region_start[region + 1] : P.region_start[region + 1];
            unsigned t = 0;
            if (lane == 0) t = atomicAdd(tile_counter + region, 1u);
            const int q = (int)__builtin_amdgcn_readfirstlane(t);
            if (q < (r1 - r0) * P.batch) {
              const int tt = q / P.batch;
              cur_frame = q - tt * P.batch;
              cur_tile = tile_order ? tile_order[r0 + tt] : r0 + tt;
              break;
            }
            region = region + 1 == P.regions ? 0 : region + 1;   // this band is done: help with the next one
            regions_left--;
          }
          cur_next = 0;
        }
        if (cur_tile >= ntiles) break;             // frame exhausted: the paths still asking retire below
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ask, 0u));
        const int avail = 64 - cur_next;
        if (want_pixel && rank < avail) {
          const int l = cur_next + rank;
          const int col = cur_tile / P.gy, by = cur_tile - col * P.gy;
          px = (P.stripe_rem + col * P.stripe_mod) * 8 + (l >> 3);
          py = by * 8 + (l & 7);
          frame = cur_frame;
          sample = 0;
          color = mk(0, 0, 0);
          want_pixel = false;
        }
        const int n = __popcll(ask);
        cur_next += n < avail ? n : avail;
        ask = __ballot(want_pixel);
      }
      if (between && px >= 0) {                    // the next path of this slot: K:1065-1073
        rng.init(sample_seed(P, px, py, sample, frame));
        camera_ray(P, px, py, rng, path.rayo, path.raydir);
        path.atten = splat(1.0f);
        bounce = 0;
        has_ray = true;
      }
      retired += (int)__popcll(__ballot(hpid >= 0 && !has_ray));
      // ---- state back to memory, rays into the ring
      if (has_ray) {
        rec[0] = make_float4(path.rayo.x, path.rayo.y, path.rayo.z, path.raydir.x);
        rec[1] = make_float4(path.raydir.y, path.raydir.z, 0.0f, 0.0f);
        rec[2] = make_float4(path.atten.x, path.atten.y, path.atten.z, color.x);
        rec[3] = make_float4(color.y, color.z, __int_as_float(px), __int_as_float(py));
        rec[4] = make_float4(__uint_as_float(rng.v0), __uint_as_float(rng.v1), __uint_as_float(rng.v2), __uint_as_float(rng.v3));
        rec[5] = make_float4(__uint_as_float(rng.v4), __uint_as_float(rng.d), 0.0f, __int_as_float((frame & 0xffff) | (bounce << 16)));
        rec[6] = make_float4(__int_as_float(sample), 0.0f, 0.0f, 0.0f);
      }
      const unsigned long long push = __ballot(has_ray);
      if (push != 0ull) {
        if (has_ray) {
          const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(push >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)push, 0u));
          const unsigned e = (pub + rank) & (RS - 1);
          ray_pid[sq][e] = hpid + sq * RS;
          ray_o[sq][0][e] = path.rayo.x; ray_o[sq][1][e] = path.rayo.y; ray_o[sq][2][e] = path.rayo.z;
          ray_d[sq][0][e] = path.raydir.x; ray_d[sq][1][e] = path.raydir.y; ray_d[sq][2][e] = path.raydir.z;
        }
        pub += (unsigned)__popcll(push);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) *(volatile unsigned*)&ray_published[sq] = pub;
      }
    }
  }
  if (lane == 0) {
    atomicAdd(&P.counters[8], __builtin_readcyclecounter() - t_begin);
    atomicAdd(&P.counters[15], __builtin_amdgcn_s_memrealtime() - r_begin);
    atomicAdd(&P.counters[10], (unsigned long long)n_iter);
    atomicAdd(&P.counters[11], (unsigned long long)n_phase);
    atomicAdd(&P.counters[12], (unsigned long long)n_nodestep);
    atomicAdd(&P.counters[13], (unsigned long long)n_leafstep);
    atomicAdd(&P.counters[14], (unsigned long long)n_served);
  }
}

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
    h = closest_hit_wide<true>(wide_rsrc(P), P.wide_pmax, ld3(o + 3 * i), ld3(d + 3 * i), c, stack);
  } else {
    h = closest_hit_threaded<true>(walk_rsrc(P), ld3(o + 3 * i), ld3(d + 3 * i), c);
  }
  t[i] = h.t; slot[i] = h.slot;
  if (visits) visits[i] = (int32_t)c.V;
}

}  // namespace dr

// ------------------------------------------------------------------ context
using namespace dr;

struct dr_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // dr_render_accumulate_async: two batches may be in flight, each with its own pair of events
  hipEvent_t pev0[2] = {nullptr, nullptr}, pev1[2] = {nullptr, nullptr};
  bool pending[2] = {false, false}; uint64_t pending_frames[2] = {0, 0}, pending_samples[2] = {0, 0}; int pending_next = 0;
  // resident scene
  DevUnit* walk = nullptr; size_t walk_bytes = 0;
  DevUnit* wide = nullptr; size_t wide_bytes = 0; int wide_depth = 0, wide_nodes = 0; float wide_pmax = 0;   // null: scene not representable (threaded walk is used)
  int wide_tree = 1;        // structure of the wide walk's tree: 1 binned SAH (default), 0 the reference's topology collapsed
  DevPair* pairs = nullptr;
  DevPrim* prims = nullptr;
  DevShade* shade = nullptr;
  DevTex* tex = nullptr;
  uint32_t* texels = nullptr;
  int n_prims = 0, n_tex = 0, tree_depth = 0;
  std::vector<int> slot_to_orig;
  // frame + accumulator
  int32_t* frame = nullptr; size_t frame_elems = 0;
  int32_t* accum = nullptr; size_t accum_elems = 0; int accW = 0, accH = 0;
  uint8_t* present = nullptr; size_t present_bytes = 0;
  // multi-GPU gather: two packed copies of this context's stripe (double buffer), sized for the accumulator
  int32_t* packed[2] = {nullptr, nullptr}; size_t packed_elems[2] = {0, 0};
  float4* paths = nullptr; size_t paths_waves = 0;      // render_paired_kernel: two path records per lane
  unsigned* abort_flag = nullptr;                        // render_roles_kernel: set by a wave that waited too long (protocol failure)
  bool roles_used = false;
  int roles = 0;            // wide walk, long launches: 3 / 7 = workgroups of that many trace waves + one shade wave, 6 = 6 + 2 (render_roles_kernel)
  int paired = 0;           // wide walk, long launches: 1 = two paths per lane (render_paired_kernel: measured slower), 0 = the one-path kernel
  int pair_thresh = 48;     // ... phase once this many lanes have a path to service (32, 48 or 56)
  unsigned long long* counters = nullptr;
  unsigned* tile_counters = nullptr; int tile_cursor = 0; int num_cus = 256;
  // cost feedback (persistent kernel): per-pixel cost of the last frame, per-tile cost, tile order
  unsigned* pixel_cost = nullptr; unsigned* tile_cost = nullptr; int* tile_order = nullptr; int* region_start = nullptr;
  int order_capacity = 0;          // tiles the three buffers are sized for
  bool order_valid = false;        // tile_order was computed for `order_key`
  int order_age = 0;               // launches since the view (order_key) changed
  int order_follows_camera = 1;    // a view with the same frame geometry but other settings starts from the previous view's tile order
  int feedback_every = 8;          // ... the order is recomputed after the first two of them and then after every feedback_every-th
  float order_key[18] = {0};       // settings13 + W, H, stripe, tile grid of the frame the order belongs to
  bool feedback = true;
  int stripe_mod = 1, stripe_rem = 0;
  int traversal = DR_TRAVERSAL_WIDE;
  bool count = false;
  // tunables (dr_context_set_option / DOGERAY_OPTIONS)
  int kernel = DR_KERNEL_PERSISTENT;
  int occupancy = 6;        // waves per SIMD the kernel is built and launched for (persistent: 4, 5, or 6 = six for the lean wide build and five for the others; tile kernel: 4 or 6)
  int trav_min = 32;        // persistent kernel: shade/refill once fewer lanes than this are walking
  int park_min = 16;        // persistent kernel: leaf steps (parked leaves) once this many lanes stand at one (0 = on the spot)
  int unroll = 2;           // persistent kernel: node steps per loop iteration
  int xcd_regions = 1;      // persistent kernel: one tile queue per XCD (image bands), with stealing
  int heavy_factor = 1;     // tile order: tiles costlier than this x the mean start first, the rest keep their natural order (0 = all natural, -1 = all by cost)
  int coop_steps = 2;       // persistent kernel, drain phase: rays older than this many steps are shared with idle lanes / finished cooperatively (0 = off)
  int coop_tiles_per_wave = 32;   // wide walk: launches with fewer tiles per wave than this run the build with the work-sharing drain
  int coop_lanes = 8;       // ... in waves with at most this many lanes still walking
  int split_parts = 4;      // short launches: the tiles with last frame's longest pixels are handed out in this many parts (1, 2, 4, 8), the rest of each wave helps
  int split_waves = 12;     // ... as many of them as give this many percent of the waves a part to start with
  int split_steps = 400;    // ... tiles whose longest pixel took at least this many node steps (multiple of 16)
  int short_one_queue = 1;  // short launches use one tile queue instead of one per XCD
  int coop_rounds = 2;      // work sharing: hand-over rounds per loop iteration
  int wave_log_on = 0;      // persistent kernel writes begin / queue-empty / end stamps of every wave (dr_stats_wave_log)
  unsigned long long* wave_log = nullptr; int wave_log_waves = 0;
  int batch_frames = 32;    // persistent kernel: at most this many frames per launch in dr_render_accumulate
  float cur_settings[13] = {0};
  dr_stats stats;
};

namespace {

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                        \
      return DR_ERR_DEVICE;                                                                \
    }                                                                                      \
  } while (0)

template <class T>
int upload(T*& dst, const std::vector<T>& src) {
  if (dst) { (void)hipFree(dst); dst = nullptr; }
  size_t bytes = src.size() * sizeof(T);
  if (bytes == 0) bytes = sizeof(T);
  HIP_TRY(hipMalloc((void**)&dst, bytes));
  if (!src.empty()) HIP_TRY(hipMemcpy(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return DR_OK;
}

int ensure(int32_t*& buf, size_t& have, size_t need) {
  if (have >= need && buf) return DR_OK;
  if (buf) { (void)hipFree(buf); buf = nullptr; have = 0; }
  HIP_TRY(hipMalloc((void**)&buf, need * sizeof(int32_t)));
  have = need;
  return DR_OK;
}

struct V3h { float x, y, z; };
inline V3h hsub(V3h a, V3h b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3h hmul(V3h a, V3h b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3h hdiv(V3h a, V3h b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3h hsplat(float a) { return {a, a, a}; }
inline float hdot(V3h a, V3h b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3h hcross(V3h a, V3h b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3h hnorm(V3h v) { float inv = 1.0f / sqrtf(hdot(v, v)); return {v.x * inv, v.y * inv, v.z * inv}; }
inline void st3(float* d, V3h v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline int hf2i(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
}

// settings[13] -> per-launch constants.  The camera block is K:1016-1052, evaluated once on the
// host (it is identical for every pixel) with the reference's float/double promotions.
int make_params(dr_context* c, const float* st, int W, int H, float background, uint64_t seed, RenderParams& P, int batch_hint = 1) {
  if (!c->walk) { set_error("no scene uploaded"); return DR_ERR_INVALID; }
  if (W <= 0 || H <= 0 || W > 65000 || H > 65000 || (size_t)W * (size_t)H > (size_t)1 << 28) { set_error("bad frame size"); return DR_ERR_INVALID; }      // (x and y share a word in the phase stash)
  const int div = hf2i(st[11]);
  if (div < 1) { set_error("divisor must be >= 1"); return DR_ERR_INVALID; }
  const int backtex = hf2i(st[12]);
  if (backtex >= c->n_tex) { set_error("backtex refers to a texture that is not loaded"); return DR_ERR_INVALID; }
  memset(&P, 0, sizeof(P));
  memcpy(c->cur_settings, st, sizeof(c->cur_settings));
  P.walk = c->walk; P.walk_bytes = (uint32_t)c->walk_bytes; P.pairs = c->pairs; P.prims = c->prims; P.shade = c->shade; P.tex = c->tex; P.texels = c->texels;
  P.wide = c->wide; P.wide_bytes = (uint32_t)c->wide_bytes; P.wide_pmax = c->wide_pmax;
  P.counters = c->counters;
  P.wave_log = c->wave_log_on ? c->wave_log : nullptr;
  float aspect = float(W / st[11]) / float(H / st[11]);           // K:1016 (int / float)
  float fov = (float)((double)st[8] * M_PI / 180);                // K:1020
  float vh = (float)(2.0 * (double)tanf(fov / 2));                // K:1023
  float vw = aspect * vh;
  V3h from = {st[0], st[1], st[2]}, at = {st[3], st[4], st[5]};
  float focus = st[7];
  V3h vup = {0, 1, 0};
  V3h wu = hnorm(hsub(from, at));
  V3h uu = hnorm(hcross(vup, wu));
  V3h vu = hcross(wu, uu);
  V3h hor = hmul(hmul(hsplat(focus), hsplat(vw)), uu);            // K:1047
  V3h ver = hmul(hmul(hsplat(focus), hsplat(vh)), vu);
  V3h llc = hsub(hsub(hsub(from, hdiv(hor, hsplat(2))), hdiv(ver, hsplat(2))), hmul(hsplat(focus), wu));
  st3(P.from, from); st3(P.llc, llc); st3(P.hor, hor); st3(P.ver, ver); st3(P.uu, uu); st3(P.vu, vu);
  P.lens_radius = st[6] / 2;                                      // K:1052
  P.bgint = background;
  P.spp_f = st[10];
  P.scale = (float)(1.0 / (double)st[10]);                        // K:1081
  P.den_w = (double)float(W / st[11]);                            // K:1067
  P.den_h = (double)float(H / st[11]);
  P.seed = seed;
  P.W = W; P.H = H;
  P.gx = W / div / 8; P.gy = H / div / 8;                         // K:2636
  P.stripe_mod = c->stripe_mod; P.stripe_rem = c->stripe_rem;
  P.ncols = P.gx > c->stripe_rem ? (P.gx - c->stripe_rem + c->stripe_mod - 1) / c->stripe_mod : 0;
  P.seed_stride = 8u * (unsigned)P.gx;                            // blockDim.x * gridDim.x, K:1065
  P.max_depth = hf2i(st[9]);
  P.backtex = backtex;
  P.batch = 1;
  P.batch_seed_stride = 0;
  P.coop_steps = c->coop_steps; P.coop_rounds = c->coop_rounds; P.split_parts = c->split_parts;
  P.coop_lanes = c->coop_lanes;
  {
    const int tiles = P.ncols * P.gy;
    P.regions = c->xcd_regions ? MAX_REGIONS : 1;
    if (tiles < 64 * MAX_REGIONS) P.regions = 1;                 // tiny frames: one queue
    // a short launch (few tiles per wave: one frame, or a thin stripe of a few) ends when its slowest band ends; one queue
    // balances better there than eight (1.88 instead of 2.00 ms for a single 1920x1080 frame of the bench scene)
    if (c->short_one_queue && (long long)tiles * batch_hint < (long long)c->coop_tiles_per_wave * c->num_cus * 20) P.regions = 1;
    for (int r = 0; r <= MAX_REGIONS; r++) P.region_start[r] = r <= P.regions ? (int)(((long long)tiles * r + P.regions - 1) / P.regions) : tiles;
  }
  return DR_OK;
}

constexpr int TILE_COUNTERS = 1024;

// the traversal a launch really uses: the wide walk needs its structure (scenes it cannot represent walk the threaded links)
inline int traversal_of(const dr_context* c) { return (c->traversal == DR_TRAVERSAL_WIDE && !c->wide) ? DR_TRAVERSAL_THREADED : c->traversal; }
inline bool uses_persistent(const dr_context* c) { return c->kernel == DR_KERNEL_PERSISTENT && traversal_of(c) != DR_TRAVERSAL_ORDERED; }

template <int OCC>
void launch_tile(dr_context* c, const RenderParams& P) {
  const int tiles = P.ncols * P.gy;
  dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
  const int mode = traversal_of(c);
  if (c->count) {
    if (mode == DR_TRAVERSAL_ORDERED) hipLaunchKernelGGL((render_kernel<true, DR_TRAVERSAL_ORDERED, OCC>), grid, block, 0, c->stream, P);
    else if (mode == DR_TRAVERSAL_WIDE) hipLaunchKernelGGL((render_kernel<true, DR_TRAVERSAL_WIDE, OCC>), grid, block, 0, c->stream, P);
    else hipLaunchKernelGGL((render_kernel<true, DR_TRAVERSAL_THREADED, OCC>), grid, block, 0, c->stream, P);
  } else {
    if (mode == DR_TRAVERSAL_ORDERED) hipLaunchKernelGGL((render_kernel<false, DR_TRAVERSAL_ORDERED, OCC>), grid, block, 0, c->stream, P);
    else if (mode == DR_TRAVERSAL_WIDE) hipLaunchKernelGGL((render_kernel<false, DR_TRAVERSAL_WIDE, OCC>), grid, block, 0, c->stream, P);
    else hipLaunchKernelGGL((render_kernel<false, DR_TRAVERSAL_THREADED, OCC>), grid, block, 0, c->stream, P);
  }
}

// two path records per lane for `waves` waves (render_paired_kernel); false if the memory is not to be had
bool ensure_paths(dr_context* c, size_t waves) {       // `waves` x 128 path records
  if (c->paths && c->paths_waves >= waves) return true;
  if (c->paths) { (void)hipFree(c->paths); c->paths = nullptr; c->paths_waves = 0; }
  if (hipMalloc((void**)&c->paths, waves * 64 * 2 * PATH_UNITS * sizeof(float4)) != hipSuccess) { (void)hipGetLastError(); return false; }
  c->paths_waves = waves;
  return true;
}

// workgroups of trace waves + a shade wave; false if it cannot run (then the caller launches the one-path kernel)
template <int NT, int NS>
bool launch_roles(dr_context* c, const RenderParams& P, unsigned* counter) {
  const int blocks = c->num_cus * 16 / (NT + NS);       // 16 waves per CU
  if (!ensure_paths(c, (size_t)blocks * NS * 512 / 128)) return false;
  if (!c->abort_flag) {
    if (hipMalloc((void**)&c->abort_flag, sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); c->abort_flag = nullptr; return false; }
    (void)hipMemsetAsync(c->abort_flag, 0, sizeof(unsigned), c->stream);
  }
  hipLaunchKernelGGL((render_roles_kernel<NT, NS, 8, 2>), dim3((unsigned)blocks), dim3((NT + NS) * 64), 0, c->stream, P, counter, (const int*)nullptr, (const int*)nullptr,
                     c->paths, c->abort_flag);
  c->roles_used = true;
  return true;
}

template <int OCC, int TRAV_MIN, int PARK_MIN, int P_UNROLL = 1>
void launch_persistent(dr_context* c, const RenderParams& P_in, unsigned* counter, const int* order, unsigned* pixel_cost) {
  RenderParams P = P_in;
  const int* rstart = order ? c->region_start : nullptr;        // identity order: the split travels in P.region_start
  int work = P.ncols * P.gy * P.batch;
  int blocks = c->num_cus * OCC;                   // OCC waves per SIMD on every CU
  if (blocks * 4 > work) blocks = (work + 3) / 4;
  if (P.wave_log && blocks * 4 > WAVE_LOG_WAVES) P.wave_log = nullptr;
  c->wave_log_waves = P.wave_log ? blocks * 4 : 0;
  dim3 grid((unsigned)blocks), block(256);
  if (traversal_of(c) == DR_TRAVERSAL_WIDE) {
    // the cooperative drain shortens a launch's tail; with many tiles per wave the tail does not show and the leaner build is faster
    const bool coop = P.coop_steps > 0 && (long long)work < (long long)c->coop_tiles_per_wave * blocks * 4;
    if (!coop && !DR_WAVE_LOG_DETAIL) c->wave_log_waves = 0;      // only the work-sharing build writes the log
    const bool degenerate = P.max_depth <= 0 || !(P.spp_f > 0.0f);
    if (c->paired && !c->count && !coop && !degenerate && OCC == 5 && ensure_paths(c, (size_t)blocks * 4)) {
      // long launch: two paths per lane
      if (c->pair_thresh <= 32) hipLaunchKernelGGL((render_paired_kernel<5, 32, 8, 2>), grid, block, 0, c->stream, P, counter, order, rstart, pixel_cost, c->paths);
      else if (c->pair_thresh >= 56) hipLaunchKernelGGL((render_paired_kernel<5, 56, 8, 2>), grid, block, 0, c->stream, P, counter, order, rstart, pixel_cost, c->paths);
      else hipLaunchKernelGGL((render_paired_kernel<5, 48, 8, 2>), grid, block, 0, c->stream, P, counter, order, rstart, pixel_cost, c->paths);
      return;
    }
    if (c->count) hipLaunchKernelGGL((render_persistent_kernel<true, OCC, TRAV_MIN, PARK_MIN, P_UNROLL, true, false>), grid, block, 0, c->stream, P, counter, order, rstart, pixel_cost);
    else if (coop) hipLaunchKernelGGL((render_persistent_kernel<false, OCC, TRAV_MIN, PARK_MIN, P_UNROLL, true, true>), grid, block, 0, c->stream, P, counter, order, rstart, pixel_cost);
    else hipLaunchKernelGGL((render_persistent_kernel<false, OCC, TRAV_MIN, PARK_MIN, P_UNROLL, true, false>), grid, block, 0, c->stream, P, counter, order, rstart, pixel_cost);
    return;
  }
  if (c->count) hipLaunchKernelGGL((render_persistent_kernel<true, OCC, TRAV_MIN, PARK_MIN, P_UNROLL, false>), grid, block, 0, c->stream, P, counter, order, rstart, pixel_cost);
  else hipLaunchKernelGGL((render_persistent_kernel<false, OCC, TRAV_MIN, PARK_MIN, P_UNROLL, false>), grid, block, 0, c->stream, P, counter, order, rstart, pixel_cost);
}

// Six waves per SIMD (occupancy 6): only the wide walk's lean build fits -- 80 VGPRs and 26 KiB of LDS per workgroup -- and only with the
// default thresholds; every other launch (counting build, work-sharing build of short launches, other tunings) runs five.
bool launch_wide_lean6(dr_context* c, const RenderParams& P_in, unsigned* counter, const int* order, unsigned* pixel_cost) {
  if (traversal_of(c) != DR_TRAVERSAL_WIDE || c->count || c->paired || c->trav_min != 32 || c->park_min != 16 || c->unroll != 2) return false;
  RenderParams P = P_in;
  const long long work = (long long)P.ncols * P.gy * P.batch;
  if (P.coop_steps > 0 && work < (long long)c->coop_tiles_per_wave * c->num_cus * 5 * 4) return false;      // a short launch: work-sharing build
  int blocks = c->num_cus * 6;
  if ((long long)blocks * 4 > work) blocks = (int)((work + 3) / 4);
  if (!DR_WAVE_LOG_DETAIL || blocks * 4 > WAVE_LOG_WAVES) P.wave_log = nullptr;      // only experiment builds log the lean kernel's waves
  c->wave_log_waves = P.wave_log ? blocks * 4 : 0;
  hipLaunchKernelGGL((render_persistent_kernel<false, 6, 32, 16, 2, true, false>), dim3((unsigned)blocks), dim3(256), 0, c->stream, P, counter, order,
                     order ? c->region_start : nullptr, pixel_cost);
  return true;
}

// The instantiated tunings; dr_context_set_option only accepts these values.
template <int OCC>
void launch_persistent_occ(dr_context* c, const RenderParams& P, unsigned* counter, const int* order, unsigned* pcost) {
  const int key = c->trav_min * 100 + c->park_min + 10000 * (c->unroll - 1);
  switch (key) {
    case 13208: launch_persistent<OCC, 32, 8, 2>(c, P, counter, order, pcost); break;
    case 13216: launch_persistent<OCC, 32, 16, 2>(c, P, counter, order, pcost); break;
    case 23208: launch_persistent<OCC, 32, 8, 3>(c, P, counter, order, pcost); break;
    case 3200: launch_persistent<OCC, 32, 0>(c, P, counter, order, pcost); break;
    case 4800: launch_persistent<OCC, 48, 0>(c, P, counter, order, pcost); break;
    case 4808: launch_persistent<OCC, 48, 8>(c, P, counter, order, pcost); break;
    case 3216: launch_persistent<OCC, 32, 16>(c, P, counter, order, pcost); break;
    default:   launch_persistent<OCC, 32, 8>(c, P, counter, order, pcost); break;
  }
}

// Cost-feedback buffers for `tiles` tiles; returns the order to use for this launch (or null)
// and the per-pixel cost buffer to fill (or null).
void feedback_buffers(dr_context* c, const RenderParams& P, int tiles, const int*& order, unsigned*& pcost) {
  order = nullptr; pcost = nullptr;
  if (!c->feedback) return;
  if (c->order_capacity < tiles) {
    for (void* b : {(void*)c->pixel_cost, (void*)c->tile_cost, (void*)c->tile_order, (void*)c->region_start}) if (b) (void)hipFree(b);
    c->pixel_cost = nullptr; c->tile_cost = nullptr; c->tile_order = nullptr; c->region_start = nullptr; c->order_capacity = 0; c->order_valid = false;
    if (hipMalloc((void**)&c->pixel_cost, (size_t)tiles * 64 * sizeof(unsigned)) == hipSuccess &&
        hipMalloc((void**)&c->tile_cost, (size_t)tiles * sizeof(unsigned)) == hipSuccess &&
        hipMalloc((void**)&c->region_start, (2 * MAX_REGIONS + 1) * sizeof(int)) == hipSuccess &&
        hipMalloc((void**)&c->tile_order, (size_t)tiles * sizeof(int)) == hipSuccess)
      c->order_capacity = tiles;
    else return;
  }
  // the stored order belongs to one view: same settings, size and stripe (progressive frames)
  float key[18] = {0};
  memcpy(key, c->cur_settings, 13 * sizeof(float));
  key[13] = (float)P.W; key[14] = (float)P.H; key[15] = (float)(P.stripe_mod * 1024 + P.stripe_rem) + 0.125f * (float)P.regions;
  key[16] = (float)P.ncols; key[17] = (float)P.gy;      // the tile grid (the preview divisor settings[11] changes it with W and H unchanged)
  if (c->order_valid && memcmp(c->order_key, key, sizeof(key)) == 0) order = c->tile_order;
  else if (c->order_valid && c->order_follows_camera && memcmp(c->order_key + 13, key + 13, 5 * sizeof(float)) == 0) {
    // same frame geometry, other camera / depth / samples (an interactive viewer moving the camera, K:2341-2500: every frame is a new
    // view): the last view's costs are a better guess than none -- any order is a valid order -- and they are refreshed at once
    order = c->tile_order;
    memcpy(c->order_key, key, sizeof(key));
    c->order_age = 0;
  } else { memcpy(c->order_key, key, sizeof(key)); c->order_valid = false; }
  pcost = c->pixel_cost;
}

// enqueue one launch (P.batch frames); no events, no sync
void enqueue_frame(dr_context* c, const RenderParams& P) {
  const int tiles = P.ncols * P.gy;
  if (uses_persistent(c)) {
    if (c->tile_cursor + MAX_REGIONS > TILE_COUNTERS) {
      (void)hipMemsetAsync(c->tile_counters, 0, TILE_COUNTERS * sizeof(unsigned), c->stream);
      c->tile_cursor = 0;
    }
    unsigned* counter = c->tile_counters + c->tile_cursor;      // one counter per region
    c->tile_cursor += MAX_REGIONS;
    if (c->roles && traversal_of(c) == DR_TRAVERSAL_WIDE && !c->count && P.max_depth > 0 && P.spp_f > 0.0f &&
        (long long)tiles * P.batch >= (long long)c->coop_tiles_per_wave * c->num_cus * 20 && (c->roles == 7 ? launch_roles<7, 1>(c, P, counter) : (c->roles == 6 ? launch_roles<6, 2>(c, P, counter) : launch_roles<3, 1>(c, P, counter)))) return;
    const int* order; unsigned* pcost;
    feedback_buffers(c, P, tiles, order, pcost);
    if (c->occupancy >= 6 && launch_wide_lean6(c, P, counter, order, pcost)) {}
    else if (c->occupancy >= 5) launch_persistent_occ<5>(c, P, counter, order, pcost);
    else launch_persistent_occ<4>(c, P, counter, order, pcost);
    // next launch's order from this launch's costs (stream-ordered, no host sync).  The view does not change between the frames of
    // a progressive render, so after the first two launches of a view the order is refreshed every feedback_every-th launch only
    // (the two kernels take 75 us: nothing for a launch of 32 frames, 6 % of a launch of one)
    if (pcost && !order) c->order_age = 0;
    if (pcost && (c->order_age < 2 || c->order_age % c->feedback_every == 0)) {
      hipLaunchKernelGGL(tile_cost_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, c->stream, c->pixel_cost, c->tile_cost, tiles);
      hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, c->stream, c->tile_cost, c->tile_order, c->region_start, tiles, P.regions, c->heavy_factor, c->split_steps,
                         c->split_parts > 1 ? (int)((long long)c->num_cus * (c->occupancy >= 5 ? 5 : 4) * 4 * c->split_waves / (100 * c->split_parts)) : 0);      // at most split_waves % of the waves start with a part of a split tile
      c->order_valid = true;
    }
    c->order_age++;
    return;
  }
  if (c->occupancy >= 6) launch_tile<6>(c, P);
  else launch_tile<4>(c, P);
}

int set_option(dr_context* c, const std::string& name, int v) {
  if (name == "kernel") { if (v != DR_KERNEL_TILE && v != DR_KERNEL_PERSISTENT) goto bad; c->kernel = v; }
  else if (name == "occupancy") { if (v != 4 && v != 5 && v != 6) goto bad; c->occupancy = v; }
  else if (name == "trav_min") { if (v != 32 && v != 48) goto bad; c->trav_min = v; }
  else if (name == "park_min") { if (v != 0 && v != 8 && v != 16) goto bad; c->park_min = v; }
  else if (name == "heavy_factor") { if (v < -1 || v > 1000) goto bad; c->heavy_factor = v; c->order_valid = false; }
  else if (name == "coop_steps") { if (v < 0) goto bad; c->coop_steps = v; }
  else if (name == "coop_lanes") { if (v < 1 || v > 64) goto bad; c->coop_lanes = v; }
  else if (name == "split_parts") { if (v != 1 && v != 2 && v != 4 && v != 8) goto bad; c->split_parts = v; }
  else if (name == "split_waves") { if (v < 1 || v > 1000) goto bad; c->split_waves = v; c->order_valid = false; }
  else if (name == "split_steps") { if (v < 16 || v > 4080) goto bad; c->split_steps = v & ~15; c->order_valid = false; }
  else if (name == "short_one_queue") { c->short_one_queue = v != 0; c->order_valid = false; }
  else if (name == "coop_rounds") { if (v < 1 || v > 16) goto bad; c->coop_rounds = v; }
  else if (name == "wave_log") {
    if (v != 0 && v != 1) goto bad;
    if (v && !c->wave_log) {
      const size_t bytes = (size_t)WAVE_LOG_WAVES * 16 * sizeof(unsigned long long) + PIXEL_LOG_WORDS * sizeof(unsigned);
      if (hipSetDevice(c->device) != hipSuccess || hipMalloc((void**)&c->wave_log, bytes) != hipSuccess) { c->wave_log = nullptr; set_error("cannot allocate the wave log"); return DR_ERR_DEVICE; }
      (void)hipMemsetAsync(c->wave_log, 0, bytes, c->stream);
    }
    c->wave_log_on = v;
  }
  else if (name == "coop_tiles_per_wave") { if (v < 0) goto bad; c->coop_tiles_per_wave = v; }
  else if (name == "paired") { c->paired = v != 0; }
  else if (name == "roles") { if (v != 0 && v != 3 && v != 6 && v != 7) goto bad; c->roles = v; }
  else if (name == "pair_thresh") { if (v != 32 && v != 48 && v != 56) goto bad; c->pair_thresh = v; }
  else if (name == "xcd_regions") { c->xcd_regions = v != 0; c->order_valid = false; }
  else if (name == "unroll") { if (v < 1 || v > 3) goto bad; c->unroll = v; }
  else if (name == "batch_frames") { if (v < 1 || v > 256) goto bad; c->batch_frames = v; }
  else if (name == "feedback") { c->feedback = v != 0; c->order_valid = false; }
  else if (name == "order_follows_camera") { c->order_follows_camera = v != 0; }
  else if (name == "feedback_every") { if (v < 1) goto bad; c->feedback_every = v; }
  else if (name == "wide_tree") { if (v != 0 && v != 1) goto bad; c->wide_tree = v; }      // takes effect at the next dr_context_upload_scene
  else { set_error("unknown option '" + name + "'"); return DR_ERR_INVALID; }
  return DR_OK;
bad:
  set_error("value not supported for option '" + name + "'");
  return DR_ERR_INVALID;
}

int launch_render(dr_context* c, const RenderParams& P) {
  int tiles = P.ncols * P.gy;
  if (tiles <= 0) return DR_OK;
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  enqueue_frame(c, P);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  return DR_OK;
}

int check_abort(dr_context* c) {
  if (!c->roles_used) return DR_OK;
  unsigned f = 0;
  HIP_TRY(hipMemcpyAsync(&f, c->abort_flag, sizeof(f), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->roles_used = false;
  if (f) { (void)hipMemsetAsync(c->abort_flag, 0, sizeof(unsigned), c->stream); set_error("render_roles_kernel gave up waiting (internal protocol failure): the frame is incomplete"); return DR_ERR_DEVICE; }
  return DR_OK;
}

int collect_time(dr_context* c, uint64_t frames, uint64_t samples) {
  HIP_TRY(hipEventSynchronize(c->ev1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->stats.kernel_ms += ms;
  c->stats.frames += frames;
  c->stats.samples += samples;
  return DR_OK;
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t n) { HIP_TRY(hipMalloc((void**)&p, (n ? n : 1) * sizeof(T))); return DR_OK; }
  int put(const T* src, size_t n) { HIP_TRY(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice)); return DR_OK; }
  int get(T* dst, size_t n) { HIP_TRY(hipMemcpy(dst, p, n * sizeof(T), hipMemcpyDeviceToHost)); return DR_OK; }
};

}  // namespace

extern "C" {

int dr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int dr_context_create(int device_ordinal, dr_context** out) {
  if (!out) { set_error("out is null"); return DR_ERR_INVALID; }
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device (this library has no CPU fallback)"); return DR_ERR_DEVICE; }
  if (device_ordinal < 0 || device_ordinal >= n) { set_error("device ordinal out of range"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(device_ordinal));
  dr_context* c = new dr_context();
  c->device = device_ordinal;
  if (const char* env = getenv("DOGERAY_OPTIONS")) {   // "name=value,name=value": tuning experiments without recompiling callers
    std::string e(env);
    size_t pos = 0;
    while (pos < e.size()) {
      size_t comma = e.find(',', pos);
      if (comma == std::string::npos) comma = e.size();
      std::string kv = e.substr(pos, comma - pos);
      size_t eq = kv.find('=');
      if (eq != std::string::npos && set_option(c, kv.substr(0, eq), atoi(kv.c_str() + eq + 1)) != DR_OK) {
        delete c;
        return DR_ERR_INVALID;
      }
      pos = comma + 1;
    }
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
  }
  memset(&c->stats, 0, sizeof(c->stats));
  // the stream first: every memset below is ordered on it, like the kernels that use the buffers
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
      hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->pev0[0]) != hipSuccess || hipEventCreate(&c->pev1[0]) != hipSuccess ||
      hipEventCreate(&c->pev0[1]) != hipSuccess || hipEventCreate(&c->pev1[1]) != hipSuccess) {
    set_error("cannot create stream/events");
    dr_context_destroy(c);
    return DR_ERR_DEVICE;
  }
  if (hipMalloc((void**)&c->tile_counters, TILE_COUNTERS * sizeof(unsigned)) != hipSuccess ||
      hipMemsetAsync(c->tile_counters, 0, TILE_COUNTERS * sizeof(unsigned), c->stream) != hipSuccess ||
      hipMalloc((void**)&c->counters, 16 * sizeof(unsigned long long)) != hipSuccess ||
      hipMemsetAsync(c->counters, 0, 16 * sizeof(unsigned long long), c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess) {
    set_error("cannot allocate tile counters / statistics");
    dr_context_destroy(c);
    return DR_ERR_DEVICE;
  }
  *out = c;
  return DR_OK;
}

void dr_context_destroy(dr_context* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  void* bufs[] = {c->abort_flag, c->wave_log, c->paths, c->packed[0], c->packed[1], c->walk, c->wide, c->pairs, c->prims, c->shade, c->tex, c->texels, c->frame, c->accum, c->present, c->counters, c->tile_counters, c->pixel_cost, c->tile_cost, c->tile_order, c->region_start};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (int k = 0; k < 2; k++) { if (c->pev0[k]) (void)hipEventDestroy(c->pev0[k]); if (c->pev1[k]) (void)hipEventDestroy(c->pev1[k]); }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int dr_context_upload_scene(dr_context* c, const dr_scene* s) {
  if (!c || !s) { set_error("null argument"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  DeviceImage img;
  int rc = DR_OK;
  try {
    rc = linearise(s->host, img, c->wide_tree);
  } catch (const std::exception& e) {
    set_error(std::string("scene could not be linearised: ") + e.what());
    return DR_ERR_NOMEM;
  }
  if (rc != DR_OK) return rc;
  if ((rc = upload(c->walk, img.walk)) != DR_OK) return rc;
  c->walk_bytes = img.walk.size() * sizeof(DevUnit);
  if (c->wide) { (void)hipFree(c->wide); c->wide = nullptr; }
  c->wide_bytes = 0; c->wide_depth = img.wide_depth; c->wide_nodes = img.wide_nodes; c->wide_pmax = img.wide_pmax;
  if (!img.wide.empty()) {
    if ((rc = upload(c->wide, img.wide)) != DR_OK) return rc;
    c->wide_bytes = img.wide.size() * sizeof(DevUnit);
  }
  if ((rc = upload(c->pairs, img.pairs)) != DR_OK) return rc;
  if ((rc = upload(c->prims, img.prims)) != DR_OK) return rc;
  if ((rc = upload(c->shade, img.shade)) != DR_OK) return rc;
  if ((rc = upload(c->tex, img.tex)) != DR_OK) return rc;
  if ((rc = upload(c->texels, img.texels)) != DR_OK) return rc;
  c->n_prims = (int)img.prims.size();
  c->n_tex = (int)img.tex.size();
  c->slot_to_orig = img.slot_to_orig;
  int depth = 0;
  while (((size_t)1 << depth) < img.prims.size()) depth++;
  c->tree_depth = depth;
  return DR_OK;
}

int dr_context_set_stripe(dr_context* c, int mod, int rem) {
  if (!c || mod < 1 || rem < 0 || rem >= mod) { set_error("stripe: need mod >= 1 and 0 <= rem < mod"); return DR_ERR_INVALID; }
  c->stripe_mod = mod; c->stripe_rem = rem;
  return DR_OK;
}

int dr_context_set_option(dr_context* c, const char* name, int value) {
  if (!c || !name) { set_error("null argument"); return DR_ERR_INVALID; }
  return set_option(c, name, value);
}

int dr_context_get_option(const dr_context* c, const char* name, int* value) {
  if (!c || !name || !value) { set_error("null argument"); return DR_ERR_INVALID; }
  const std::string n = name;
  if (n == "kernel") *value = c->kernel;
  else if (n == "batch_frames") *value = c->batch_frames;
  else if (n == "feedback") *value = c->feedback ? 1 : 0;
  else if (n == "feedback_every") *value = c->feedback_every;
  else if (n == "order_follows_camera") *value = c->order_follows_camera;
  else if (n == "occupancy") *value = c->occupancy;
  else if (n == "trav_min") *value = c->trav_min;
  else if (n == "park_min") *value = c->park_min;
  else if (n == "unroll") *value = c->unroll;
  else if (n == "xcd_regions") *value = c->xcd_regions;
  else if (n == "heavy_factor") *value = c->heavy_factor;
  else if (n == "coop_steps") *value = c->coop_steps;
  else if (n == "coop_lanes") *value = c->coop_lanes;
  else if (n == "wave_log") *value = c->wave_log_on;
  else if (n == "coop_rounds") *value = c->coop_rounds;
  else if (n == "short_one_queue") *value = c->short_one_queue;
  else if (n == "split_parts") *value = c->split_parts;
  else if (n == "split_steps") *value = c->split_steps;
  else if (n == "split_waves") *value = c->split_waves;
  else if (n == "coop_tiles_per_wave") *value = c->coop_tiles_per_wave;
  else if (n == "paired") *value = c->paired;
  else if (n == "roles") *value = c->roles;
  else if (n == "pair_thresh") *value = c->pair_thresh;
  else if (n == "tree_depth") *value = c->tree_depth;
  else if (n == "wide_tree") *value = c->wide_tree;
  else if (n == "wide_depth") *value = c->wide ? c->wide_depth : 0;          // 0: the scene has no wide structure
  else if (n == "wide_nodes") *value = c->wide ? c->wide_nodes : 0;
  else if (n == "traversal") *value = traversal_of(c);                        // the traversal launches really use
  else { set_error("unknown option " + n); return DR_ERR_INVALID; }
  return DR_OK;
}

int dr_context_set_traversal(dr_context* c, int mode) {
  if (!c || (mode != DR_TRAVERSAL_THREADED && mode != DR_TRAVERSAL_ORDERED && mode != DR_TRAVERSAL_WIDE)) { set_error("unknown traversal mode"); return DR_ERR_INVALID; }
  if (mode == DR_TRAVERSAL_ORDERED && c->walk && c->tree_depth > ORDERED_STACK) {
    set_error("ordered traversal supports at most 2^24 primitives");
    return DR_ERR_SCENE;
  }
  c->traversal = mode;
  return DR_OK;
}

int dr_render_frame(dr_context* c, const float settings13[13], int W, int H, float background, uint64_t frame_seed,
                    int32_t* out_int3) {
  if (!c || !settings13) { set_error("null argument"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  RenderParams P;
  int rc = make_params(c, settings13, W, H, background, frame_seed, P);
  if (rc != DR_OK) return rc;
  if (c->traversal == DR_TRAVERSAL_ORDERED && c->tree_depth > ORDERED_STACK) { set_error("tree too deep for ordered traversal"); return DR_ERR_SCENE; }
  size_t elems = (size_t)W * H * 3;
  if ((rc = ensure(c->frame, c->frame_elems, elems)) != DR_OK) return rc;
  HIP_TRY(hipMemsetAsync(c->frame, 0, elems * sizeof(int32_t), c->stream));   // unrendered margins are 0
  P.out = c->frame;
  P.accumulate = 0;
  if ((rc = launch_render(c, P)) != DR_OK) return rc;
  c->stats.launches += 1;
  uint64_t samples = (uint64_t)P.ncols * P.gy * 64ull * (uint64_t)(P.spp_f > 0 ? ceilf(P.spp_f) : 0);
  if ((rc = collect_time(c, 1, samples)) != DR_OK) return rc;
  if (out_int3) {
    HIP_TRY(hipMemcpyAsync(out_int3, c->frame, elems * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DR_OK;
}

int dr_accum_reset(dr_context* c, int W, int H) {
  if (!c || W <= 0 || H <= 0) { set_error("bad argument"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  size_t elems = (size_t)W * H * 3;
  int rc = ensure(c->accum, c->accum_elems, elems);
  if (rc != DR_OK) return rc;
  c->accW = W; c->accH = H;
  HIP_TRY(hipMemsetAsync(c->accum, 0, elems * sizeof(int32_t), c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DR_OK;
}

}  // extern "C"

namespace {
// enqueues the launches of `nframes` frames between two event records; no host synchronisation
int accumulate_enqueue(dr_context* c, const float settings13[13], int W, int H, float background, uint64_t frame_seed,
                       uint64_t seed_stride, int nframes, hipEvent_t e0, hipEvent_t e1, uint64_t& samples) {
  samples = 0;
  if (!c || !settings13 || nframes < 0) { set_error("bad argument"); return DR_ERR_INVALID; }
  if (!c->accum || c->accW != W || c->accH != H) { set_error("call dr_accum_reset(W, H) first"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  RenderParams P;
  const int per = (uses_persistent(c) && c->batch_frames > 1) ? c->batch_frames : 1;
  int rc = make_params(c, settings13, W, H, background, frame_seed, P, nframes < per ? nframes : per);
  if (rc != DR_OK) return rc;
  if (c->traversal == DR_TRAVERSAL_ORDERED && c->tree_depth > ORDERED_STACK) { set_error("tree too deep for ordered traversal"); return DR_ERR_SCENE; }
  P.out = c->accum;
  P.accumulate = 1;
  int tiles = P.ncols * P.gy;
  HIP_TRY(hipEventRecord(e0, c->stream));
  // The persistent kernel renders the frames in batches of `batch_frames` per launch (one work
  // queue over all their tiles, atomic accumulation); the per-tile kernel takes one frame per launch.
  const int per_launch = (uses_persistent(c) && c->batch_frames > 1) ? c->batch_frames : 1;
  uint64_t launches = 0;
  for (int k = 0; k < nframes && tiles > 0; k += per_launch) {
    P.seed = frame_seed + (uint64_t)k * seed_stride;
    P.batch = nframes - k < per_launch ? nframes - k : per_launch;
    P.batch_seed_stride = seed_stride;
    P.accumulate = P.batch > 1 ? 2 : 1;
    enqueue_frame(c, P);
    launches++;
  }
  c->stats.launches += launches;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(e1, c->stream));
  samples = (uint64_t)(tiles > 0 ? tiles : 0) * 64ull * (uint64_t)(P.spp_f > 0 ? ceilf(P.spp_f) : 0) * (uint64_t)nframes;
  return DR_OK;
}

// time of an asynchronous batch whose events are still outstanding (waits for that batch, not for later ones)
int collect_pending(dr_context* c, int k) {
  if (!c->pending[k]) return DR_OK;
  HIP_TRY(hipEventSynchronize(c->pev1[k]));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, c->pev0[k], c->pev1[k]));
  c->stats.kernel_ms += ms;
  c->stats.frames += c->pending_frames[k];
  c->stats.samples += c->pending_samples[k];
  c->pending[k] = false;
  return DR_OK;
}
}  // namespace

extern "C" {

int dr_render_accumulate(dr_context* c, const float settings13[13], int W, int H, float background, uint64_t frame_seed,
                         uint64_t seed_stride, int nframes) {
  if (!c) { set_error("bad argument"); return DR_ERR_INVALID; }
  if (nframes == 0) return DR_OK;
  uint64_t samples = 0;
  int rc = accumulate_enqueue(c, settings13, W, H, background, frame_seed, seed_stride, nframes, c->ev0, c->ev1, samples);
  if (rc != DR_OK) return rc;
  if ((rc = collect_time(c, (uint64_t)nframes, samples)) != DR_OK) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return check_abort(c);
}

int dr_render_accumulate_async(dr_context* c, const float settings13[13], int W, int H, float background, uint64_t frame_seed,
                               uint64_t seed_stride, int nframes) {
  if (!c) { set_error("bad argument"); return DR_ERR_INVALID; }
  if (nframes == 0) return DR_OK;
  const int k = c->pending_next;
  int rc = collect_pending(c, k);           // at most two batches in flight: reusing a pair of events waits for the batch before last
  if (rc != DR_OK) return rc;
  uint64_t samples = 0;
  if ((rc = accumulate_enqueue(c, settings13, W, H, background, frame_seed, seed_stride, nframes, c->pev0[k], c->pev1[k], samples)) != DR_OK) return rc;
  c->pending[k] = true; c->pending_frames[k] = (uint64_t)nframes; c->pending_samples[k] = samples;
  c->pending_next = k ^ 1;
  return DR_OK;
}

int dr_context_synchronize(dr_context* c) {
  if (!c) { set_error("null context"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = collect_pending(c, c->pending_next)) != DR_OK) return rc;       // older first
  if ((rc = collect_pending(c, c->pending_next ^ 1)) != DR_OK) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return check_abort(c);
}

int dr_context_stream(dr_context* c, void** hip_stream) {
  if (!c || !hip_stream) { set_error("null argument"); return DR_ERR_INVALID; }
  *hip_stream = (void*)c->stream;
  return DR_OK;
}

int dr_accum_pack_stripe(dr_context* c, int slot, void** dev_ptr, uint64_t* bytes) {
  if (!c || !c->accum || (slot != 0 && slot != 1)) { set_error("pack: no accumulator, or slot not 0/1"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  const int gx = c->accW / 8;
  const int ncols = gx > c->stripe_rem ? (gx - c->stripe_rem + c->stripe_mod - 1) / c->stripe_mod : 0;
  const size_t run = (size_t)8 * (size_t)c->accH * 3;                 // int32 per block column
  // sized for the largest stripe of this partition (rank 0's), so that every rank's buffer can take part in one
  // equal-sized gather
  const size_t need = (size_t)((gx + c->stripe_mod - 1) / c->stripe_mod > 0 ? (gx + c->stripe_mod - 1) / c->stripe_mod : 1) * run;
  int rc = ensure(c->packed[slot], c->packed_elems[slot], need);
  if (rc != DR_OK) return rc;
  if (ncols > 0) {
    const int run4 = (int)(run / 4);
    int bx = (run4 + 255) / 256; if (bx > 64) bx = 64;
    hipLaunchKernelGGL(stripe_copy_kernel, dim3((unsigned)bx, (unsigned)ncols), dim3(256), 0, c->stream, reinterpret_cast<int4*>(c->packed[slot]),
                       reinterpret_cast<const int4*>(c->accum), ncols, run4, 0ll, (long long)run4, (long long)c->stripe_rem * run4, (long long)c->stripe_mod * run4);
    HIP_TRY(hipGetLastError());
  }
  if (dev_ptr) *dev_ptr = c->packed[slot];
  if (bytes) *bytes = (uint64_t)ncols * run * sizeof(int32_t);
  return DR_OK;
}

int dr_accum_unpack_stripes(dr_context* c, const void* packed_dev, uint64_t rank_stride_bytes, int world, int first_rank, void* hip_stream) {
  if (!c || !c->accum || !packed_dev || world < 1 || first_rank < 0 || first_rank > world || (rank_stride_bytes & 15ull)) { set_error("unpack: bad argument"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : c->stream;
  const int gx = c->accW / 8;
  const size_t run = (size_t)8 * (size_t)c->accH * 3;
  const int run4 = (int)(run / 4);
  int bx = (run4 + 255) / 256; if (bx > 64) bx = 64;
  for (int r = first_rank; r < world; r++) {
    const int ncols = gx > r ? (gx - r + world - 1) / world : 0;
    if (ncols == 0) continue;
    if ((uint64_t)ncols * run * sizeof(int32_t) > rank_stride_bytes) { set_error("unpack: a rank's stripe is larger than rank_stride_bytes"); return DR_ERR_INVALID; }
    hipLaunchKernelGGL(stripe_copy_kernel, dim3((unsigned)bx, (unsigned)ncols), dim3(256), 0, stream, reinterpret_cast<int4*>(c->accum),
                       reinterpret_cast<const int4*>(packed_dev), ncols, run4, (long long)r * run4, (long long)world * run4,
                       (long long)((uint64_t)r * rank_stride_bytes / 16), (long long)run4);
  }
  HIP_TRY(hipGetLastError());
  return DR_OK;
}

int dr_accum_read(dr_context* c, int32_t* out_int3) {
  if (!c || !out_int3 || !c->accum) { set_error("no accumulator"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(out_int3, c->accum, (size_t)c->accW * c->accH * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DR_OK;
}

int dr_accum_present(dr_context* c, int divide_by, uint8_t* out_rgb8) {
  if (!c || !out_rgb8 || !c->accum || divide_by == 0) { set_error("bad argument"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  size_t bytes = (size_t)c->accW * c->accH * 3;
  if (c->present_bytes < bytes) {
    if (c->present) (void)hipFree(c->present);
    c->present = nullptr; c->present_bytes = 0;
    HIP_TRY(hipMalloc((void**)&c->present, bytes));
    c->present_bytes = bytes;
  }
  int n = c->accW * c->accH;
  hipLaunchKernelGGL(present_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->accum, c->present, c->accW, c->accH, divide_by);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out_rgb8, c->present, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DR_OK;
}

int dr_accum_device_ptr(dr_context* c, void** dev_ptr, uint64_t* bytes) {
  if (!c || !dev_ptr || !c->accum) { set_error("no accumulator"); return DR_ERR_INVALID; }
  *dev_ptr = c->accum;
  if (bytes) *bytes = (uint64_t)c->accW * c->accH * 3 * sizeof(int32_t);
  return DR_OK;
}

int dr_stats_enable_counters(dr_context* c, int on) {
  if (!c) { set_error("null context"); return DR_ERR_INVALID; }
  c->count = on != 0;
  return DR_OK;
}

int dr_stats_reset(dr_context* c) {
  if (!c) { set_error("null context"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemsetAsync(c->counters, 0, 16 * sizeof(unsigned long long), c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memset(&c->stats, 0, sizeof(c->stats));
  return DR_OK;
}

int dr_stats_get(dr_context* c, dr_stats* out) {
  if (!c || !out) { set_error("null argument"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  unsigned long long h[16];
  HIP_TRY(hipMemcpy(h, c->counters, sizeof(h), hipMemcpyDeviceToHost));
  *out = c->stats;
  out->rays = h[0]; out->node_visits = h[1]; out->prim_tests = h[2]; out->shades = h[3]; out->texels = h[4];
  if (c->count) out->samples = h[5];
  out->trav_slots = h[6]; out->ray_slots = h[7];
  for (int k = 0; k < 8; k++) out->diag[k] = h[8 + k];
  return DR_OK;
}

int dr_stats_wave_log(dr_context* c, unsigned long long* out, int max_waves, int* n_waves) {
  if (!c || !out || !n_waves || max_waves < 0) { set_error("bad argument"); return DR_ERR_INVALID; }
  if (!c->wave_log) { set_error("wave log is off (option wave_log)"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const int n = c->wave_log_waves < max_waves ? c->wave_log_waves : max_waves;
  if (n > 0) HIP_TRY(hipMemcpy(out, c->wave_log, (size_t)n * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  *n_waves = n;
  return DR_OK;
}

int dr_stats_pixel_times(dr_context* c, unsigned* out, size_t capacity, size_t* n) {
  if (!c || !out || !n) { set_error("bad argument"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  size_t m = c->wave_log && c->pixel_cost ? (size_t)2 * c->order_capacity * 64 : 0;
  if (m > PIXEL_LOG_WORDS || m > capacity) m = 0;
  if (m > 0) HIP_TRY(hipMemcpy(out, c->wave_log + (size_t)WAVE_LOG_WAVES * 16, m * sizeof(unsigned), hipMemcpyDeviceToHost));
  *n = m;
  return DR_OK;
}

int dr_stats_pixel_cost(dr_context* c, unsigned* out, size_t capacity, size_t* n) {
  if (!c || !out || !n) { set_error("bad argument"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const size_t have = c->pixel_cost ? (size_t)c->order_capacity * 64 : 0;      // pixel (tile, lane-in-tile) at tile * 64 + lane, tile = block column * gy + block row
  const size_t m = have < capacity ? have : capacity;
  if (m > 0) HIP_TRY(hipMemcpy(out, c->pixel_cost, m * sizeof(unsigned), hipMemcpyDeviceToHost));
  *n = m;
  return DR_OK;
}

int dr_context_probe_gather(dr_context* c, uint32_t hot_records, int iters, double* records_per_s) {
  if (!c || !records_per_s || iters < 1) { set_error("bad argument"); return DR_ERR_INVALID; }
  if (!c->wide) { set_error("no wide walk resident"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  RenderParams P;
  memset(&P, 0, sizeof(P));
  P.wide = c->wide; P.wide_bytes = (uint32_t)c->wide_bytes;
  const unsigned total = (unsigned)(c->wide_bytes / 64);
  const unsigned nrec = (hot_records == 0 || hot_records > total) ? total : hot_records;
  DevBuf<unsigned> out;
  int rc = out.alloc(1);
  if (rc != DR_OK) return rc;
  const int blocks = c->num_cus * 5;
  hipLaunchKernelGGL(gather_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, P, nrec, iters / 8 + 1, out.p);      // warm-up
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  hipLaunchKernelGGL(gather_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, P, nrec, iters, out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *records_per_s = (double)blocks * 256.0 * (double)iters / ((double)ms * 1e-3);
  return DR_OK;
}

// ---- KAT hooks
#define KAT_PRE(n)                                                        \
  if (!c || (n) < 0) { set_error("bad argument"); return DR_ERR_INVALID; } \
  HIP_TRY(hipSetDevice(c->device));                                        \
  if ((n) == 0) return DR_OK;                                              \
  int rc_ = DR_OK;                                                         \
  (void)rc_;
#define KAT_DO(expr) if ((rc_ = (expr)) != DR_OK) return rc_

int dr_kat_rng(dr_context* c, uint64_t seed, int n, double* out) {
  KAT_PRE(n);
  DevBuf<double> d; KAT_DO(d.alloc((size_t)n));
  hipLaunchKernelGGL(kat_rng_kernel, dim3(1), dim3(64), 0, c->stream, seed, n, d.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return d.get(out, (size_t)n);
}

int dr_kat_aabb(dr_context* c, int n, const float* o, const float* d, const float* mn, const float* mx, int32_t* hit, float* dist) {
  KAT_PRE(n);
  DevBuf<float> bo, bd, bmn, bmx, bdist; DevBuf<int32_t> bhit;
  size_t m = (size_t)n * 3;
  KAT_DO(bo.alloc(m)); KAT_DO(bd.alloc(m)); KAT_DO(bmn.alloc(m)); KAT_DO(bmx.alloc(m)); KAT_DO(bdist.alloc((size_t)n)); KAT_DO(bhit.alloc((size_t)n));
  KAT_DO(bo.put(o, m)); KAT_DO(bd.put(d, m)); KAT_DO(bmn.put(mn, m)); KAT_DO(bmx.put(mx, m));
  hipLaunchKernelGGL(kat_aabb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, bo.p, bd.p, bmn.p, bmx.p, bhit.p, bdist.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  KAT_DO(bhit.get(hit, (size_t)n));
  return bdist.get(dist, (size_t)n);
}

int dr_kat_tri(dr_context* c, int n, const float* o, const float* d, const float* v0, const float* v1, const float* v2, float* t) {
  KAT_PRE(n);
  DevBuf<float> bo, bd, b0, b1, b2, bt;
  size_t m = (size_t)n * 3;
  KAT_DO(bo.alloc(m)); KAT_DO(bd.alloc(m)); KAT_DO(b0.alloc(m)); KAT_DO(b1.alloc(m)); KAT_DO(b2.alloc(m)); KAT_DO(bt.alloc((size_t)n));
  KAT_DO(bo.put(o, m)); KAT_DO(bd.put(d, m)); KAT_DO(b0.put(v0, m)); KAT_DO(b1.put(v1, m)); KAT_DO(b2.put(v2, m));
  hipLaunchKernelGGL(kat_tri_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, bo.p, bd.p, b0.p, b1.p, b2.p, bt.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return bt.get(t, (size_t)n);
}

int dr_kat_sphere(dr_context* c, int n, const float* o, const float* d, const float* centre, const float* radius, float* t) {
  KAT_PRE(n);
  DevBuf<float> bo, bd, bc, br, bt;
  size_t m = (size_t)n * 3;
  KAT_DO(bo.alloc(m)); KAT_DO(bd.alloc(m)); KAT_DO(bc.alloc(m)); KAT_DO(br.alloc((size_t)n)); KAT_DO(bt.alloc((size_t)n));
  KAT_DO(bo.put(o, m)); KAT_DO(bd.put(d, m)); KAT_DO(bc.put(centre, m)); KAT_DO(br.put(radius, (size_t)n));
  hipLaunchKernelGGL(kat_sphere_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, bo.p, bd.p, bc.p, br.p, bt.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return bt.get(t, (size_t)n);
}

int dr_kat_optics(dr_context* c, int n, const float* v, const float* nrm, const float* eta, float* refl, float* refr, float* schlick) {
  KAT_PRE(n);
  DevBuf<float> bv, bn, be, b1, b2, b3;
  size_t m = (size_t)n * 3;
  KAT_DO(bv.alloc(m)); KAT_DO(bn.alloc(m)); KAT_DO(be.alloc((size_t)n)); KAT_DO(b1.alloc(m)); KAT_DO(b2.alloc(m)); KAT_DO(b3.alloc((size_t)n));
  KAT_DO(bv.put(v, m)); KAT_DO(bn.put(nrm, m)); KAT_DO(be.put(eta, (size_t)n));
  hipLaunchKernelGGL(kat_optics_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, bv.p, bn.p, be.p, b1.p, b2.p, b3.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  KAT_DO(b1.get(refl, m)); KAT_DO(b2.get(refr, m));
  return b3.get(schlick, (size_t)n);
}

int dr_kat_normal(dr_context* c, int n, const int32_t* object_index, const float* o, const float* d, const float* t, float* normal, float* texco) {
  KAT_PRE(n);
  if (!c->walk || !object_index || !o || !d || !t || !normal || !texco) { set_error("no scene uploaded, or null argument"); return DR_ERR_INVALID; }
  std::vector<int32_t> slot_of((size_t)c->n_prims, -1), slots((size_t)n);
  for (int sidx = 0; sidx < c->n_prims; sidx++) slot_of[(size_t)c->slot_to_orig[(size_t)sidx]] = sidx;
  for (int i = 0; i < n; i++) {
    if (object_index[i] < 0 || object_index[i] >= c->n_prims) { set_error("object index out of range"); return DR_ERR_INVALID; }
    slots[(size_t)i] = slot_of[(size_t)object_index[i]];
  }
  DevBuf<int32_t> bs; DevBuf<float> bo, bd, bt, bn, bc;
  size_t m = (size_t)n * 3;
  KAT_DO(bs.alloc((size_t)n)); KAT_DO(bo.alloc(m)); KAT_DO(bd.alloc(m)); KAT_DO(bt.alloc((size_t)n)); KAT_DO(bn.alloc(m)); KAT_DO(bc.alloc(m));
  KAT_DO(bs.put(slots.data(), (size_t)n)); KAT_DO(bo.put(o, m)); KAT_DO(bd.put(d, m)); KAT_DO(bt.put(t, (size_t)n));
  RenderParams P;
  memset(&P, 0, sizeof(P));
  P.prims = c->prims; P.shade = c->shade;
  hipLaunchKernelGGL(kat_normal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, P, n, bs.p, bo.p, bd.p, bt.p, bn.p, bc.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  KAT_DO(bn.get(normal, m));
  return bc.get(texco, m);
}

int dr_kat_hit(dr_context* c, int n, const float* o, const float* d, float* t, int32_t* idx, int32_t* visits) {
  KAT_PRE(n);
  if (!c->walk) { set_error("no scene uploaded"); return DR_ERR_INVALID; }
  DevBuf<float> bo, bd, bt; DevBuf<int32_t> bs, bv;
  size_t m = (size_t)n * 3;
  KAT_DO(bo.alloc(m)); KAT_DO(bd.alloc(m)); KAT_DO(bt.alloc((size_t)n)); KAT_DO(bs.alloc((size_t)n)); KAT_DO(bv.alloc((size_t)n));
  KAT_DO(bo.put(o, m)); KAT_DO(bd.put(d, m));
  RenderParams P;
  memset(&P, 0, sizeof(P));
  P.walk = c->walk; P.walk_bytes = (uint32_t)c->walk_bytes; P.pairs = c->pairs; P.prims = c->prims;
  P.wide = c->wide; P.wide_bytes = (uint32_t)c->wide_bytes; P.wide_pmax = c->wide_pmax;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (traversal_of(c) == DR_TRAVERSAL_WIDE) hipLaunchKernelGGL((kat_hit_kernel<DR_TRAVERSAL_WIDE>), grid, block, 0, c->stream, P, n, bo.p, bd.p, bt.p, bs.p, bv.p);
  else if (c->traversal == DR_TRAVERSAL_ORDERED) hipLaunchKernelGGL((kat_hit_kernel<DR_TRAVERSAL_ORDERED>), grid, block, 0, c->stream, P, n, bo.p, bd.p, bt.p, bs.p, bv.p);
  else hipLaunchKernelGGL((kat_hit_kernel<DR_TRAVERSAL_THREADED>), grid, block, 0, c->stream, P, n, bo.p, bd.p, bt.p, bs.p, bv.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  KAT_DO(bt.get(t, (size_t)n));
  if (visits) KAT_DO(bv.get(visits, (size_t)n));
  std::vector<int32_t> slots((size_t)n);
  KAT_DO(bs.get(slots.data(), (size_t)n));
  for (int i = 0; i < n; i++) idx[i] = slots[(size_t)i] >= 0 ? c->slot_to_orig[(size_t)slots[(size_t)i]] : 0;   // hit() returns index 0 on a miss (K:507)
  return DR_OK;
}

}  // extern "C"
