// The render kernels: the per-tile kernel (the reference's launch shape, kernel.cu K:2634-2640) and the persistent kernel (a wave is a
// pool of 64 path slots that refill from tile queues).  Launchers at the end; context.cpp picks options, this file picks the instantiation.
#include <hip/hip_runtime.h>

#include "device_core.hpp"
#include "kernels.hpp"
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
      auto closest = [&](V3 o, V3 d, Ctr& cc) { return closest_hit_wide<COUNT>(wide, P.wide_pmax, P.wide_mu.e, P.wide_mu.l, P.wide_mu.v, o, d, cc, stack); };
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

// PERFRAME: every frame of the launch's batch is stored into a buffer of its own (P.out + frame * P.out_frame_stride) instead of being added into one: the
// grouped present pipeline (dr_pipeline_*: the frames of a group share one launch and are added and shown one by one afterwards).
template <bool COUNT, int OCC, int TRAV_MIN, int PARK_MIN, int P_UNROLL, bool WIDE, bool COOP = true, bool PERFRAME = false>
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
  if (WIDE) {                      // the phase-only state lives in the stash (see the phase): no pixel yet
    int* const st0 = my_lds + WIDE_STACK * 64 + lane;
    for (int k = 1; k < 8; k++) st0[k * 64] = 0;
  }
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
  WideRay wr = wide_ray_none();    // WIDE: clamped 1/direction and margins of the folded node test, a function of the lane's ray
  SignMask sg = sign_mask(inv);    // ... and the signs of 1/direction as select masks
  Xorwow rng; rng.v0 = rng.v1 = rng.v2 = rng.v3 = rng.v4 = rng.d = 0;
  int px = -1, py = 0, sample = 0, bounce = 0;
  int frame = 0;                   // frame of the batch the pixel in this slot belongs to
  int pcode = 0;                   // tile * 64 + lane-in-tile of the pixel in this slot
  unsigned steps = 0;              // node steps spent on the pixel in this slot (the cost fed back)
  unsigned rstart = 0;             // value of `steps` when the current ray started
  const bool degenerate = P.max_depth <= 0 || !(P.spp_f > 0.0f);

  // diagnostic stamps (counting build only): wave lifetime, cycles inside the shade/refill phase
  unsigned long long t_begin = 0, t_phase = 0, n_iter = 0, n_phase = 0, n_nodestep = 0, n_leafstep = 0, n_shaded = 0;
  float* const ray_log = COUNT ? reinterpret_cast<float*>(P.counters[41]) : nullptr;      // measurement aid (dr_context_probe_trace): statistics words 40-42 = rays logged, the log, its room
  const unsigned long long ray_log_room = COUNT ? P.counters[42] : 0ull;
  unsigned long long pb_cands = 0, pb_turns = 0, pb_sphere = 0, pb_disk = 0, pb_retired = 0;      // ... the phase's budget (dr_stats_phase_counts)
  unsigned long long r_begin = 0;
  // wave lifetime in shader cycles and in 100 MHz ticks, every build: two clock reads per wave, written to the statistics buffer only
  t_begin = __builtin_readcyclecounter(); r_begin = __builtin_amdgcn_s_memrealtime();
  bool held = false;               // COOP: this wave has pixels of a split tile and fetches no new tiles while they live
  // wave log (option wave_log): when this wave first found the queue empty, loop iterations since.  Only in the build that short launches
  // use (their timeline is what the log is for): the two scalar instructions per iteration cost the long launches 0.5 %
  constexpr bool WAVE_LOG = WIDE && COOP;
  unsigned long long r_empty = 0, n_after = 0;
  ParkedLeaf pk; pk.v0x = 0; pk.C = pk.D = u32x4{0, 0, 0, 0}; pk.info = 0; pk.parked = false;   // PARK_MIN > 0 only
  for (;;) {
    DR_MARK("loop_top");
    const unsigned long long walking = __ballot(tr.node >= 0 || (PARK_MIN > 0 && pk.parked));
    if (COUNT) n_iter++;
    if (WAVE_LOG && r_empty != 0ull) n_after++;
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
      DR_MARK("phase_begin");
      if (COUNT) { t0 = __builtin_readcyclecounter(); n_phase++; }
      const bool shade_me = tr.node == -1 && !(PARK_MIN > 0 && pk.parked) && !waits_for_helpers;
      if (COUNT) n_shaded += __popcll(__ballot(shade_me));
      bool fresh_ray = false;                  // this lane starts a new ray in this phase: 1/direction is recomputed after the phase
      constexpr bool BOUNCE_HOME = WIDE && !COOP;      // lean build: the bounce count lives in stash word 9 (one register more for the walk)
      float* const st = reinterpret_cast<float*>(my_lds) + (WIDE ? WIDE_STACK * 64 : 0) + lane;      // slot k of this lane: st[k * 64]
      if (WIDE) {
        // ten words (26 KiB of LDS per workgroup with the stack: six workgroups per CU).  Words 2-7 (colour, x + 1 (0: no pixel) with y -- make_params
        // bounds the frame size --, pixel code, sample) and the frame of the batch (upper half of word 1) are needed in this phase only: they LIVE
        // here and are registers only between the read after shading and the write at the end of the phase (the walk loop has eight registers
        // more for the node step).  What goes in now is the walk's state: stack top, stack pointer, step counts.
        st[0 * 64] = __uint_as_float(ws.top);
        reinterpret_cast<unsigned char*>(st + 1 * 64)[0] = (unsigned char)ws.sp;
        st[8 * 64] = __uint_as_float(steps);
        if (COOP) st[9 * 64] = __uint_as_float(rstart);      // (lean build: the ray's first step is of no use to it, and word 9 is where `bounce` lives)
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
      DR_MARK("phase_shade");
      bool ended = false;
      V3 radiance = mk(0, 0, 0);
      // (the hit is shaded in two parts around the phase's ONE rejection loop, which serves the lanes that scatter -- a point in the unit
      // sphere -- and, further down, the lanes that start a path -- a point in the unit disk: device_core.hpp rand_points_merged.  A path that ends
      // here, at an emissive surface or at the depth limit, draws nothing more: its generator is re-seeded with its next pixel.)
      ShadeCtx sc; sc.hitpoint = sc.N = sc.ocolor = mk(0, 0, 0); sc.add_x = sc.rough = sc.ir = sc.r5 = 0.0f; sc.mat = -1; sc.front = false;
      bool scatter_me = false;
      if (shade_me) {
        if (tr.best_slot >= 0 && tr.best_t > 0.0f) {
          if (!shade_prepare<COUNT>(P, path, tr.best_t, tr.best_slot, rng, c, sc, radiance)) ended = true;
          else {
            if (BOUNCE_HOME) bounce = __float_as_int(st[9 * 64]);
            bounce++;
            if (bounce >= P.max_depth) ended = true;        // depth exhausted: black (K:981)
            else scatter_me = true;
            if (BOUNCE_HOME) st[9 * 64] = __int_as_float(bounce);
          }
        } else {
          radiance = shade_miss<COUNT>(P, path, c);
          ended = true;
        }
      }
      DR_MARK("phase_unstash");
      if (WIDE) {
        asm volatile("" ::: "memory");
        color = mk(st[2 * 64], st[3 * 64], st[4 * 64]);
        { const int xy = __float_as_int(st[5 * 64]); px = (int)((unsigned)xy >> 16) - 1; py = xy & 0xffff; }
        pcode = __float_as_int(st[6 * 64]);
        frame = (int)((unsigned)__float_as_int(st[1 * 64]) >> 16); sample = __float_as_int(st[7 * 64]);
        steps = __float_as_uint(st[8 * 64]);
        if (COOP) rstart = __float_as_uint(st[9 * 64]);
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
            store_pixel(P, px, py, color, PERFRAME ? (size_t)frame * (size_t)P.out_frame_stride : (size_t)0);
            if (pixel_cost) pixel_cost[pcode & 0x7fffffff] = steps;
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
              break;
            }
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
        }
        const int n = __popcll(need);
        cur_next += n < avail ? n : avail;
        need = __ballot(want_pixel);
      }
      // ---- start the next path of every lane that has a pixel and no path
      DR_MARK("phase_camera");
      bool new_path = false;
      float cam_nu = 0.0f, cam_nv = 0.0f;
      if (tr.node == -2 && px >= 0) {
        if (degenerate) {
          sample = 0x7fffffff;                     // nothing to trace: the pixel is stored as 0 next round
          if (COUNT) c.samples++;
        } else {
          rng.init(sample_seed(P, px, py, sample, frame));
          if (COUNT) { c.samples++; c.rays++; }
          camera_prepare(P, px, py, rng, cam_nu, cam_nv);
          new_path = true;
        }
      }
      {
        // ---- the phase's rejection loop, once for the wave: sphere points for the lanes that scatter, disk points for the lanes that start a path
        unsigned my_turns = 0;
        const int draw_kind = scatter_me && shade_needs_sphere(sc) ? 3 : (new_path ? 2 : 0);
        const V3 pt = rand_points_merged<COUNT>(rng, draw_kind, &my_turns);
        if (COUNT) {
          // the phase's budget (dr_stats_phase_counts): how many turns the wave's rejection loop ran (= its unluckiest lane's), how many lane-turns were useful
          unsigned wave_turns = my_turns;
          for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)wave_turns, off, 64); wave_turns = o > wave_turns ? o : wave_turns; }
          unsigned long long lane_turns = my_turns;
          for (int off = 32; off > 0; off >>= 1) lane_turns += __shfl_xor(lane_turns, off, 64);
          const unsigned n_sphere = (unsigned)__popcll(__ballot(draw_kind == 3)), n_disk = (unsigned)__popcll(__ballot(draw_kind == 2));
          const unsigned n_retired = (unsigned)__popcll(__ballot(tr.node == -3));
          if (lane == 0) atomicAdd(&P.counters[16 + (wave_turns < 15u ? wave_turns : 15u)], 1ull);      // histogram of turns per phase (the sums below leave with the wave)
          pb_cands += lane_turns; pb_turns += wave_turns; pb_sphere += n_sphere; pb_disk += n_disk; pb_retired += n_retired;
        }
        if (scatter_me) shade_scatter(path, sc, pt, rng);
        if (new_path) {
          camera_finish(P, cam_nu, cam_nv, pt, path.rayo, path.raydir);
          path.atten = splat(1.0f);
          bounce = 0;
          if (BOUNCE_HOME) st[9 * 64] = __int_as_float(0);
          trav_begin(tr);
          rstart = steps;
          fresh_ray = true;
        }
      }
      if (COUNT && ray_log && fresh_ray && frame == 0) {      // measurement aid (dr_context_probe_trace): the rays the launch's first frame traces, in the
        // order a per-bounce wavefront would hold them: bounce by bounce, pixels in tile order.  Statistics words 40-42 = rays logged, the log, its room (entries)
        atomicAdd(&P.counters[40], 1ull);
        const unsigned long long k = (unsigned long long)bounce * (unsigned long long)(ntiles * 64) + (unsigned long long)(pcode & 0x7fffffff);
        if (k < ray_log_room) {
          float* const r = ray_log + k * 8;
          r[0] = path.rayo.x; r[1] = path.rayo.y; r[2] = path.rayo.z; r[3] = 0.0f; r[4] = path.raydir.x; r[5] = path.raydir.y; r[6] = path.raydir.z; r[7] = 0.0f;
        }
      }
      DR_MARK("phase_restore");
      if (WIDE) {
        asm volatile("" ::: "memory");
        ws.top = __float_as_uint(st[0 * 64]); ws.sp = __float_as_int(st[1 * 64]) & 0xff;
        // the phase-only state goes home (the values in registers are dead from here to the next phase)
        st[2 * 64] = color.x; st[3 * 64] = color.y; st[4 * 64] = color.z;
        st[5 * 64] = __int_as_float(((px + 1) << 16) | py); st[6 * 64] = __int_as_float(pcode);
        st[7 * 64] = __int_as_float(sample);
        reinterpret_cast<unsigned short*>(st + 1 * 64)[1] = (unsigned short)frame;
        asm volatile("" ::: "memory");
        color = mk(0, 0, 0); px = -1; py = 0; pcode = 0; sample = 0; frame = 0;
        if (fresh_ray) { ws.top = 0u; ws.sp = 0; ws.sb = 0; }
        inv = mk(1.0f / path.raydir.x, 1.0f / path.raydir.y, 1.0f / path.raydir.z);      // 1/direction and the folded test's margins are
        wr = wide_ray(path.rayo, path.raydir, inv, P.wide_pmax, P.wide_mu.e, P.wide_mu.l, P.wide_mu.v);                // recomputed for every lane rather than stashed
        sg = sign_mask(inv);
      } else {
        asm volatile("" ::: "memory");
        pk.v0x = st[0 * 64]; pk.C = u32x4{__float_as_uint(st[1 * 64]), __float_as_uint(st[2 * 64]), __float_as_uint(st[3 * 64]), __float_as_uint(st[4 * 64])};
        pk.D = u32x4{__float_as_uint(st[5 * 64]), __float_as_uint(st[6 * 64]), __float_as_uint(st[7 * 64]), __float_as_uint(st[8 * 64])};
        pk.info = __float_as_int(st[9 * 64]); pk.parked = st[10 * 64] != 0.0f;
        inv = mk(st[11 * 64], st[12 * 64], st[13 * 64]);
        if (fresh_ray) inv = mk(1.0f / path.raydir.x, 1.0f / path.raydir.y, 1.0f / path.raydir.z);
      }
      if (COUNT) t_phase += __builtin_readcyclecounter() - t0;
      DR_MARK("phase_end");
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
      // (with the phase-only state at home in stash words 1-7, the exchange has words 8-9 of the stash -- the step counts, in registers
      // between phases -- to itself: 128 words = 16 hand-overs of 8 words per round)
      constexpr int XCH_MAX = 16;
      const int n_most = n_idle < n_give ? n_idle : n_give;
      const int n = n_most < XCH_MAX ? n_most : XCH_MAX;
      if (n == 0) break;
      {
        int* const xch = my_lds + (WIDE_STACK + 8) * 64;      // field f of hand-over e at xch[f * XCH_MAX + e]
        const int rank_g = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(givers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)givers, 0u));
        const int rank_i = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
        if (can_give && rank_g < n) {
          const int root = share >= 0 && share < 64 ? share : lane;               // the lane whose pixel this ray belongs to
          if (share < 0) { share_key[lane] = hit_key(tr.best_t, tr.best_slot); share_pend[lane] = 0u; share = 64; }
          atomicAdd(&share_pend[root], 1u);
          unsigned word;
          if (ws.sp > ws.sb) { word = (unsigned)my_stack[ws.sb * 64]; ws.sb++; }
          else { word = ws.top; ws.top = 0u; }
          xch[rank_g + 0 * XCH_MAX] = __float_as_int(path.rayo.x); xch[rank_g + 1 * XCH_MAX] = __float_as_int(path.rayo.y); xch[rank_g + 2 * XCH_MAX] = __float_as_int(path.rayo.z);
          xch[rank_g + 3 * XCH_MAX] = __float_as_int(path.raydir.x); xch[rank_g + 4 * XCH_MAX] = __float_as_int(path.raydir.y); xch[rank_g + 5 * XCH_MAX] = __float_as_int(path.raydir.z);
          xch[rank_g + 6 * XCH_MAX] = (int)word; xch[rank_g + 7 * XCH_MAX] = root;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tr.node == -3 && rank_i < n) {
          path.rayo = mk(__int_as_float(xch[rank_i + 0 * XCH_MAX]), __int_as_float(xch[rank_i + 1 * XCH_MAX]), __int_as_float(xch[rank_i + 2 * XCH_MAX]));
          path.raydir = mk(__int_as_float(xch[rank_i + 3 * XCH_MAX]), __int_as_float(xch[rank_i + 4 * XCH_MAX]), __int_as_float(xch[rank_i + 5 * XCH_MAX]));
          ws.top = (unsigned)xch[rank_i + 6 * XCH_MAX]; ws.sp = 0; ws.sb = 0;
          share = xch[rank_i + 7 * XCH_MAX];
          inv = mk(1.0f / path.raydir.x, 1.0f / path.raydir.y, 1.0f / path.raydir.z);
          wr = wide_ray(path.rayo, path.raydir, inv, P.wide_pmax, P.wide_mu.e, P.wide_mu.l, P.wide_mu.v);
          sg = sign_mask(inv);
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
      // them stand at one, or nobody can take a node step.
      // A leaf step and a node step taken together share their fetch round trip, but each then runs for about half the wave (the two are
      // different code).  The kernel is bound by instruction issue, not by latency (seven waves per SIMD are no faster than six,
      // profiles/r3_h), so outside the drain the node lanes of the lean build sit out a leaf step and the node steps in between run fuller:
      // 0.626 -> 0.598 ms/frame with park_min 20 (profiles/r3_i_exclusive_steps.txt).  The work-sharing build keeps the merged steps (one
      // round trip for both kinds of lanes): what exclusive steps gain in its bulk they lose in its tail, 1.10 against 1.075 ms for a single frame.
      // (Leaf postponing, "leaves as soon as they outnumber the nodes" and other thresholds: measured and dropped, profiles/r3_t_*, r3_p_*.)
      // (the first step is written out and the further ones loop: one loop over all of them compiles to a 3.5 % slower lean kernel -- the same
      // work, another schedule; profiles/r4_a_cleanup_ab.txt)
      const bool at_leaf = tr.node >= 0 && (tr.node & 1);
      const unsigned long long leaves = __ballot(at_leaf);
      const unsigned long long nodes = __ballot(tr.node >= 0 && !(tr.node & 1));
      constexpr int park_thr = PARK_MIN > 0 ? PARK_MIN : 1;
      const bool draining = cur_tile >= ntiles || held;
      const bool do_leaves = leaves != 0ull && ((int)__popcll(leaves) >= park_thr || nodes == 0ull || draining);
      constexpr bool EXCLUSIVE = !COOP;
      const bool do_nodes = !EXCLUSIVE || !do_leaves || draining;
      if (COUNT) { n_leafstep += do_leaves; n_nodestep += nodes != 0ull && do_nodes; }
      if (tr.node >= 0 && (at_leaf ? do_leaves : do_nodes)) {
        if (COUNT) { if (first_active_lane()) c.trav_slots += 64; }
        const WideRec r = wide_fetch(walk, tr.node);
        if (at_leaf) wide_leaf_compute<COUNT>(r, path.rayo, path.raydir, inv, sg, tr, ws, my_stack, c);
        else wide_node_compute<COUNT>(r, wr, sg, tr, ws, my_stack, c);
        steps++;
      }
      for (int u = 1; u < P_UNROLL; u++) {         // the further steps of an iteration decide anew
        const bool at_leaf2 = tr.node >= 0 && (tr.node & 1);
        const unsigned long long leaves2 = __ballot(at_leaf2);
        const unsigned long long nodes2 = __ballot(tr.node >= 0 && !(tr.node & 1));
        const bool do_leaves2 = leaves2 != 0ull && ((int)__popcll(leaves2) >= park_thr || nodes2 == 0ull || draining);
        const bool do_nodes2 = !EXCLUSIVE || !do_leaves2 || draining;
        if (COUNT) { n_leafstep += do_leaves2; n_nodestep += nodes2 != 0ull && do_nodes2; }
        if (tr.node >= 0 && (at_leaf2 ? do_leaves2 : do_nodes2)) {
          if (COUNT) { if (first_active_lane()) c.trav_slots += 64; }
          const WideRec r = wide_fetch(walk, tr.node);
          if (at_leaf2) wide_leaf_compute<COUNT>(r, path.rayo, path.raydir, inv, sg, tr, ws, my_stack, c);
          else wide_node_compute<COUNT>(r, wr, sg, tr, ws, my_stack, c);
          steps++;
        }
      }
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
      w[0] = r_begin; w[1] = r_empty; w[2] = r_end; w[3] = n_after;
    }
  }
  if (COUNT && lane == 0) {
    atomicAdd(&P.counters[9], t_phase);
    atomicAdd(&P.counters[10], n_iter);
    atomicAdd(&P.counters[11], n_phase);
    atomicAdd(&P.counters[12], n_nodestep);
    atomicAdd(&P.counters[13], n_leafstep);
    atomicAdd(&P.counters[14], n_shaded);
    atomicAdd(&P.counters[32], pb_cands);        // candidates drawn by all lanes
    atomicAdd(&P.counters[33], pb_turns);        // turns the waves ran
    atomicAdd(&P.counters[34], pb_sphere);       // lanes that drew a point in the sphere (scatter)
    atomicAdd(&P.counters[35], pb_disk);         // lanes that drew a point in the disk (new path)
    atomicAdd(&P.counters[36], pb_retired);      // retired lanes summed over phases
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

// (COOP_PARK, COOP_UNROLL: the work-sharing build's own leaf-step threshold and steps per iteration -- short launches like fewer lanes per leaf step and more
// steps between two looks at the queue: 32 / 12 / 4 against the lean build's 32 / 20 / 2, one frame per launch 0.915 against 0.945 ms, profiles/r4_q_short_launch_schedule.txt)
template <int OCC, int TRAV_MIN, int PARK_MIN, int P_UNROLL = 1, int COOP_PARK = PARK_MIN, int COOP_UNROLL = P_UNROLL>
int launch_persistent(hipStream_t stream, const RenderParams& P_in, const PersistentCfg& cfg, unsigned* counter, const int* order, const int* rstart, unsigned* pixel_cost) {
  RenderParams P = P_in;
  int work = P.ncols * P.gy * P.batch;
  int blocks = cfg.num_cus * OCC;                   // OCC waves per SIMD on every CU
  if (blocks * 4 > work) blocks = (work + 3) / 4;
  if (P.wave_log && blocks * 4 > WAVE_LOG_WAVES) P.wave_log = nullptr;
  int log_waves = P.wave_log ? blocks * 4 : 0;
  dim3 grid((unsigned)blocks), block(256);
  if (cfg.traversal == DR_TRAVERSAL_WIDE) {
    // the cooperative drain shortens a launch's tail; with many tiles per wave the tail does not show and the leaner build is faster
    const bool coop = P.coop_steps > 0 && (long long)work < (long long)cfg.coop_tiles_per_wave * blocks * 4;
    if (!coop) log_waves = 0;      // only the work-sharing build writes the log
    if (cfg.count) hipLaunchKernelGGL((render_persistent_kernel<true, OCC, TRAV_MIN, PARK_MIN, P_UNROLL, true, false>), grid, block, 0, stream, P, counter, order, rstart, pixel_cost);
    else if (coop) hipLaunchKernelGGL((render_persistent_kernel<false, OCC, TRAV_MIN, COOP_PARK, COOP_UNROLL, true, true>), grid, block, 0, stream, P, counter, order, rstart, pixel_cost);
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
  if (P.out_frame_stride) hipLaunchKernelGGL((render_persistent_kernel<false, 6, 32, 20, 2, true, false, true>), dim3((unsigned)blocks), dim3(256), 0, stream, P, counter, order, rstart, pixel_cost);
  else hipLaunchKernelGGL((render_persistent_kernel<false, 6, 32, 20, 2, true, false>), dim3((unsigned)blocks), dim3(256), 0, stream, P, counter, order, rstart, pixel_cost);
  return true;
}

// The instantiated schedules (option "schedule"): 0 = the tuned one -- shade / refill below 32 walking lanes, leaf steps for 20 lanes, two steps per
// loop iteration (work-sharing build: 12 lanes, four steps) --; 1 and 2 keep the other paths of the loop alive in the tests (leaf steps for 8 lanes, one step per iteration; shade / refill
// below 48 lanes, leaves tested on the spot).  Every other combination rounds 2 and 3 measured is in profiles/r2_*, r3_p_*.
template <int OCC>
int launch_persistent_occ(hipStream_t stream, const RenderParams& P, const PersistentCfg& cfg, unsigned* counter, const int* order, const int* rstart, unsigned* pcost) {
  switch (cfg.schedule) {
    case 0:  return launch_persistent<OCC, 32, 20, 2, 12, 4>(stream, P, cfg, counter, order, rstart, pcost);
    case 1:  return launch_persistent<OCC, 32, 8, 1>(stream, P, cfg, counter, order, rstart, pcost);
    default: return launch_persistent<OCC, 48, 0, 1>(stream, P, cfg, counter, order, rstart, pcost);
  }
}

}  // namespace

bool persistent_kernel_can_store_per_frame(const PersistentCfg& cfg) {
  return cfg.traversal == DR_TRAVERSAL_WIDE && !cfg.count && cfg.schedule == 0 && cfg.occupancy >= 6;
}

int launch_persistent_kernel(hipStream_t stream, const RenderParams& P, const PersistentCfg& cfg, unsigned* tile_counter, const int* order,
                             const int* region_start, unsigned* pixel_cost) {
  const int* rstart = order ? region_start : nullptr;        // identity order: the split travels in P.region_start
  int log_waves = 0;
  // (a caller that sets out_frame_stride has asked persistent_kernel_can_store_per_frame for this very configuration: context.cpp pipeline_flush)
  if (P.out_frame_stride && persistent_kernel_can_store_per_frame(cfg)) {
    if (launch_wide_lean6(stream, P, cfg, tile_counter, order, rstart, pixel_cost, log_waves)) return log_waves;
    // a short group: the work-sharing build, five waves per SIMD
    const int work = P.ncols * P.gy * P.batch;
    int blocks = cfg.num_cus * 5;
    if (blocks * 4 > work) blocks = (work + 3) / 4;
    RenderParams Q = P;
    if (Q.wave_log && blocks * 4 > WAVE_LOG_WAVES) Q.wave_log = nullptr;
    hipLaunchKernelGGL((render_persistent_kernel<false, 5, 32, 12, 4, true, true, true>), dim3((unsigned)blocks), dim3(256), 0, stream, Q, tile_counter, order, rstart, pixel_cost);
    return Q.wave_log ? blocks * 4 : 0;
  }
  if (cfg.occupancy >= 6 && launch_wide_lean6(stream, P, cfg, tile_counter, order, rstart, pixel_cost, log_waves)) return log_waves;
  if (cfg.occupancy >= 5) return launch_persistent_occ<5>(stream, P, cfg, tile_counter, order, rstart, pixel_cost);
  return launch_persistent_occ<4>(stream, P, cfg, tile_counter, order, rstart, pixel_cost);
}

}  // namespace dr
