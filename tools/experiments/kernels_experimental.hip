// Round 2's two restructurings of the persistent kernel that were measured SLOWER than it (DESIGN.md "tried and dropped"): two paths per
// lane, and workgroups of trace waves + shade waves.  Kept for the record and for A/B runs, but only in -DDOGERAY_EXPERIMENTAL builds
// (tools/exp_variant.sh); the product library carries the stubs at the end of this file.
#include <hip/hip_runtime.h>

#include "device_core.hpp"
#include "kernels.hpp"
#include "../../include/dogeray_amd.h"

namespace dr {

#ifdef DOGERAY_EXPERIMENTAL
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
  WideRay wr = wide_ray_none();
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
    WideRay wr = wide_ray_none();
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

// ------------------------------------------------------------------ launchers
static_assert(PATH_UNITS == EXPERIMENTAL_PATH_UNITS, "kernels.hpp sizes the path records");
bool experimental_built() { return true; }

bool launch_paired_kernel(hipStream_t stream, const RenderParams& P, int blocks, int pair_thresh, unsigned* counter, const int* order,
                          const int* rstart, unsigned* pixel_cost, void* paths_v) {
  float4* paths = reinterpret_cast<float4*>(paths_v);
  dim3 grid((unsigned)blocks), block(256);
  if (pair_thresh <= 32) hipLaunchKernelGGL((render_paired_kernel<5, 32, 8, 2>), grid, block, 0, stream, P, counter, order, rstart, pixel_cost, paths);
  else if (pair_thresh >= 56) hipLaunchKernelGGL((render_paired_kernel<5, 56, 8, 2>), grid, block, 0, stream, P, counter, order, rstart, pixel_cost, paths);
  else hipLaunchKernelGGL((render_paired_kernel<5, 48, 8, 2>), grid, block, 0, stream, P, counter, order, rstart, pixel_cost, paths);
  return true;
}

namespace {
template <int NT, int NS>
void launch_roles(hipStream_t stream, const RenderParams& P, int blocks, unsigned* counter, float4* paths, unsigned* abort_flag) {
  hipLaunchKernelGGL((render_roles_kernel<NT, NS, 8, 2>), dim3((unsigned)blocks), dim3((NT + NS) * 64), 0, stream, P, counter, (const int*)nullptr, (const int*)nullptr,
                     paths, abort_flag);
}
inline int roles_nt(int roles) { return roles == 7 ? 7 : (roles == 6 ? 6 : 3); }
inline int roles_ns(int roles) { return roles == 6 ? 2 : 1; }
}  // namespace

int roles_blocks(int roles, int num_cus) { return num_cus * 16 / (roles_nt(roles) + roles_ns(roles)); }       // 16 waves per CU
size_t roles_path_waves(int roles, int blocks) { return (size_t)blocks * roles_ns(roles) * 512 / 128; }
bool launch_roles_kernel(hipStream_t stream, const RenderParams& P, int roles, int blocks, unsigned* counter, void* paths_v, unsigned* abort_flag) {
  float4* paths = reinterpret_cast<float4*>(paths_v);
  if (roles == 7) launch_roles<7, 1>(stream, P, blocks, counter, paths, abort_flag);
  else if (roles == 6) launch_roles<6, 2>(stream, P, blocks, counter, paths, abort_flag);
  else launch_roles<3, 1>(stream, P, blocks, counter, paths, abort_flag);
  return true;
}

#else   // product build: the two kernels are not compiled in

bool experimental_built() { return false; }
bool launch_paired_kernel(hipStream_t, const RenderParams&, int, int, unsigned*, const int*, const int*, unsigned*, void*) { return false; }
bool launch_roles_kernel(hipStream_t, const RenderParams&, int, int, unsigned*, void*, unsigned*) { return false; }
int roles_blocks(int, int) { return 0; }
size_t roles_path_waves(int, int) { return 0; }

#endif
}  // namespace dr
