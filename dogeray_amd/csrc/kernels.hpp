// Launchers of the device code, one translation unit per family of kernels:
//   kernels_render.hip        the per-tile kernel (reference launch shape) and the persistent kernel (waves as pools of 64 path slots)
//   kernels_aux.hip           tile-order feedback, present divide, stripe copies of the multi-GPU gather, gather probe, known-answer kernels
// (the measured-slower kernels of rounds 2 and 3 -- two paths per lane, waves with roles, the pool kernel -- are archived under tools/experiments/)
// context.cpp (host only: resident scene, options, the C ABI) calls these and never sees a kernel.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "device_layout.h"

namespace dr {

constexpr int MAX_REGIONS = 8;           // tile queues of the persistent kernels (one per XCD)
constexpr int WAVE_LOG_WAVES = 16384;    // waves the wave log (option wave_log) has room for

// What a launch of the persistent kernel needs to know of the context's options (dr_context_set_option)
struct PersistentCfg {
  int traversal;             // the traversal the launch really uses (DR_TRAVERSAL_WIDE or DR_TRAVERSAL_THREADED)
  int occupancy;             // 4, 5 or 6 waves per SIMD
  int schedule;              // 0, 1, 2: option "schedule" (kernels_render.hip launch_persistent_occ)
  int num_cus;
  int coop_tiles_per_wave;
  bool count;                // counting build
};

// kernels_render.hip
void launch_tile_kernel(hipStream_t stream, const RenderParams& P, int traversal, bool count, int occupancy);
// returns the number of waves that write the wave log (0: the launched build does not log)
int launch_persistent_kernel(hipStream_t stream, const RenderParams& P, const PersistentCfg& cfg, unsigned* tile_counter, const int* order,
                             const int* region_start, unsigned* pixel_cost);

constexpr int COUNTER_WORDS = 48;   // 64-bit words of a context's statistics buffer: [0, 8) ray counters, [8, 16) dr_stats.diag, [16, 48) the shade / refill phase's budget (dr_stats_phase_counts)

// the launch configuration has builds that store every frame of a batch into its own buffer (RenderParams::out_frame_stride != 0)
bool persistent_kernel_can_store_per_frame(const PersistentCfg& cfg);

// kernels_aux.hip
void launch_tile_feedback(hipStream_t stream, const unsigned* pixel_cost, unsigned* tile_cost, int* tile_order, int* region_start, int tiles, int regions,
                          int heavy_factor, int split_steps, int split_limit);
void launch_present(hipStream_t stream, const int32_t* acc, uint8_t* rgb, int W, int H, int div);
void launch_frame_add(hipStream_t stream, int32_t* acc, const int32_t* frame, size_t n);      // acc += frame (pipelined single frames)
void launch_stripe_copy(hipStream_t stream, int32_t* dst, const int32_t* src, int ncols, int run4, long long dst_first4, long long dst_stride4,
                        long long src_first4, long long src_stride4);
void launch_gather_probe(hipStream_t stream, const RenderParams& P, int blocks, unsigned nrec, int iters, unsigned* out);
// trace-only probe (dr_context_probe_trace): variant 0 = one ray per lane, waves wait for their slowest; 1.. = persistent waves refilling from the ray list
// (occupancy / lanes that must be free before a refill / lanes at a leaf before a leaf step: 6/1/20, 6/8/20, 6/16/20, 8/8/20, 8/8/28, 6/8/28, 8/4/32)
void launch_trace_probe(hipStream_t stream, const RenderParams& P, int num_cus, int variant, const float* rays, unsigned n, unsigned* cursor, unsigned* out);
void launch_kat_rng(hipStream_t stream, uint64_t seed, int n, double* out);
void launch_kat_aabb(hipStream_t stream, int n, const float* o, const float* d, const float* mn, const float* mx, int32_t* hit, float* dist);
void launch_kat_node_planes(hipStream_t stream, int n, const uint32_t* w, const float* a, const float* b, float* t_mix, float* t_cvt);
void launch_kat_tri(hipStream_t stream, int n, const float* o, const float* d, const float* v0, const float* v1, const float* v2, float* t);
void launch_kat_sphere(hipStream_t stream, int n, const float* o, const float* d, const float* c, const float* r, float* t);
void launch_kat_optics(hipStream_t stream, int n, const float* v, const float* nrm, const float* eta, float* refl, float* refr, float* sch);
void launch_kat_normal(hipStream_t stream, const RenderParams& P, int n, const int32_t* slot, const float* o, const float* d, const float* t, float* nrm, float* texco);
void launch_kat_hit(hipStream_t stream, const RenderParams& P, int traversal, int n, const float* o, const float* d, float* t, int32_t* slot, int32_t* visits);

}  // namespace dr
