// Launchers of the device code, one translation unit per family of kernels:
//   kernels_render.hip        the per-tile kernel (reference launch shape) and the persistent kernel (waves as pools of 64 path slots)
//   kernels_pool.hip          the pool kernel: every wave owns 128 paths in LDS and runs one kind of step (node / leaf / shade) at a time, at full width
//   kernels_aux.hip           tile-order feedback, present divide, stripe copies of the multi-GPU gather, gather probe, known-answer kernels
//   kernels_experimental.hip  round 2's two measured-slower kernels (two paths per lane, waves with roles); only in
//                             -DDOGERAY_EXPERIMENTAL builds (tools/exp_variant.sh), not in the product library
// context.cpp (host only: resident scene, options, the C ABI) calls these and never sees a kernel.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "device_layout.h"

namespace dr {

constexpr int MAX_REGIONS = 8;           // tile queues of the persistent kernels (one per XCD)
constexpr int WAVE_LOG_WAVES = 16384;    // waves the wave log (option wave_log) has room for
#ifndef DR_WAVE_LOG_DETAIL
#define DR_WAVE_LOG_DETAIL 0             // experiment builds: the wave log also counts phases, hand-overs and walking lanes of the drain
#endif
constexpr size_t PIXEL_LOG_WORDS = DR_WAVE_LOG_DETAIL ? (size_t)2 * 4096 * 4096 : 0;   // experiment builds: start and end stamp of every pixel behind the wave log

// What a launch of the persistent kernel needs to know of the context's options (dr_context_set_option)
struct PersistentCfg {
  int traversal;             // the traversal the launch really uses (DR_TRAVERSAL_WIDE or DR_TRAVERSAL_THREADED)
  int occupancy;             // 4, 5 or 6 waves per SIMD
  int trav_min, park_min, unroll;
  int num_cus;
  int coop_tiles_per_wave;
  bool count;                // counting build
};

// kernels_render.hip
void launch_tile_kernel(hipStream_t stream, const RenderParams& P, int traversal, bool count, int occupancy);
// returns the number of waves that write the wave log (0: the launched build does not log)
int launch_persistent_kernel(hipStream_t stream, const RenderParams& P, const PersistentCfg& cfg, unsigned* tile_counter, const int* order,
                             const int* region_start, unsigned* pixel_cost);

// kernels_pool.hip
struct PoolCfg {
  int num_cus;
  int shade_min;             // a wave shades once this many of its paths wait for it (or nothing else can be done)
  int shape;                 // 0: 4 stack words in LDS, 15 waves per CU; 1: 3 words, 16 waves; 2: 8 words, 12 waves
  bool diag;                 // the build that counts batches, paths per batch and cycles per part of the loop (counters[16..])
};
constexpr int COUNTER_WORDS = 64;   // 64-bit words of a context's statistics buffer: [0, 8) ray counters, [8, 16) dr_stats.diag, [16, 48) kernel diagnostics
bool pool_kernel_can_render(const RenderParams& P);       // the launch shapes the pool kernel covers (wide walk resident, one sample per pixel ...)
void launch_pool_kernel(hipStream_t stream, const RenderParams& P, const PoolCfg& cfg, unsigned* tile_counter, const int* order,
                        const int* region_start, unsigned* pixel_cost, unsigned* scratch);
size_t pool_scratch_words(int num_cus);                   // global scratch of a launch (stack overflow words), 32-bit words

// kernels_aux.hip
void launch_tile_feedback(hipStream_t stream, const unsigned* pixel_cost, unsigned* tile_cost, int* tile_order, int* region_start, int tiles, int regions,
                          int heavy_factor, int split_steps, int split_limit);
void launch_present(hipStream_t stream, const int32_t* acc, uint8_t* rgb, int W, int H, int div);
void launch_frame_add(hipStream_t stream, int32_t* acc, const int32_t* frame, size_t n);      // acc += frame (pipelined single frames)
void launch_stripe_copy(hipStream_t stream, int32_t* dst, const int32_t* src, int ncols, int run4, long long dst_first4, long long dst_stride4,
                        long long src_first4, long long src_stride4);
void launch_gather_probe(hipStream_t stream, const RenderParams& P, int blocks, unsigned nrec, int iters, unsigned* out);
void launch_kat_rng(hipStream_t stream, uint64_t seed, int n, double* out);
void launch_kat_aabb(hipStream_t stream, int n, const float* o, const float* d, const float* mn, const float* mx, int32_t* hit, float* dist);
void launch_kat_node_planes(hipStream_t stream, int n, const uint32_t* w, const float* a, const float* b, float* t_mix, float* t_cvt);
void launch_kat_tri(hipStream_t stream, int n, const float* o, const float* d, const float* v0, const float* v1, const float* v2, float* t);
void launch_kat_sphere(hipStream_t stream, int n, const float* o, const float* d, const float* c, const float* r, float* t);
void launch_kat_optics(hipStream_t stream, int n, const float* v, const float* nrm, const float* eta, float* refl, float* refr, float* sch);
void launch_kat_normal(hipStream_t stream, const RenderParams& P, int n, const int32_t* slot, const float* o, const float* d, const float* t, float* nrm, float* texco);
void launch_kat_hit(hipStream_t stream, const RenderParams& P, int traversal, int n, const float* o, const float* d, float* t, int32_t* slot, int32_t* visits);

// kernels_experimental.hip (stubs that return false in the product build)
constexpr int EXPERIMENTAL_PATH_UNITS = 7;      // 16-byte units per path record of the two experimental kernels
bool experimental_built();
bool launch_paired_kernel(hipStream_t stream, const RenderParams& P, int blocks, int pair_thresh, unsigned* tile_counter, const int* order,
                          const int* region_start, unsigned* pixel_cost, void* paths);
bool launch_roles_kernel(hipStream_t stream, const RenderParams& P, int roles, int blocks, unsigned* tile_counter, void* paths, unsigned* abort_flag);
int roles_blocks(int roles, int num_cus);       // workgroups a roles launch uses
size_t roles_path_waves(int roles, int blocks); // waves' worth of path records it needs

}  // namespace dr
