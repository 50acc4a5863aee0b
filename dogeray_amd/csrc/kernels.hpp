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
constexpr int WAVE_LOG_WAVES = 49152;    // waves the wave log (option wave_log) has room for

// What a launch of the persistent kernel needs to know of the context's options (dr_context_set_option)
struct PersistentCfg {
  int traversal;             // the traversal the launch really uses (DR_TRAVERSAL_WIDE or DR_TRAVERSAL_THREADED)
  int occupancy;             // 4, 5 or 6 waves per SIMD
  int schedule;              // 0, 1, 2: option "schedule" (kernels_render.hip launch_persistent_occ)
  int num_cus;
  int coop_tiles_per_wave;
  bool count;                // counting build
  int wgs_per_cu = 0;        // 0: as many as the build's occupancy allows; otherwise this many workgroups per CU (one half of a duo launch)
};

// Hand-off between the kernels of a short launch's chain (persistent_kernel.hpp: template parameter HAND; kernels_handoff.hip)
constexpr int HANDOFF_WORDS = 29 + WIDE_STACK;      // 32-bit words per path in a hand-off list
constexpr int HANDOFF_CTL_WORDS = 8;                 // control words of a launch behind its MAX_REGIONS tile counters: [k] = number of paths in the list stage k - 1 wrote
constexpr int HANDOFF_FLAG_WORDS = 64 * 64;          // a render stream's "the tile queues are empty" flag: 64 copies, 256 bytes apart
struct Handoff {
  unsigned* ctl;             // the launch's control words (zero when the chain starts)
  const uint32_t* in;        // list this kernel starts from (HAND 2): word w of entry i at in[w * cap + i]
  uint32_t* out;             // list it dumps into (null: it runs to the end)
  int32_t cap;               // entries a list has room for
  int32_t in_count, out_count;   // which control words count `in` and `out`
  int32_t wait;              // loop iterations before the dump: HAND 1 after the wave knows the queues are empty, HAND 2 (with a list to dump into) after its start
  int32_t margin;            // HAND 1: a wave watches the "queues empty" flag once its newest tile is among the last `margin` positions of its queue
  uint32_t* flags;           // HAND 1: the stream's flag (HANDOFF_FLAG_WORDS words; copy k at flags[k * 64]); up = holds `epoch`
  uint32_t epoch;            // ... this launch's number among the stream's chains (>= 1)
  int32_t resident_waves;    // HAND 2: waves of this launch that are resident at once
};
// What context.cpp decides about a short launch's chain (options "handoff", "handoff_mid", "handoff_mid_wait")
struct HandoffPlan {
  unsigned* ctl;             // HANDOFF_CTL_WORDS zeroed words
  uint32_t* list[2];         // two lists of `cap` entries (HANDOFF_WORDS * cap words each)
  uint32_t* flags; uint32_t epoch;
  int32_t cap;
  int32_t wait;              // iterations a wave of the first stage goes on after it knows the queues are empty
  int32_t mid;               // 0: tiles -> final; 1: a lean stage between them (no tiles, no sharing), 2: a work-sharing-build stage without sharing
  int32_t mid_wait;          // iterations the middle stage runs before it dumps
};

// kernels_render.hip
void launch_tile_kernel(hipStream_t stream, const RenderParams& P, int traversal, bool count, int occupancy);
// returns the number of waves that write the wave log (0: the launched build does not log)
int launch_persistent_kernel(hipStream_t stream, const RenderParams& P, const PersistentCfg& cfg, unsigned* tile_counter, const int* order,
                             const int* region_start, unsigned* pixel_cost);
// the work-sharing build would render this launch (a short one): what the hand-off chain replaces
bool persistent_launch_is_short(const RenderParams& P, const PersistentCfg& cfg);

// kernels_handoff.hip: a short launch of the wide walk as a chain of kernels (lean build on the tile queues -> [middle stage] -> work-sharing build on the
// paths left); returns the number of waves that wrote the wave log (stage after stage, WAVE_LOG_STAGE_WAVES entries apart)
int launch_handoff_chain(hipStream_t stream, const RenderParams& P, const PersistentCfg& cfg, unsigned* tile_counter, const int* order,
                         const int* region_start, unsigned* pixel_cost, const HandoffPlan& plan);
constexpr int WAVE_LOG_STAGE_WAVES = 16384;  // wave-log entries reserved per stage of a chain

constexpr int COUNTER_WORDS = 16;   // 64-bit words of a context's statistics buffer: [0, 8) ray counters, [8, 16) dr_stats.diag

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

}  // namespace dr
