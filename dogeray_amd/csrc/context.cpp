// Device context: resident scene, options, render launches, on-device accumulation, stats, KAT hooks -- the host side of the C ABI.
// Replaces CudaStarter (kernel.cu K:2562-2669), which mallocs, uploads the whole scene, launches, synchronises, downloads and
// frees on every call.  No kernels here: kernels.hpp declares their launchers.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "linearise.hpp"
#include "params_host.hpp"
#include "scene_host.hpp"

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
  DevUnit* wide = nullptr; size_t wide_bytes = 0; int wide_depth = 0, wide_nodes = 0; float wide_pmax = 0; WideMu wide_mu = {0, 0, 0}; int wide_own_bounds = 0;   // null: scene not representable (threaded walk is used)
  int wide_tree = 2;        // structure of the wide walk's tree: 2 binned SAH with small triangles entered by their own bounds (default), 1 binned SAH over the reference's leaf boxes, 0 the reference's topology collapsed
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
  int schedule = 0;         // persistent kernel: 0 = shade / refill below 32 walking lanes, leaf steps for 20 lanes, two steps per iteration (tuned); 1 = 32 / 8 / 1; 2 = 48 / leaves on the spot / 1
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
  int reserve_cus = 0;      // persistent kernel: launch workgroups for this many CUs fewer than the device has (room for a gather's copy / RCCL kernels beside the rendering)
  int wave_log_on = 0;      // persistent kernel writes begin / queue-empty / end stamps of every wave (dr_stats_wave_log)
  unsigned long long* wave_log = nullptr; int wave_log_waves = 0;
  int batch_frames = 32;    // persistent kernel: at most this many frames per launch in dr_render_accumulate
  float cur_settings[13] = {0};
  dr_stats stats;
  // pipelined single frames (dr_pipeline_*): a second render stream, a stream that folds finished frames into the accumulator in
  // frame order, and PIPE_DEPTH frame buffers / present buffers that rotate
  static constexpr int PIPE_STREAMS = 4;           // render streams (stream itself is number 0); option pipe_streams uses 2 .. 4 of them
  static constexpr int PIPE_DEPTH = PIPE_STREAMS + 1;
  int pipe_streams = 2;     // render streams the pipeline alternates between
  int pipe_lean = 0;        // pipelined launches run the lean build with one queue per XCD instead of the work-sharing build (their tails overlap other frames)
  hipStream_t pipe_stream[PIPE_STREAMS] = {nullptr, nullptr, nullptr, nullptr}, acc_stream = nullptr;      // pipe_stream[0] = stream
  // a slot = one GROUP of frames in flight: frames submitted one after the other (same view, seeds in arithmetic progression) share one launch -- each
  // rendered into a buffer of its own -- and are added to the accumulator and presented one by one, in ticket order (option pipe_group; 1 = a launch per frame)
  static constexpr int PIPE_GROUP_MAX = 16;
  struct PipeSlot {
    int32_t* frames = nullptr; size_t elems_each = 0; int cap_frames = 0;      // cap_frames buffers of elems_each int32, one after the other
    int rect[6] = {-1, 0, 0, 0, 0, 0};             // W, H, gx, gy, stripe mod, stripe rem the buffers were last rendered with (their margins are 0)
    hipEvent_t rendered = nullptr;                 // end of the group's launch
    hipEvent_t added[PIPE_GROUP_MAX] = {nullptr};  // frame f of the group has been added (and presented)
    uint8_t* rgb_dev[PIPE_GROUP_MAX] = {nullptr}; uint8_t* rgb_host[PIPE_GROUP_MAX] = {nullptr}; size_t rgb_bytes[PIPE_GROUP_MAX] = {0};
    int div[PIPE_GROUP_MAX] = {0};                 // divisor frame f was presented with (0: not presented)
    bool fwaited[PIPE_GROUP_MAX] = {false};        // dr_pipeline_wait has returned for frame f
    uint64_t first = 0; int count = 0;             // tickets [first, first + count)
    bool drained = true;                           // the host has waited for the group's last add: its buffers may be reused at once
  };
  PipeSlot pipe_slot[PIPE_DEPTH];
  uint64_t pipe_groups = 0;                        // groups launched so far (slot = group % (streams + 1), render stream = group % streams)
  int pipe_group = 8;                              // most frames per group
  struct PipePending { float st[13]; int W, H; float bg; uint64_t seed; int div; };
  std::vector<PipePending> pipe_pending;           // submitted, not launched yet: tickets [pipe_next - size, pipe_next)
  uint64_t pipe_next = 0;                          // ticket of the next frame
  hipEvent_t pipe_last[PIPE_STREAMS] = {nullptr, nullptr, nullptr, nullptr};    // end of the newest launch on each render stream
  bool pipe_last_set[PIPE_STREAMS] = {false, false, false, false};
  hipEvent_t pipe_barrier = nullptr; bool pipe_barrier_set = false;      // end of the newest tile-order refresh: later launches read that order
  hipEvent_t pipe_sync = nullptr;                  // orders the pipeline after earlier work on `stream`
  bool pipe_dirty = false;                         // frames have gone through the pipeline since the last join
  bool pipe_ready = false;                         // every stream and event of the pipeline exists (pipeline_setup)
  bool pipe_hold_order = false;                    // enqueue_frame: use the stored tile order as it is, record no costs (a launch beside another one)
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

// settings[13] -> per-launch constants: the view (params_host.hpp: the camera block K:1016-1052, evaluated once on the host), the
// resident scene, and the scheduling options.
int make_params(dr_context* c, const float* st, int W, int H, float background, uint64_t seed, RenderParams& P, int batch_hint = 1) {
  if (!c->walk) { set_error("no scene uploaded"); return DR_ERR_INVALID; }
  memset(&P, 0, sizeof(P));
  if (const char* why = fill_view_params(st, W, H, background, seed, c->stripe_mod, c->stripe_rem, P)) { set_error(why); return DR_ERR_INVALID; }
  if (P.backtex >= c->n_tex) { set_error("backtex refers to a texture that is not loaded"); return DR_ERR_INVALID; }
  memcpy(c->cur_settings, st, sizeof(c->cur_settings));
  P.walk = c->walk; P.walk_bytes = (uint32_t)c->walk_bytes; P.pairs = c->pairs; P.prims = c->prims; P.shade = c->shade; P.tex = c->tex; P.texels = c->texels;
  P.wide = c->wide; P.wide_bytes = (uint32_t)c->wide_bytes; P.wide_pmax = c->wide_pmax; P.wide_mu = c->wide_mu;
  P.counters = c->counters;
  P.wave_log = c->wave_log_on ? c->wave_log : nullptr;
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

PersistentCfg persistent_cfg(const dr_context* c) {
  PersistentCfg cfg;
  cfg.traversal = traversal_of(c); cfg.occupancy = c->occupancy; cfg.schedule = c->schedule;
  cfg.num_cus = c->num_cus - c->reserve_cus > 0 ? c->num_cus - c->reserve_cus : 1; cfg.coop_tiles_per_wave = c->coop_tiles_per_wave; cfg.count = c->count;
  return cfg;
}

// enqueue one launch (P.batch frames); no events, no sync
void enqueue_frame(dr_context* c, const RenderParams& P_in) {
  RenderParams P = P_in;
  const int tiles = P.ncols * P.gy;
  if (uses_persistent(c)) {
    if (c->tile_cursor + MAX_REGIONS > TILE_COUNTERS) {
      (void)hipMemsetAsync(c->tile_counters, 0, TILE_COUNTERS * sizeof(unsigned), c->stream);
      c->tile_cursor = 0;
    }
    unsigned* counter = c->tile_counters + c->tile_cursor;      // one counter per region
    c->tile_cursor += MAX_REGIONS;
    const int* order; unsigned* pcost;
    if (c->pipe_hold_order) {
      // a pipelined launch runs beside the previous frame's: it may read the tile order but nobody may write it (or the costs) meanwhile
      const float geom[5] = {(float)P.W, (float)P.H, (float)(P.stripe_mod * 1024 + P.stripe_rem) + 0.125f * (float)P.regions, (float)P.ncols, (float)P.gy};
      order = (c->order_valid && c->order_capacity >= tiles && memcmp(c->order_key + 13, geom, sizeof(geom)) == 0) ? c->tile_order : nullptr;
      pcost = nullptr;
    } else feedback_buffers(c, P, tiles, order, pcost);
    if (!c->wave_log_on) P.wave_log = nullptr;
    c->wave_log_waves = launch_persistent_kernel(c->stream, P, persistent_cfg(c), counter, order, c->region_start, pcost);
    // next launch's order from this launch's costs (stream-ordered, no host sync).  The view does not change between the frames of
    // a progressive render, so after the first two launches of a view the order is refreshed every feedback_every-th launch only
    // (the two kernels take 75 us: nothing for a launch of 32 frames, 6 % of a launch of one)
    if (pcost && !order) c->order_age = 0;
    if (pcost && (c->order_age < 2 || c->order_age % c->feedback_every == 0)) {
      launch_tile_feedback(c->stream, c->pixel_cost, c->tile_cost, c->tile_order, c->region_start, tiles, P.regions, c->heavy_factor, c->split_steps,
                           c->split_parts > 1 ? (int)((long long)c->num_cus * (c->occupancy >= 5 ? 5 : 4) * 4 * c->split_waves / (100 * c->split_parts)) : 0);      // at most split_waves % of the waves start with a part of a split tile
      c->order_valid = true;
    }
    c->order_age++;
    return;
  }
  launch_tile_kernel(c->stream, P, traversal_of(c), c->count, c->occupancy);
}

int join_pipeline(dr_context* c);

int set_option(dr_context* c, const std::string& name, int v) {
  if (name == "kernel") { if (v != DR_KERNEL_TILE && v != DR_KERNEL_PERSISTENT) goto bad; c->kernel = v; }
  else if (name == "occupancy") { if (v != 4 && v != 5 && v != 6) goto bad; c->occupancy = v; }
  else if (name == "schedule") { if (v < 0 || v > 2) goto bad; c->schedule = v; }
  else if (name == "heavy_factor") { if (v < -1 || v > 1000) goto bad; c->heavy_factor = v; c->order_valid = false; }
  else if (name == "coop_steps") { if (v < 0) goto bad; c->coop_steps = v; }
  else if (name == "coop_lanes") { if (v < 1 || v > 64) goto bad; c->coop_lanes = v; }
  else if (name == "split_parts") { if (v != 1 && v != 2 && v != 4 && v != 8) goto bad; c->split_parts = v; }
  else if (name == "split_waves") { if (v < 1 || v > 1000) goto bad; c->split_waves = v; c->order_valid = false; }
  else if (name == "split_steps") { if (v < 16 || v > 4080) goto bad; c->split_steps = v & ~15; c->order_valid = false; }
  else if (name == "short_one_queue") { c->short_one_queue = v != 0; c->order_valid = false; }
  else if (name == "coop_rounds") { if (v < 1 || v > 16) goto bad; c->coop_rounds = v; }
  else if (name == "reserve_cus") { if (v < 0 || v > 64) goto bad; c->reserve_cus = v; }
  else if (name == "wave_log") {
    if (v != 0 && v != 1) goto bad;
    if (v && !c->wave_log) {
      const size_t bytes = (size_t)WAVE_LOG_WAVES * 16 * sizeof(unsigned long long);
      if (hipSetDevice(c->device) != hipSuccess || hipMalloc((void**)&c->wave_log, bytes) != hipSuccess) { c->wave_log = nullptr; set_error("cannot allocate the wave log"); return DR_ERR_DEVICE; }
      (void)hipMemsetAsync(c->wave_log, 0, bytes, c->stream);
    }
    c->wave_log_on = v;
  }
  else if (name == "coop_tiles_per_wave") { if (v < 0) goto bad; c->coop_tiles_per_wave = v; }
  else if (name == "pipe_streams") {
    if (v < 2 || v > dr_context::PIPE_STREAMS) goto bad;
    if (v != c->pipe_streams) {               // slots and streams are numbered by ticket: drain, then start again from ticket 0
      if (hipSetDevice(c->device) != hipSuccess || join_pipeline(c) != DR_OK || hipStreamSynchronize(c->stream) != hipSuccess) { set_error("pipe_streams: cannot drain the pipeline"); return DR_ERR_DEVICE; }
      for (int k = 0; k < dr_context::PIPE_DEPTH; k++) { c->pipe_slot[k].drained = true; c->pipe_slot[k].count = 0; }
      c->pipe_next = 0; c->pipe_groups = 0; c->pipe_streams = v;
    }
  }
  else if (name == "pipe_lean") { c->pipe_lean = v != 0; }
  else if (name == "pipe_group") { if (v < 1 || v > dr_context::PIPE_GROUP_MAX) goto bad; c->pipe_group = v; }
  else if (name == "xcd_regions") { c->xcd_regions = v != 0; c->order_valid = false; }
  else if (name == "batch_frames") { if (v < 1 || v > 256) goto bad; c->batch_frames = v; }
  else if (name == "feedback") { c->feedback = v != 0; c->order_valid = false; }
  else if (name == "order_follows_camera") { c->order_follows_camera = v != 0; }
  else if (name == "feedback_every") { if (v < 1) goto bad; c->feedback_every = v; }
  else if (name == "wide_tree") { if (v < 0 || v > 2) goto bad; c->wide_tree = v; }      // takes effect at the next dr_context_upload_scene
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

int collect_time(dr_context* c, uint64_t frames, uint64_t samples) {
  HIP_TRY(hipEventSynchronize(c->ev1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->stats.kernel_ms += ms;
  c->stats.frames += frames;
  c->stats.samples += samples;
  return DR_OK;
}

// Work queued through the pipeline runs on two more streams: everything else (which uses `stream`) is ordered behind it here.
int pipeline_flush(dr_context* c);
int join_pipeline(dr_context* c) {
  if (!c->pipe_pending.empty()) { const int rc = pipeline_flush(c); if (rc != DR_OK) return rc; }
  if (!c->pipe_dirty) return DR_OK;
  for (int k = 0; k < dr_context::PIPE_DEPTH; k++) {
    const dr_context::PipeSlot& sl = c->pipe_slot[k];
    if (sl.count > 0) HIP_TRY(hipStreamWaitEvent(c->stream, sl.added[sl.count - 1], 0));      // (adds run in order: the last one ends the group)
  }
  for (int k = 1; k < dr_context::PIPE_STREAMS; k++) if (c->pipe_last_set[k]) HIP_TRY(hipStreamWaitEvent(c->stream, c->pipe_last[k], 0));
  c->pipe_dirty = false;
  return DR_OK;
}

int pipeline_setup(dr_context* c) {
  if (c->pipe_ready) return DR_OK;
  // (a failed attempt leaves what it created in place -- dr_context_destroy releases it -- and pipe_ready false: the next call creates what is missing)
  bool ok = c->acc_stream || hipStreamCreateWithFlags(&c->acc_stream, hipStreamNonBlocking) == hipSuccess;
  c->pipe_stream[0] = c->stream;
  for (int k = 1; k < dr_context::PIPE_STREAMS; k++) ok = ok && (c->pipe_stream[k] || hipStreamCreateWithFlags(&c->pipe_stream[k], hipStreamNonBlocking) == hipSuccess);
  if (!ok) { set_error("pipeline: cannot create streams"); return DR_ERR_DEVICE; }
  auto event = [](hipEvent_t& e) { return e || hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess; };
  for (int k = 0; k < dr_context::PIPE_DEPTH; k++) {
    ok = ok && event(c->pipe_slot[k].rendered);
    for (int f = 0; f < dr_context::PIPE_GROUP_MAX; f++) ok = ok && event(c->pipe_slot[k].added[f]);
  }
  for (int k = 0; k < dr_context::PIPE_STREAMS; k++) ok = ok && event(c->pipe_last[k]);
  ok = ok && event(c->pipe_barrier) && event(c->pipe_sync);
  if (!ok) { set_error("pipeline: cannot create events"); return DR_ERR_DEVICE; }
  c->pipe_ready = true;
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
      hipMalloc((void**)&c->counters, COUNTER_WORDS * sizeof(unsigned long long)) != hipSuccess ||
      hipMemsetAsync(c->counters, 0, COUNTER_WORDS * sizeof(unsigned long long), c->stream) != hipSuccess ||
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
  // every stream that may still touch the buffers, before they are freed
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int k = 1; k < dr_context::PIPE_STREAMS; k++) if (c->pipe_stream[k]) (void)hipStreamSynchronize(c->pipe_stream[k]);
  if (c->acc_stream) (void)hipStreamSynchronize(c->acc_stream);
  void* bufs[] = {c->wave_log, c->packed[0], c->packed[1], c->walk, c->wide, c->pairs, c->prims, c->shade, c->tex, c->texels, c->frame, c->accum, c->present, c->counters, c->tile_counters, c->pixel_cost, c->tile_cost, c->tile_order, c->region_start};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (int k = 0; k < 2; k++) { if (c->pev0[k]) (void)hipEventDestroy(c->pev0[k]); if (c->pev1[k]) (void)hipEventDestroy(c->pev1[k]); }
  for (int k = 0; k < dr_context::PIPE_DEPTH; k++) {
    dr_context::PipeSlot& sl = c->pipe_slot[k];
    if (sl.frames) (void)hipFree(sl.frames);
    if (sl.rendered) (void)hipEventDestroy(sl.rendered);
    for (int f = 0; f < dr_context::PIPE_GROUP_MAX; f++) {
      if (sl.rgb_dev[f]) (void)hipFree(sl.rgb_dev[f]);
      if (sl.rgb_host[f]) (void)hipHostFree(sl.rgb_host[f]);
      if (sl.added[f]) (void)hipEventDestroy(sl.added[f]);
    }
  }
  for (hipEvent_t e : {c->pipe_last[0], c->pipe_last[1], c->pipe_last[2], c->pipe_last[3], c->pipe_barrier, c->pipe_sync}) if (e) (void)hipEventDestroy(e);
  for (int k = 1; k < dr_context::PIPE_STREAMS; k++) if (c->pipe_stream[k]) (void)hipStreamDestroy(c->pipe_stream[k]);
  if (c->acc_stream) (void)hipStreamDestroy(c->acc_stream);
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
  c->wide_bytes = 0; c->wide_depth = img.wide_depth; c->wide_nodes = img.wide_nodes; c->wide_pmax = img.wide_pmax; c->wide_mu = img.wide_mu; c->wide_own_bounds = img.wide_own_bounds;
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
  else if (n == "schedule") *value = c->schedule;
  else if (n == "xcd_regions") *value = c->xcd_regions;
  else if (n == "heavy_factor") *value = c->heavy_factor;
  else if (n == "coop_steps") *value = c->coop_steps;
  else if (n == "coop_lanes") *value = c->coop_lanes;
  else if (n == "wave_log") *value = c->wave_log_on;
  else if (n == "coop_rounds") *value = c->coop_rounds;
  else if (n == "reserve_cus") *value = c->reserve_cus;
  else if (n == "short_one_queue") *value = c->short_one_queue;
  else if (n == "split_parts") *value = c->split_parts;
  else if (n == "split_steps") *value = c->split_steps;
  else if (n == "split_waves") *value = c->split_waves;
  else if (n == "coop_tiles_per_wave") *value = c->coop_tiles_per_wave;
  else if (n == "pipe_streams") *value = c->pipe_streams;
  else if (n == "pipe_lean") *value = c->pipe_lean;
  else if (n == "pipe_group") *value = c->pipe_group;
  else if (n == "tree_depth") *value = c->tree_depth;
  else if (n == "wide_tree") *value = c->wide_tree;
  else if (n == "wide_own_bounds") *value = c->wide ? c->wide_own_bounds : 0;
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
  { const int jrc = join_pipeline(c); if (jrc != DR_OK) return jrc; }
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
  { const int jrc = join_pipeline(c); if (jrc != DR_OK) return jrc; }
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
  { const int jrc = join_pipeline(c); if (jrc != DR_OK) return jrc; }
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
  return DR_OK;
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
  { const int jrc = join_pipeline(c); if (jrc != DR_OK) return jrc; }
  int rc;
  if ((rc = collect_pending(c, c->pending_next)) != DR_OK) return rc;       // older first
  if ((rc = collect_pending(c, c->pending_next ^ 1)) != DR_OK) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DR_OK;
}

}  // extern "C"

namespace {

// how many frames a group may hold right now: the option, if the launch configuration has the builds that store every frame of a batch separately
int pipeline_group_size(const dr_context* c) {
  if (c->pipe_group <= 1 || c->pipe_lean || !uses_persistent(c) || !persistent_kernel_can_store_per_frame(persistent_cfg(c))) return 1;
  return c->pipe_group;
}

// the slot that holds a launched ticket (null: not in flight any more, or still pending)
dr_context::PipeSlot* pipeline_slot_of(dr_context* c, uint64_t ticket) {
  for (dr_context::PipeSlot& sl : c->pipe_slot)
    if (sl.count > 0 && ticket >= sl.first && ticket < sl.first + (uint64_t)sl.count) return &sl;
  return nullptr;
}

int pipeline_flush_some(dr_context* c, int n);

// launches the frames submitted since the last launch: as ONE group if the launch configuration (still) has the builds that store every frame of a batch
// separately -- an option may have changed since the frames were submitted --, else one by one
int pipeline_flush(dr_context* c) {
  while (!c->pipe_pending.empty()) {
    const int all = (int)c->pipe_pending.size();
    const int rc = pipeline_flush_some(c, pipeline_group_size(c) >= all ? all : 1);
    if (rc != DR_OK) { c->pipe_pending.clear(); return rc; }
  }
  return DR_OK;
}

// launches the first n pending frames as one group
int pipeline_flush_some(dr_context* c, int n) {
  const dr_context::PipePending first = c->pipe_pending[0];
  const uint64_t stride = n > 1 ? c->pipe_pending[1].seed - first.seed : 0;
  const uint64_t first_ticket = c->pipe_next - (uint64_t)c->pipe_pending.size();
  const int W = first.W, H = first.H;
  RenderParams P;
  // (pipe_lean: the lean six-wave build and one queue per XCD, as for long launches -- the tail it leaves runs beside the next frames)
  const int saved_ctpw = c->coop_tiles_per_wave;
  if (c->pipe_lean) c->coop_tiles_per_wave = 0;
  int rc = make_params(c, first.st, W, H, first.bg, first.seed, P, n);
  c->coop_tiles_per_wave = saved_ctpw;
  if (rc != DR_OK) return rc;
  const uint64_t g = c->pipe_groups;
  const int nstreams = c->pipe_streams, depth = nstreams + 1;
  const int si = (int)(g % (uint64_t)nstreams);
  dr_context::PipeSlot& sl = c->pipe_slot[g % (uint64_t)depth];
  hipStream_t rs = c->pipe_stream[si];
  const size_t elems = (size_t)W * H * 3;
  if (!c->pipe_dirty) {                       // the first group after other work: the pipeline's streams start behind it
    HIP_TRY(hipEventRecord(c->pipe_sync, c->stream));
    for (int q = 1; q < dr_context::PIPE_STREAMS; q++) HIP_TRY(hipStreamWaitEvent(c->pipe_stream[q], c->pipe_sync, 0));
    HIP_TRY(hipStreamWaitEvent(c->acc_stream, c->pipe_sync, 0));
    c->pipe_dirty = true;
  }
  // the slot's previous group: its present buffers go back to the caller first, and its adds must have run
  if (!sl.drained && sl.count > 0) { HIP_TRY(hipEventSynchronize(sl.added[sl.count - 1])); sl.drained = true; }
  if (sl.count > 0) HIP_TRY(hipStreamWaitEvent(rs, sl.added[sl.count - 1], 0));
  if (sl.elems_each != elems || sl.cap_frames < n || !sl.frames) {
    if (sl.frames) { HIP_TRY(hipStreamSynchronize(c->acc_stream)); (void)hipFree(sl.frames); sl.frames = nullptr; sl.cap_frames = 0; }
    const int cap = n > c->pipe_group ? n : c->pipe_group;
    HIP_TRY(hipMalloc((void**)&sl.frames, (size_t)cap * elems * sizeof(int32_t)));
    sl.elems_each = elems; sl.cap_frames = cap;
    sl.rect[0] = -1;
  }
  // pixels outside the rendered block grid are 0 (K:2633-2636): the buffers are cleared when that grid changes, every frame of a grid
  // overwrites the same pixels
  const int rect[6] = {W, H, P.gx, P.gy, P.stripe_mod, P.stripe_rem};
  if (memcmp(rect, sl.rect, sizeof(rect)) != 0) {
    HIP_TRY(hipMemsetAsync(sl.frames, 0, (size_t)sl.cap_frames * elems * sizeof(int32_t), rs));
    memcpy(sl.rect, rect, sizeof(rect));
  }
  P.out = sl.frames;
  P.accumulate = 0;
  P.batch = n; P.batch_seed_stride = stride;
  P.out_frame_stride = n > 1 ? (uint32_t)elems : 0u;      // (a group of one is an ordinary launch: every build can render it)
  const int tiles = P.ncols * P.gy;
  // the tile order and the costs it is made from are shared by all launches: a launch that refreshes them runs alone (after the other
  // streams' newest launches, and the launches after it wait for the refresh); all others read the order as it is
  bool refresh = false;
  if (c->feedback && uses_persistent(c) && tiles > 0) {
    const float geom[5] = {(float)P.W, (float)P.H, (float)(P.stripe_mod * 1024 + P.stripe_rem) + 0.125f * (float)P.regions, (float)P.ncols, (float)P.gy};
    // (a refresh costs the overlap of two launches, 0.89 against 0.82 ms/frame when every 8th single-frame launch refreshes: four times rarer here)
    refresh = !c->order_valid || c->order_capacity < tiles || memcmp(c->order_key + 13, geom, sizeof(geom)) != 0 || c->order_age < 2 ||
              c->order_age % (4 * c->feedback_every) == 0;
  }
  if (c->pipe_barrier_set) HIP_TRY(hipStreamWaitEvent(rs, c->pipe_barrier, 0));
  // the tile counters are cleared (on this launch's stream) when the cursor wraps: like a refresh, that launch runs alone -- nobody may
  // still count on the old values, and nobody may start on the new ones before they are cleared
  const bool alone = refresh || c->tile_cursor + MAX_REGIONS > TILE_COUNTERS;
  if (alone)
    for (int q = 0; q < dr_context::PIPE_STREAMS; q++) if (q != si && c->pipe_last_set[q]) HIP_TRY(hipStreamWaitEvent(rs, c->pipe_last[q], 0));
  if (tiles > 0) {
    hipStream_t saved = c->stream;
    c->stream = rs; c->pipe_hold_order = !refresh;
    if (c->pipe_lean) c->coop_tiles_per_wave = 0;
    enqueue_frame(c, P);
    c->stream = saved; c->pipe_hold_order = false; c->coop_tiles_per_wave = saved_ctpw;
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(c->pipe_last[si], rs)); c->pipe_last_set[si] = true;
  if (alone) { HIP_TRY(hipEventRecord(c->pipe_barrier, rs)); c->pipe_barrier_set = true; }
  HIP_TRY(hipEventRecord(sl.rendered, rs));
  // fold into the accumulator, in ticket order (K:2213-2218), and make the image of exactly the frames so far (K:2287)
  HIP_TRY(hipStreamWaitEvent(c->acc_stream, sl.rendered, 0));
  for (int f = 0; f < n; f++) {
    launch_frame_add(c->acc_stream, c->accum, sl.frames + (size_t)f * elems, elems);
    const int div = c->pipe_pending[(size_t)f].div;
    sl.div[f] = 0; sl.fwaited[f] = false;
    if (div != 0) {
      const size_t bytes = (size_t)W * H * 3;
      if (sl.rgb_bytes[f] < bytes) {
        HIP_TRY(hipStreamSynchronize(c->acc_stream));
        if (sl.rgb_dev[f]) (void)hipFree(sl.rgb_dev[f]);
        if (sl.rgb_host[f]) (void)hipHostFree(sl.rgb_host[f]);
        sl.rgb_dev[f] = nullptr; sl.rgb_host[f] = nullptr; sl.rgb_bytes[f] = 0;
        HIP_TRY(hipMalloc((void**)&sl.rgb_dev[f], bytes));
        HIP_TRY(hipHostMalloc((void**)&sl.rgb_host[f], bytes, hipHostMallocDefault));
        sl.rgb_bytes[f] = bytes;
      }
      launch_present(c->acc_stream, c->accum, sl.rgb_dev[f], W, H, div);
      HIP_TRY(hipMemcpyAsync(sl.rgb_host[f], sl.rgb_dev[f], bytes, hipMemcpyDeviceToHost, c->acc_stream));
      sl.div[f] = div;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sl.added[f], c->acc_stream));
  }
  sl.first = first_ticket; sl.count = n; sl.drained = false;
  c->pipe_groups = g + 1;
  c->stats.launches += tiles > 0 ? 1 : 0;
  c->stats.frames += (uint64_t)n;
  c->stats.samples += (uint64_t)(tiles > 0 ? tiles : 0) * 64ull * (uint64_t)(P.spp_f > 0 ? ceilf(P.spp_f) : 0) * (uint64_t)n;
  c->pipe_pending.erase(c->pipe_pending.begin(), c->pipe_pending.begin() + n);
  return DR_OK;
}

}  // namespace

extern "C" {

int dr_pipeline_submit(dr_context* c, const float settings13[13], int W, int H, float background, uint64_t frame_seed, int present_divide_by,
                       uint64_t* ticket) {
  if (!c || !settings13) { set_error("null argument"); return DR_ERR_INVALID; }
  if (!c->accum || c->accW != W || c->accH != H) { set_error("call dr_accum_reset(W, H) first"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  int rc = pipeline_setup(c);
  if (rc != DR_OK) return rc;
  if (c->traversal == DR_TRAVERSAL_ORDERED && c->tree_depth > ORDERED_STACK) { set_error("tree too deep for ordered traversal"); return DR_ERR_SCENE; }
  {   // a frame that cannot be rendered must fail here, not when its group is launched
    RenderParams probe;
    if ((rc = make_params(c, settings13, W, H, background, frame_seed, probe, 1)) != DR_OK) return rc;
  }
  // a group holds frames of ONE view whose seeds are in arithmetic progression (what a progressive render submits): anything else starts a new group
  if (!c->pipe_pending.empty()) {
    const dr_context::PipePending& p0 = c->pipe_pending[0];
    const dr_context::PipePending& pl = c->pipe_pending.back();
    const bool same_view = memcmp(p0.st, settings13, 13 * sizeof(float)) == 0 && p0.W == W && p0.H == H && p0.bg == background;
    const bool in_step = c->pipe_pending.size() == 1 || frame_seed - pl.seed == c->pipe_pending[1].seed - p0.seed;
    if (!same_view || !in_step) { if ((rc = pipeline_flush(c)) != DR_OK) return rc; }
  }
  dr_context::PipePending p;
  memcpy(p.st, settings13, sizeof(p.st)); p.W = W; p.H = H; p.bg = background; p.seed = frame_seed; p.div = present_divide_by;
  c->pipe_pending.push_back(p);
  const uint64_t k = c->pipe_next;
  c->pipe_next = k + 1;
  if (ticket) *ticket = k;
  if ((int)c->pipe_pending.size() >= pipeline_group_size(c)) return pipeline_flush(c);
  return DR_OK;
}

int dr_pipeline_wait(dr_context* c, uint64_t ticket, uint8_t* out_rgb8) {
  if (!c || !c->pipe_ready) { set_error("pipeline: nothing submitted"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  if (ticket < c->pipe_next && ticket + c->pipe_pending.size() >= c->pipe_next) {      // submitted, its group not launched yet: it is now
    const int rc = pipeline_flush(c);
    if (rc != DR_OK) return rc;
  }
  dr_context::PipeSlot* sl = pipeline_slot_of(c, ticket);
  if (!sl) { set_error("pipeline: ticket not in flight (the pipeline keeps pipe_streams + 1 groups of frames)"); return DR_ERR_INVALID; }
  const int f = (int)(ticket - sl->first);
  HIP_TRY(hipEventSynchronize(sl->added[f]));
  sl->fwaited[f] = true;
  if (f == sl->count - 1) sl->drained = true;
  if (out_rgb8) {
    if (!sl->div[f]) { set_error("pipeline: that frame was submitted without a present"); return DR_ERR_INVALID; }
    memcpy(out_rgb8, sl->rgb_host[f], (size_t)c->accW * c->accH * 3);
  }
  return DR_OK;
}

int dr_pipeline_image(dr_context* c, uint64_t ticket, const uint8_t** rgb8) {
  if (!c || !c->pipe_ready || !rgb8) { set_error("pipeline: nothing submitted, or null argument"); return DR_ERR_INVALID; }
  dr_context::PipeSlot* sl = pipeline_slot_of(c, ticket);
  if (!sl) { set_error("pipeline: ticket not in flight (the pipeline keeps pipe_streams + 1 groups of frames)"); return DR_ERR_INVALID; }
  const int f = (int)(ticket - sl->first);
  if (!sl->fwaited[f]) { set_error("pipeline: dr_pipeline_wait(ticket) comes first"); return DR_ERR_INVALID; }
  if (!sl->div[f]) { set_error("pipeline: that frame was submitted without a present"); return DR_ERR_INVALID; }
  *rgb8 = sl->rgb_host[f];
  return DR_OK;
}

int dr_render_accumulate_pipelined(dr_context* c, const float settings13[13], int W, int H, float background, uint64_t frame_seed,
                                   uint64_t seed_stride, int nframes) {
  if (!c || nframes < 0) { set_error("bad argument"); return DR_ERR_INVALID; }
  uint64_t last = 0;
  for (int k = 0; k < nframes; k++) {
    const int rc = dr_pipeline_submit(c, settings13, W, H, background, frame_seed + (uint64_t)k * seed_stride, 0, &last);
    if (rc != DR_OK) return rc;
  }
  if (nframes > 0) {
    const int rc = dr_pipeline_wait(c, last, nullptr);      // adds run in order: the last one ends the batch
    if (rc != DR_OK) return rc;
    for (dr_context::PipeSlot& sl : c->pipe_slot) sl.drained = true;
  }
  return DR_OK;
}

int dr_context_stream(dr_context* c, void** hip_stream) {
  if (!c || !hip_stream) { set_error("null argument"); return DR_ERR_INVALID; }
  *hip_stream = (void*)c->stream;
  return DR_OK;
}

int dr_accum_reserve_pack(dr_context* c, int slot) {
  if (!c || !c->accum || (slot != 0 && slot != 1)) { set_error("pack: no accumulator, or slot not 0/1"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  const int gx = c->accW / 8;
  const size_t run = (size_t)8 * (size_t)c->accH * 3;                 // int32 per block column
  // sized for the largest stripe of this partition (rank 0's), so that every rank's buffer can take part in one
  // equal-sized gather
  const size_t need = (size_t)((gx + c->stripe_mod - 1) / c->stripe_mod > 0 ? (gx + c->stripe_mod - 1) / c->stripe_mod : 1) * run;
  return ensure(c->packed[slot], c->packed_elems[slot], need);
}

int dr_accum_pack_stripe(dr_context* c, int slot, void** dev_ptr, uint64_t* bytes) {
  if (!c || !c->accum || (slot != 0 && slot != 1)) { set_error("pack: no accumulator, or slot not 0/1"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  { const int jrc = join_pipeline(c); if (jrc != DR_OK) return jrc; }
  const int gx = c->accW / 8;
  const int ncols = gx > c->stripe_rem ? (gx - c->stripe_rem + c->stripe_mod - 1) / c->stripe_mod : 0;
  const size_t run = (size_t)8 * (size_t)c->accH * 3;                 // int32 per block column
  // sized for the largest stripe of this partition (rank 0's), so that every rank's buffer can take part in one
  // equal-sized gather
  int rc = dr_accum_reserve_pack(c, slot);
  if (rc != DR_OK) return rc;
  if (ncols > 0) {
    const int run4 = (int)(run / 4);
    launch_stripe_copy(c->stream, c->packed[slot], c->accum, ncols, run4, 0ll, (long long)run4, (long long)c->stripe_rem * run4, (long long)c->stripe_mod * run4);
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
  for (int r = first_rank; r < world; r++) {
    const int ncols = gx > r ? (gx - r + world - 1) / world : 0;
    if (ncols == 0) continue;
    if ((uint64_t)ncols * run * sizeof(int32_t) > rank_stride_bytes) { set_error("unpack: a rank's stripe is larger than rank_stride_bytes"); return DR_ERR_INVALID; }
    launch_stripe_copy(stream, c->accum, reinterpret_cast<const int32_t*>(packed_dev), ncols, run4, (long long)r * run4, (long long)world * run4,
                       (long long)((uint64_t)r * rank_stride_bytes / 16), (long long)run4);
  }
  HIP_TRY(hipGetLastError());
  return DR_OK;
}

int dr_accum_read(dr_context* c, int32_t* out_int3) {
  if (!c || !out_int3 || !c->accum) { set_error("no accumulator"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  { const int jrc = join_pipeline(c); if (jrc != DR_OK) return jrc; }
  HIP_TRY(hipMemcpyAsync(out_int3, c->accum, (size_t)c->accW * c->accH * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DR_OK;
}

int dr_accum_present(dr_context* c, int divide_by, uint8_t* out_rgb8) {
  if (!c || !out_rgb8 || !c->accum || divide_by == 0) { set_error("bad argument"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  { const int jrc = join_pipeline(c); if (jrc != DR_OK) return jrc; }
  size_t bytes = (size_t)c->accW * c->accH * 3;
  if (c->present_bytes < bytes) {
    if (c->present) (void)hipFree(c->present);
    c->present = nullptr; c->present_bytes = 0;
    HIP_TRY(hipMalloc((void**)&c->present, bytes));
    c->present_bytes = bytes;
  }
  launch_present(c->stream, c->accum, c->present, c->accW, c->accH, divide_by);
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
  HIP_TRY(hipMemsetAsync(c->counters, 0, COUNTER_WORDS * sizeof(unsigned long long), c->stream));
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

int dr_stats_phase_counts(dr_context* c, unsigned long long* out, int n) {
  if (!c || !out || n < 0 || n > COUNTER_WORDS - 16) { set_error("bad argument"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (n > 0) HIP_TRY(hipMemcpy(out, c->counters + 16, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
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
  launch_gather_probe(c->stream, P, blocks, nrec, iters / 8 + 1, out.p);      // warm-up
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  launch_gather_probe(c->stream, P, blocks, nrec, iters, out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *records_per_s = (double)blocks * 256.0 * (double)iters / ((double)ms * 1e-3);
  return DR_OK;
}

int dr_context_probe_trace(dr_context* c, const float settings13[13], int W, int H, float background, uint64_t frame_seed, int frames, int variant,
                           double* rays_per_s, uint64_t* n_rays, uint64_t* mismatches) {
  if (!c || !settings13 || !rays_per_s || !n_rays || !mismatches || frames < 1 || variant < 0) { set_error("bad argument"); return DR_ERR_INVALID; }
  if (!c->wide || traversal_of(c) != DR_TRAVERSAL_WIDE || !uses_persistent(c)) { set_error("the trace probe needs the wide walk and the persistent kernel"); return DR_ERR_INVALID; }
  HIP_TRY(hipSetDevice(c->device));
  int rc = dr_accum_reset(c, W, H);
  if (rc != DR_OK) return rc;
  // 1. the rays of ONE frame, as a per-bounce wavefront would hold them: the counting build writes every ray a path starts to slot bounce * pixels + pixel (tile
  // order); empty slots (paths that had ended) are squeezed out on the host; `frames` copies of the list make the probe's launch long enough to time
  RenderParams Pv;
  if ((rc = make_params(c, settings13, W, H, background, frame_seed, Pv, 1)) != DR_OK) return rc;
  const size_t npix = (size_t)Pv.ncols * Pv.gy * 64;
  const size_t slots = npix * (size_t)(Pv.max_depth > 0 ? Pv.max_depth : 1);
  if (slots * (size_t)frames > 0x7fffffffull) { set_error("too many rays"); return DR_ERR_INVALID; }
  DevBuf<float> raw; DevBuf<float> log; DevBuf<unsigned> cursor; DevBuf<unsigned> out, ref;
  if ((rc = raw.alloc(slots * 8)) != DR_OK) return rc;
  HIP_TRY(hipMemset(raw.p, 0, slots * 8 * sizeof(float)));
  const bool was_counting = c->count;
  unsigned long long ctl[3] = {0ull, (unsigned long long)(uintptr_t)raw.p, (unsigned long long)slots};      // statistics words 40-42: rays logged, the log, its room
  HIP_TRY(hipMemcpy(c->counters + 40, ctl, sizeof(ctl), hipMemcpyHostToDevice));
  c->count = true;
  rc = dr_render_accumulate(c, settings13, W, H, background, frame_seed, 1000003, 1);
  c->count = was_counting;
  const unsigned long long off[2] = {0ull, 0ull};
  HIP_TRY(hipMemcpy(c->counters + 41, off, sizeof(off), hipMemcpyHostToDevice));
  if (rc != DR_OK) return rc;
  std::vector<float> host(slots * 8), packed;
  if ((rc = raw.get(host.data(), host.size())) != DR_OK) return rc;
  packed.reserve(host.size() / 2);
  for (size_t k = 0; k < slots; k++) {
    const float* r = &host[k * 8];
    if (r[4] != 0.0f || r[5] != 0.0f || r[6] != 0.0f || r[4] != r[4]) packed.insert(packed.end(), r, r + 8);      // a direction was written
  }
  const size_t per_frame = packed.size() / 8;
  const unsigned n = (unsigned)(per_frame * (size_t)frames);
  *n_rays = per_frame;
  if (n == 0) { *rays_per_s = 0; *mismatches = 0; return DR_OK; }
  if ((rc = log.alloc((size_t)n * 8)) != DR_OK) return rc;
  for (int f = 0; f < frames; f++) HIP_TRY(hipMemcpy(log.p + (size_t)f * per_frame * 8, packed.data(), per_frame * 8 * sizeof(float), hipMemcpyHostToDevice));
  std::vector<float>().swap(host);
  if ((rc = cursor.alloc(1)) != DR_OK || (rc = out.alloc((size_t)n * 2)) != DR_OK || (rc = ref.alloc((size_t)n * 2)) != DR_OK) return rc;
  RenderParams P;
  memset(&P, 0, sizeof(P));
  P.wide = c->wide; P.wide_bytes = (uint32_t)c->wide_bytes; P.wide_pmax = c->wide_pmax; P.wide_mu = c->wide_mu;
  // 2. the probe, timed (one warm-up, then the best of three)
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    HIP_TRY(hipMemsetAsync(cursor.p, 0, sizeof(unsigned), c->stream));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    launch_trace_probe(c->stream, P, c->num_cus, variant, log.p, n, cursor.p, out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (rep > 0 && ms < best) best = ms;
  }
  *rays_per_s = (double)n / ((double)best * 1e-3);
  // 3. every result against the one-ray-per-lane walk
  launch_trace_probe(c->stream, P, c->num_cus, 0, log.p, n, cursor.p, ref.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  std::vector<unsigned> a((size_t)n * 2), b((size_t)n * 2);
  if ((rc = out.get(a.data(), a.size())) != DR_OK || (rc = ref.get(b.data(), b.size())) != DR_OK) return rc;
  uint64_t bad = 0;
  for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
  *mismatches = bad;
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
  launch_kat_rng(c->stream, seed, n, d.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return d.get(out, (size_t)n);
}

int dr_kat_aabb(dr_context* c, int n, const float* o, const float* d, const float* mn, const float* mx, int32_t* hit, float* dist) {
  KAT_PRE(n);
  DevBuf<float> bo, bd, bmn, bmx, bdist; DevBuf<int32_t> bhit;
  size_t m = (size_t)n * 3;
  KAT_DO(bo.alloc(m)); KAT_DO(bd.alloc(m)); KAT_DO(bmn.alloc(m)); KAT_DO(bmx.alloc(m)); KAT_DO(bdist.alloc((size_t)n)); KAT_DO(bhit.alloc((size_t)n));
  KAT_DO(bo.put(o, m)); KAT_DO(bd.put(d, m)); KAT_DO(bmn.put(mn, m)); KAT_DO(bmx.put(mx, m));
  launch_kat_aabb(c->stream, n, bo.p, bd.p, bmn.p, bmx.p, bhit.p, bdist.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  KAT_DO(bhit.get(hit, (size_t)n));
  return bdist.get(dist, (size_t)n);
}

int dr_kat_node_planes(dr_context* c, int n, const uint32_t* w, const float* a, const float* b, float* t_mix, float* t_cvt) {
  KAT_PRE(n);
  DevBuf<uint32_t> bw;
  DevBuf<float> ba, bb, b1, b2;
  KAT_DO(bw.alloc((size_t)n)); KAT_DO(ba.alloc((size_t)n)); KAT_DO(bb.alloc((size_t)n)); KAT_DO(b1.alloc((size_t)n * 4)); KAT_DO(b2.alloc((size_t)n * 4));
  KAT_DO(bw.put(w, (size_t)n)); KAT_DO(ba.put(a, (size_t)n)); KAT_DO(bb.put(b, (size_t)n));
  launch_kat_node_planes(c->stream, n, bw.p, ba.p, bb.p, b1.p, b2.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  KAT_DO(b1.get(t_mix, (size_t)n * 4));
  return b2.get(t_cvt, (size_t)n * 4);
}

int dr_kat_tri(dr_context* c, int n, const float* o, const float* d, const float* v0, const float* v1, const float* v2, float* t) {
  KAT_PRE(n);
  DevBuf<float> bo, bd, b0, b1, b2, bt;
  size_t m = (size_t)n * 3;
  KAT_DO(bo.alloc(m)); KAT_DO(bd.alloc(m)); KAT_DO(b0.alloc(m)); KAT_DO(b1.alloc(m)); KAT_DO(b2.alloc(m)); KAT_DO(bt.alloc((size_t)n));
  KAT_DO(bo.put(o, m)); KAT_DO(bd.put(d, m)); KAT_DO(b0.put(v0, m)); KAT_DO(b1.put(v1, m)); KAT_DO(b2.put(v2, m));
  launch_kat_tri(c->stream, n, bo.p, bd.p, b0.p, b1.p, b2.p, bt.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return bt.get(t, (size_t)n);
}

int dr_kat_sphere(dr_context* c, int n, const float* o, const float* d, const float* centre, const float* radius, float* t) {
  KAT_PRE(n);
  DevBuf<float> bo, bd, bc, br, bt;
  size_t m = (size_t)n * 3;
  KAT_DO(bo.alloc(m)); KAT_DO(bd.alloc(m)); KAT_DO(bc.alloc(m)); KAT_DO(br.alloc((size_t)n)); KAT_DO(bt.alloc((size_t)n));
  KAT_DO(bo.put(o, m)); KAT_DO(bd.put(d, m)); KAT_DO(bc.put(centre, m)); KAT_DO(br.put(radius, (size_t)n));
  launch_kat_sphere(c->stream, n, bo.p, bd.p, bc.p, br.p, bt.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return bt.get(t, (size_t)n);
}

int dr_kat_optics(dr_context* c, int n, const float* v, const float* nrm, const float* eta, float* refl, float* refr, float* schlick) {
  KAT_PRE(n);
  DevBuf<float> bv, bn, be, b1, b2, b3;
  size_t m = (size_t)n * 3;
  KAT_DO(bv.alloc(m)); KAT_DO(bn.alloc(m)); KAT_DO(be.alloc((size_t)n)); KAT_DO(b1.alloc(m)); KAT_DO(b2.alloc(m)); KAT_DO(b3.alloc((size_t)n));
  KAT_DO(bv.put(v, m)); KAT_DO(bn.put(nrm, m)); KAT_DO(be.put(eta, (size_t)n));
  launch_kat_optics(c->stream, n, bv.p, bn.p, be.p, b1.p, b2.p, b3.p);
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
  launch_kat_normal(c->stream, P, n, bs.p, bo.p, bd.p, bt.p, bn.p, bc.p);
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
  P.wide = c->wide; P.wide_bytes = (uint32_t)c->wide_bytes; P.wide_pmax = c->wide_pmax; P.wide_mu = c->wide_mu;
  launch_kat_hit(c->stream, P, traversal_of(c), n, bo.p, bd.p, bt.p, bs.p, bv.p);
  HIP_TRY(hipStreamSynchronize(c->stream));
  KAT_DO(bt.get(t, (size_t)n));
  if (visits) KAT_DO(bv.get(visits, (size_t)n));
  std::vector<int32_t> slots((size_t)n);
  KAT_DO(bs.get(slots.data(), (size_t)n));
  for (int i = 0; i < n; i++) idx[i] = slots[(size_t)i] >= 0 ? c->slot_to_orig[(size_t)slots[(size_t)i]] : 0;   // hit() returns index 0 on a miss (K:507)
  return DR_OK;
}

}  // extern "C"
