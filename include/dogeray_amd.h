/*
 * dogeray_amd.h -- C ABI of the MI355X-native DOGERAY render path (libdogeray_amd.so).
 *
 * The reference (PhilipPragerUrbina/DOGERAY, raygpu/kernel.cu, cited as K:<line>) has no
 * plugin/FFI layer.  The seam this library sits behind is the free function
 *
 *     cudaError_t CudaStarter(int3* outputr, bvh* nbvhtree, singleobject* allobjects,
 *                             cudaTextureObject_t* texarray, int divisor);     K:138, K:2562-2669
 *
 * plus the host functions that feed it (getnum/read K:1113-1530, getppm* and readtextures
 * K:1915-2018, build_bvh K:1864-1909) and the ~14 globals it reads implicitly
 * (K:29-30,109,119-132).  Everything implicit there is an explicit argument here.
 *
 * Conventions
 *   - plain C types only; every function returns DR_OK (0) or a negative dr_status;
 *     dr_last_error() returns the message of the calling thread's last failure.
 *   - a dr_scene lives on the host; a dr_context owns one GPU and the resident scene.
 *   - one host thread per context at a time; calls are synchronous unless they say otherwise.
 *   - there is NO CPU fallback: device entry points fail with DR_ERR_DEVICE without a GPU.
 */
#ifndef DOGERAY_AMD_H
#define DOGERAY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR_ABI_VERSION 2

typedef enum dr_status {
  DR_OK = 0,
  DR_ERR_INVALID = -1, /* bad argument / call order                                   */
  DR_ERR_IO = -2,      /* file cannot be opened (reference: message box, K:1162,1524) */
  DR_ERR_PARSE = -3,   /* stof/stoi would have thrown in the reference                */
  DR_ERR_SCENE = -4,   /* scene the reference cannot build (fewer than 2 objects)     */
  DR_ERR_DEVICE = -5,  /* HIP error, or no GPU                                        */
  DR_ERR_NOMEM = -6
} dr_status;

typedef struct dr_scene dr_scene;     /* host: parsed .rts + textures + BVH */
typedef struct dr_context dr_context; /* device: one GPU, one resident scene */

const char* dr_last_error(void);
int dr_abi_version(void);

/* ------------------------------------------------------------------ host-side types ---- */

/* struct singleobject (K:48-74), field for field; bools widened to int32. */
typedef struct dr_object {
  int32_t type;      /* 0 sphere, 2 triangle (K:440-447) */
  float pos[3];      /* vertex 0 / sphere centre */
  float rot[3];      /* vertex 2 */
  float norm[3];     /* face normal, z == -20 means absent (K:55,750) */
  float n1[3], n2[3], n3[3];
  float t1[3], t2[3], t3[3];
  int32_t smooth, tex, mat;
  float dim[3];      /* vertex 1, or radius in dim[0] */
  float col[3];
  int32_t texnum, rtexnum;
  float addional[3]; /* [1] = roughness / IOR, [0] = diffuse mode (K:827,852,917) */
} dr_object;

/* struct bvh (K:79-96), field for field. */
typedef struct dr_bvh_node {
  int32_t active;
  int32_t children[2];
  int32_t count;
  int32_t hit_node, miss_node;
  int32_t under;
  float min[3], max[3];
  int32_t end;
} dr_bvh_node;

/* The globals the '*' settings line fills (K:1223-1299; defaults K:29-30,109,123-132). */
typedef struct dr_settings {
  float campos[3], look[3];
  float aperture, focus_dist;
  int32_t fov, max_depth, spp;
  float background;
  int32_t backtex;
  int32_t width, height;
} dr_settings;

/* ------------------------------------------------------------------ scene ingest -------- */

/* getnum + getppmnum/getppmpaths + read + readtextures' file loading (K:2055-2082).
 * texture_dir: directory scanned for entries whose path contains "ppm"/"PPM" (the reference
 * scans the process cwd, K:1981); NULL = current directory, "" = no textures.  Entries are
 * taken in sorted name order. */
int dr_scene_load(const char* rts_path, const char* texture_dir, dr_scene** out);
void dr_scene_free(dr_scene* s);

/* The same scene from arrays the caller already holds -- what the reference's main() has after
 * read()/build_bvh() (allobjects K:2061, nbvhtree K:2075, nanum/bvhnum K:1518,2073) and what it
 * hands to CudaStarter.  objects: n_objects + 1 entries (the reference allocates one more than it
 * fills).  bvh: bvhnum = 2 * (n_objects + 1) nodes from the caller's own build_bvh, or NULL to
 * build later with dr_scene_build_bvh.  Everything is copied.  Textures are added in index order
 * (dr_object.texnum / rtexnum and settings.backtex index this list, as texarray[] K:2077-2082). */
int dr_scene_create_from_arrays(const dr_object* objects, int n_objects, const dr_settings* settings,
                                const dr_bvh_node* bvh, int bvhnum, dr_scene** out);
int dr_scene_add_texture(dr_scene* s, const uint8_t* rgba, int width, int height, const char* name);

int dr_scene_num_objects(const dr_scene* s);            /* N = object lines (objnum - 1)   */
int dr_scene_get_objects(const dr_scene* s, dr_object* out /* N + 1 entries */);
int dr_scene_get_settings(const dr_scene* s, dr_settings* out);
int dr_scene_set_settings(dr_scene* s, const dr_settings* in);
int dr_scene_num_textures(const dr_scene* s);
int dr_scene_texture_info(const dr_scene* s, int i, int* width, int* height);
int dr_scene_texture_data(const dr_scene* s, int i, uint8_t* rgba /* w*h*4 */);

/* build_bvh (K:1864-1909): same tree, same node numbering, same float bounds.
 * nthreads <= 0: use all hardware threads. */
int dr_scene_build_bvh(dr_scene* s, int nthreads);
int dr_scene_bvh_size(const dr_scene* s);                /* bvhnum = 2 * (N + 1), K:2073   */
int dr_scene_bvh_used(const dr_scene* s);                /* actualbvhnum = 2N - 1          */
int dr_scene_get_bvh(const dr_scene* s, dr_bvh_node* out /* dr_scene_bvh_size entries */);

/* .rtsb sidecar: binary image of a loaded scene -- objects as parsed ('r' fields frozen at
 * the values this load drew), settings, decoded textures and, if built, the BVH -- so that a second start-up
 * skips getnum/read/build_bvh (K:2055-2094: 330 MB of text at 1M triangles).  The file carries the record sizes
 * of this ABI and a checksum; anything that does not match is DR_ERR_PARSE, never a partly filled scene. */
int dr_scene_save_binary(const dr_scene* s, const char* rtsb_path);
int dr_scene_load_binary(const char* rtsb_path, dr_scene** out);

/* ------------------------------------------------------------------ device -------------- */

int dr_device_count(void);
int dr_context_create(int device_ordinal, dr_context** out);
void dr_context_destroy(dr_context* c);

/* Uploads nodes, primitives, shading records and textures ONCE (the reference re-uploads all
 * of them on every CudaStarter call, K:2604-2629).  The BVH must have been built. */
int dr_context_upload_scene(dr_context* c, const dr_scene* s);

/* Framebuffer partition for multi-GPU: this context renders only the 8-pixel-wide block
 * columns bx with bx % mod == rem; other pixels stay 0.  Default (1, 0) = everything. */
int dr_context_set_stripe(dr_context* c, int mod, int rem);

/* Traversal used by the megakernel.  All return the same closest hit as hit() K:468-512.
 *   DR_TRAVERSAL_THREADED  the reference's order: hit/miss links, child 0 first (its visit counters are the
 *                          reference's: dr_stats.node_visits / prim_tests equal the oracle's V / L)
 *   DR_TRAVERSAL_ORDERED   near child first with a short per-lane stack, ties resolved to the
 *                          leaf the reference order would have reached first
 *   DR_TRAVERSAL_WIDE      (default) a 4-way tree with 8-bit child boxes over the reference's own leaves, nearest
 *                          entered child first, per-lane stack in LDS; the leaves keep the reference's exact boxes
 *                          and tie order.  A scene it cannot represent (non-finite boxes, > 2^24 records) is walked
 *                          THREADED; dr_context_get_option("traversal") tells which one launches use.         */
enum { DR_TRAVERSAL_THREADED = 0, DR_TRAVERSAL_ORDERED = 1, DR_TRAVERSAL_WIDE = 2 };
int dr_context_set_traversal(dr_context* c, int mode);

/* Tuning knobs of the render kernels; none of them changes a pixel.
 *   "kernel"        DR_KERNEL_PERSISTENT (default): waves are pools of 64 path slots that refill
 *                   from a tile queue; DR_KERNEL_TILE: one wave per 8x8 tile, the reference's launch shape
 *   "batch_frames"  most frames one launch of dr_render_accumulate covers (persistent kernel), default 32
 *   "feedback"      1 (default): tiles are started most-expensive-first using the previous launch's costs; the order of a view
 *                   is recomputed after its first two launches and then after every "feedback_every"-th (8); a view that differs
 *                   from the last one only in its settings (a moving camera) starts from the last view's order
 *                   ("order_follows_camera", default 1)
 *   "occupancy"     waves per SIMD.  Persistent kernel: 6 (default: six for the wide walk's lean build of long launches, five for
 *                   every other build), 5 or 4; tile kernel: 4 or 6
 *   "schedule"      persistent kernel: 0 (default, tuned) = shade / refill once fewer than 32 lanes walk, leaf steps once 20 lanes stand at
 *                   a leaf, two steps per loop iteration; 1 = 32 / 8 / one step; 2 = 48 / leaves tested on the spot / one step
 *   "xcd_regions"   1 (default): one tile queue per XCD, each an image band, with stealing; 0: one queue
 *   "heavy_factor"  with feedback: tiles that cost more than this many times the mean start first (most expensive
 *                   first), all others keep their natural order (default 1: the above-average tiles; 0: no tile is
 *                   reordered, -1: every tile by cost)
 *   "coop_steps"    work sharing (the tile queue is empty, or the wave holds a part of a split tile).  Wide walk: a ray older than
 *                   this many steps (default 2, 0 = off) hands the oldest word of its stack -- a subtree -- to a lane that has no
 *                   pixel, up to "coop_rounds" (2) times per loop iteration; all lanes of one ray keep the best hit in one LDS
 *                   word and the owner shades when every piece is done.  The kernel build that contains this is used for
 *                   launches with fewer than "coop_tiles_per_wave" (32) tiles per wave: short launches, whose tail shows; they
 *                   use one tile queue ("short_one_queue", default 1).
 *   "split_parts"   launches of ONE frame: the tiles whose longest pixel took "split_steps" (400) node steps in the previous
 *                   frame -- as many of them as give "split_waves" (12) per cent of the waves a part to start with -- are handed
 *                   out in this many parts (4; 1 = whole, 2, 8); the wave holds until those pixels are done and its other lanes
 *                   help with their rays from the first step
 *                   Threaded walk ("coop_steps" again): once the queue is empty, a ray older than that is finished by all 64
 *                   lanes breadth-first, in waves with at most "coop_lanes" (8) lanes walking
 *   "pipe_streams"  render streams the present pipeline alternates between (2, default, to 4); "pipe_group": most frames one launch of the pipeline
 *                   covers (1 to 16, default 8; see dr_pipeline_submit); "pipe_lean" 1: pipelined launches of ONE frame run the lean build on eight queues
 *                   instead of the work-sharing build (measured slower, default 0; groups are then not formed)
 *   "reserve_cus"   persistent kernel: workgroups are launched for this many CUs fewer than the device has (0 = all; dr_group ranks may leave
 *                   room for the gather's copy / RCCL kernels beside the next batch's rendering)
 *   "wave_log"      1: short launches record begin / queue empty / end of every wave (dr_stats_wave_log)
 *   "wide_tree"     tree under the wide walk, read at dr_context_upload_scene: 2 (default) binned surface-area heuristic, small
 *                   triangles entered with their own bounds instead of the reference's leaf box (those bounds padded by 0.01, K:353-354)
 *                   and every ray carrying the margin that keeps the accepted hits the reference's; 1 the same tree
 *                   over the reference's leaf boxes; 0 the reference's own topology (K:1745-1861) collapsed 4-way
 * The environment variable DOGERAY_OPTIONS="name=value,..." applies the same at context creation. */
enum { DR_KERNEL_TILE = 0, DR_KERNEL_PERSISTENT = 1 };
int dr_context_set_option(dr_context* c, const char* name, int value);
/* Read a knob back (same names), or of the uploaded scene: "tree_depth" (reference tree), "wide_depth" / "wide_nodes"
 * (wide walk; 0 = not representable), "wide_own_bounds" (triangles that entered the wide tree with their own bounds, wide_tree = 2),
 * "traversal" (the one launches really use). */
int dr_context_get_option(const dr_context* c, const char* name, int* value);

/* One CudaStarter call.  settings13 = { cam.xyz, look.xyz, aperture, focus, fov, max_depth,
 * spp, divisor, backtex } exactly as packed at K:2581; W,H = SCREEN_WIDTH/HEIGHT;
 * background = backgroundintensity[0] (K:108,2103); frame_seed replaces clock() at K:1065.
 * out_int3 (host, may be NULL) receives int32[W*H*3], pixel (x,y) at (x*H + y)*3 (K:1006):
 * trunc(mean colour * 255), unclamped; pixels outside the rendered sub-rectangle are 0. */
int dr_render_frame(dr_context* c, const float settings13[13], int W, int H, float background,
                    uint64_t frame_seed, int32_t* out_int3);

/* Progressive accumulation kept on the device (the reference accumulates on the CPU,
 * K:2213-2218).  dr_accum_reset zeroes the W*H*3 int32 accumulator; dr_render_accumulate
 * renders `nframes` frames with seeds frame_seed + k*seed_stride and adds each into it
 * (no host synchronisation between frames; returns after the last one has finished). */
int dr_accum_reset(dr_context* c, int W, int H);
int dr_render_accumulate(dr_context* c, const float settings13[13], int W, int H, float background,
                         uint64_t frame_seed, uint64_t seed_stride, int nframes);
/* The same without the final wait: the launches are queued on the context's stream and the call returns; at most two
 * such batches are in flight (a third call waits for the first).  dr_context_synchronize waits for everything queued
 * and brings dr_stats up to date.  Used to overlap the multi-GPU gather of one batch with the rendering of the next. */
int dr_render_accumulate_async(dr_context* c, const float settings13[13], int W, int H, float background,
                               uint64_t frame_seed, uint64_t seed_stride, int nframes);
int dr_context_synchronize(dr_context* c);

/* ---- the present loop at the batched rate: pipelined single frames.
 * The reference renders ONE frame per CudaStarter call, adds it to `outr` and shows the running mean (K:2154-2224, K:2213-2218,
 * K:2287).  A launch of one frame ends with its slowest pixels while most of the GPU idles.  Here frames are SUBMITTED one by one and every
 * frame still renders into a buffer of its own, but
 *   - frames submitted one after the other for the same view with seeds in arithmetic progression (what a progressive render submits) share
 *     one LAUNCH: a group of up to "pipe_group" (option, default 8; 1 = a launch per frame) frames, rendered by the kernel's batch queue into
 *     that many buffers -- a launch long enough to run the lean build and to hide its tail;
 *   - group k + 1 starts on a second stream while group k drains ("pipe_streams", default 2; streams + 1 groups of buffers rotate);
 *   - a third stream adds the finished frames to the accumulator ONE BY ONE IN TICKET ORDER, so what dr_pipeline_wait(ticket) hands out --
 *     clamp(sum of frames <= ticket / present_divide_by, 0, 255), row-major RGB8 as dr_accum_present -- is exactly the image the reference
 *     shows after that frame.
 *   dr_pipeline_submit   queues one frame (same arguments as dr_render_frame; the frame must match dr_accum_reset's size);
 *                        present_divide_by != 0 also queues the display divide and its download.  Returns at once; *ticket = 0, 1, 2 ...
 *                        The frame's group is launched when it is full, when a frame that does not continue it is submitted, or when it is waited for.
 *   dr_pipeline_wait     blocks until that frame has been added (and presented); out_rgb8 may be NULL.  A caller that keeps two groups' worth of
 *                        frames in flight gets the batched rate; one that waits for every frame before it submits the next gets groups of one
 *                        (the rate of round 3's pipeline).  Tickets of the newest pipe_streams + 1 groups can be waited for.
 *   dr_render_accumulate_pipelined   nframes frames, no presents: dr_render_accumulate's result through the pipeline.
 * Any other call on the context is ordered behind the frames submitted before it (their groups are launched first). */
int dr_pipeline_submit(dr_context* c, const float settings13[13], int W, int H, float background, uint64_t frame_seed,
                       int present_divide_by, uint64_t* ticket);
int dr_pipeline_wait(dr_context* c, uint64_t ticket, uint8_t* out_rgb8);
/* the presented image of a ticket that has been waited for, in place: a pointer into the library's pinned download buffer (W * H * 3 bytes, row-major
 * RGB8), valid until pipe_streams + 1 more frames have been submitted -- for a caller that uploads it to a texture anyway (K:2243-2275) and would
 * rather not copy 6 MB per frame twice */
int dr_pipeline_image(dr_context* c, uint64_t ticket, const uint8_t** rgb8);
int dr_render_accumulate_pipelined(dr_context* c, const float settings13[13], int W, int H, float background,
                                   uint64_t frame_seed, uint64_t seed_stride, int nframes);
/* The HIP stream (hipStream_t) every launch of this context is queued on, for callers that order their own device work
 * (a collective on packed stripes) against it with events. */
int dr_context_stream(dr_context* c, void** hip_stream);
int dr_accum_read(dr_context* c, int32_t* out_int3 /* W*H*3 */);
/* The display divide of K:2287: rgb8[(y*W + x)*3 + ch] = clamp(acc / divide_by, 0, 255). */
int dr_accum_present(dr_context* c, int divide_by, uint8_t* out_rgb8 /* W*H*3, row-major */);
/* Device address of the accumulator (int32[W*H*3]) for device-side gathers (RCCL). */
int dr_accum_device_ptr(dr_context* c, void** dev_ptr, uint64_t* bytes);
/* Multi-GPU gather, device side.  The framebuffer is column-major (K:1006), so one 8-pixel block column is one
 * contiguous run of 8*H*3 int32 and a context's stripe (dr_context_set_stripe: columns rem, rem+mod, ...) packs into
 * [ncols][8*H*3].  dr_accum_pack_stripe queues that copy on the context's stream into one of two library-owned
 * buffers (slot 0/1: pack batch k+1 while batch k is still being sent) and returns its device address and size.
 * dr_accum_unpack_stripes, on the gathering rank, copies the packed stripes of ranks first_rank .. world-1 (rank r's at
 * packed_dev + r * rank_stride_bytes, rank_stride_bytes a multiple of 16) into its own accumulator's columns
 * r, r+world, ..., on hip_stream (NULL: the context's stream).  Ranks render disjoint columns, so the unpack may run
 * beside the gathering rank's own rendering. */
int dr_accum_pack_stripe(dr_context* c, int slot, void** dev_ptr, uint64_t* bytes);
/* allocates pack buffer `slot` for the current accumulator and stripe without packing (so that no allocation falls into a
 * timed or latency-critical gather; dr_group_accum_reset does this for every rank) */
int dr_accum_reserve_pack(dr_context* c, int slot);
int dr_accum_unpack_stripes(dr_context* c, const void* packed_dev, uint64_t rank_stride_bytes, int world, int first_rank,
                            void* hip_stream);

/* ------------------------------------------------------------------ multi-GPU group ----- */
/* One process, one context and one host thread per GPU (the reference is single-device, K:2614-2615).  Rank r of n
 * renders the block columns bx % n == r of every frame (scene replicated); every `gather_every` frames each rank packs
 * its stripe and sends it to rank 0 over RCCL (ncclSend / grouped ncclRecv on a second stream per rank), double-
 * buffered so that the gather of one batch runs beside the rendering of the next.  After the call rank 0's accumulator
 * (dr_group_context(g, 0): dr_accum_read / dr_accum_present) holds the assembled sum of all frames.
 * device_ordinals NULL = 0 .. n-1.  Ranks that share a device, or DOGERAY_GROUP_TRANSPORT=copy, use peer copies. */
typedef struct dr_group dr_group;
int dr_group_create(int n, const int* device_ordinals, dr_group** out);
void dr_group_destroy(dr_group* g);
int dr_group_size(const dr_group* g);
int dr_group_uses_rccl(const dr_group* g);
int dr_group_rccl_ranks(const dr_group* g);   /* ranks of the RCCL communicator (ncclCommCount), 0 when the copy transport is used */
dr_context* dr_group_context(dr_group* g, int rank);
int dr_group_upload_scene(dr_group* g, const dr_scene* s);
int dr_group_accum_reset(dr_group* g, int W, int H);
int dr_group_render_accumulate(dr_group* g, const float settings13[13], int W, int H, float background,
                               uint64_t frame_seed, uint64_t seed_stride, int nframes, int gather_every);

/* Counters and timings since the last dr_stats_reset.  Ray = one hit() call (K:800). */
typedef struct dr_stats {
  uint64_t frames;        /* frames rendered                                            */
  uint64_t launches;      /* render kernel launches (one launch may cover a batch of frames) */
  uint64_t samples;       /* primary samples                                            */
  uint64_t rays;          /* closest-hit queries (counted only when counters are on)    */
  uint64_t node_visits;   /* V: AABB tests performed by the kernel's traversal          */
  uint64_t prim_tests;    /* L                                                          */
  uint64_t shades;        /* S: hits shaded                                             */
  uint64_t texels;        /* T                                                          */
  double kernel_ms;       /* sum of HIP-event durations of the render kernel launches   */
  uint64_t trav_slots;    /* 64 x wave-level traversal iterations: node_visits / trav_slots = SIMD efficiency of the node loop */
  uint64_t ray_slots;     /* 64 x wave-level closest-hit calls:    rays / ray_slots = SIMD efficiency of the bounce loop       */
  uint64_t diag[8];       /* counting build of the persistent kernel: wave cycles, cycles in the shade phase, loop iterations, shade phases,
                             wave-level node steps, wave-level leaf steps (wide walk), lanes shaded, wave lifetime in 100 MHz ticks */
} dr_stats;
int dr_stats_enable_counters(dr_context* c, int on); /* counting build of the kernel; off by default */
int dr_stats_reset(dr_context* c);
int dr_stats_get(dr_context* c, dr_stats* out);

/* Counting build of the persistent kernel (dr_stats_enable_counters): the shade / refill phase's budget since dr_stats_reset, n <= 32 words:
 * [0..15] histogram of the turns a wave's rejection loop ran in a phase (bin 15: 15 or more; bin 0: nobody drew), [16] candidates drawn by all lanes,
 * [17] turns summed over phases, [18] lanes that drew a point in the unit sphere (scatter, K:640-648), [19] lanes that drew a point in the unit disk
 * (new path, K:988-994), [20] retired lanes summed over phases.  With dr_stats.diag (phases, lanes shaded, cycles in phases) this prices the phase. */
int dr_stats_phase_counts(dr_context* c, unsigned long long* out, int n);

/* Timeline of the last SHORT persistent-kernel launch (fewer than coop_tiles_per_wave tiles per wave: one frame, a thin stripe;
 * option "wave_log" = 1 before the launch): sixteen words per wave --
 * begin, first time the wave found the work queue empty (0: never), end, all in 100 MHz ticks of the GPU's
 * real-time counter, and the loop iterations the wave ran after the queue was empty (words 4..15: zero).  Shows where a launch's
 * tail goes (a single frame per launch, K:2154-2224, is mostly tail).  out: 16 * max_waves words. */
int dr_stats_wave_log(dr_context* c, unsigned long long* out, int max_waves, int* n_waves);
/* Node steps each pixel of the last frame cost (the persistent kernel's feedback for its tile order, option
 * "feedback"): pixel (tile, lane) at tile * 64 + lane, tile = block column * ceil(H/8) + block row,
 * lane = (x & 7) * 8 + (y & 7).  *n = words written (0: no feedback recorded yet). */
int dr_stats_pixel_cost(dr_context* c, unsigned* out, size_t capacity, size_t* n);
/* Measurement aid (bench.py `roofline.gather`): rate at which this GPU serves divergent, dependent fetches of 64-byte
 * records from the RESIDENT wide-walk array -- the walk's memory behaviour without its arithmetic.  hot_records
 * restricts the random walk to the first records of the array (0 = all of it). */
int dr_context_probe_gather(dr_context* c, uint32_t hot_records, int iters, double* records_per_s);

/* Measurement aid (tools/exp_trace_rate.py): what the walk would cost as a kernel of its own.  Renders ONE frame with the counting build, which writes every ray
 * it traces into the order a per-bounce wavefront would hold them (bounce by bounce, pixels in tile order); then a TRACE-ONLY kernel walks `frames` copies of that
 * list -- variant 0: one ray per lane, a wave waits for its slowest; 1..7: persistent waves whose lanes refill from the ray list in a few instructions, 128 rays per
 * wave and atomic (occupancy / free lanes before a refill / lanes at a leaf before a leaf step: 6/1/20, 6/8/20, 6/16/20, 8/8/20, 8/8/28, 6/8/28, 8/4/32) -- and is
 * timed (best of three).  *n_rays = rays of the frame; *mismatches = results that differ from the one-ray-per-lane walk (t bits or slot). */
int dr_context_probe_trace(dr_context* c, const float settings13[13], int W, int H, float background, uint64_t frame_seed, int frames, int variant,
                           double* rays_per_s, uint64_t* n_rays, uint64_t* mismatches);

/* ------------------------------------------------------------------ known-answer hooks -- */
/* Run single device functions on caller data (host pointers), for parity tests. */
int dr_kat_rng(dr_context* c, uint64_t seed, int n, double* out);
int dr_kat_aabb(dr_context* c, int n, const float* o, const float* d, const float* mn, const float* mx,
                int32_t* hit, float* dist);
int dr_kat_tri(dr_context* c, int n, const float* o, const float* d, const float* v0, const float* v1,
               const float* v2, float* t);
int dr_kat_sphere(dr_context* c, int n, const float* o, const float* d, const float* centre,
                  const float* radius, float* t);
int dr_kat_optics(dr_context* c, int n, const float* v, const float* nrm, const float* eta, float* refl,
                  float* refr, float* schlick);
/* getnormal K:703-773 for object `object_index[i]` (index in the .rts file) of the resident scene, ray (o, d), hit at t:
 * the normalised, not yet flipped normal and the interpolated texture coordinate (z = 0), 3 floats each */
int dr_kat_normal(dr_context* c, int n, const int32_t* object_index, const float* o, const float* d, const float* t,
                  float* normal, float* texco);
/* The plane arithmetic of the wide walk's node test (device_core.hpp wide_node_test): for word w[i] (four plane bytes) t_mix[4 i + k] =
 * fma(byte_k read as the f16 denormal byte * 2^-24 inside v_fma_mix_f32, a[i] * 2^24, b[i]) and t_cvt[4 i + k] = fma((float)byte_k, a[i], b[i]):
 * the two must agree bit for bit (a[i] * 2^24 must not overflow) -- the check that the instruction keeps f16 denormals */
int dr_kat_node_planes(dr_context* c, int n, const uint32_t* w, const float* a, const float* b, float* t_mix, float* t_cvt);
/* closest hit against the resident scene: t (-1 = miss), ORIGINAL object index and (visits may be NULL)
 * the number of boxes the chosen traversal tested for that ray */
int dr_kat_hit(dr_context* c, int n, const float* o, const float* d, float* t, int32_t* idx, int32_t* visits);

#ifdef __cplusplus
}
#endif
#endif /* DOGERAY_AMD_H */
