// north_star's CPU baseline: "a host-C++ compile of the same kernel".  This file compiles the product's OWN device functions
// (dogeray_amd/csrc/device_core.hpp with -DDR_HOST_BUILD: slab, tri_hit, sphere_hit, the wide and the threaded walk, surface_normal,
// shade_hit / shade_miss, Xorwow, camera_ray, render_pixel -- unchanged source, only the buffer loads become bounds-checked memcpys and the
// function qualifiers vanish) together with the product's host ingest (reader, BVH builder, lineariser, wide-walk builder) into
// tools/libhostkernel.so, and renders frames with std::threads over 8-pixel block columns, one "lane" at a time.
//
// It is TEST AND BENCH INFRASTRUCTURE: tests/test_host_kernel.py compares its frames with the oracle's pixel for pixel (a check of the
// kernel's arithmetic that needs no GPU) and bench.py reports it as cpu_baseline.kind "same-source" beside the oracle's "port".
// It is not linked into libdogeray_amd.so, which has no CPU path (dr_context_create fails without a GPU).
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#define DR_HOST_BUILD 1
#include "../dogeray_amd/csrc/device_core.hpp"
#include "../dogeray_amd/csrc/linearise.hpp"
#include "../dogeray_amd/csrc/params_host.hpp"
#include "../dogeray_amd/csrc/scene_host.hpp"

using namespace dr;

namespace {
struct HkScene {
  dr_scene* scene = nullptr;
  DeviceImage img;
};
thread_local std::string hk_err;

template <bool COUNT>
void render_columns(const RenderParams& P, int traversal, int first, int step, Ctr& total) {
  std::vector<int> stack((size_t)WIDE_STACK * 64 > (size_t)ORDERED_STACK * 64 ? (size_t)WIDE_STACK * 64 : (size_t)ORDERED_STACK * 64);
  Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
  const WalkRsrc walk = walk_rsrc(P), wide = wide_rsrc(P);
  for (int col = first; col < P.ncols; col += step) {
    const int bx = P.stripe_rem + col * P.stripe_mod;
    for (int by = 0; by < P.gy; by++)
      for (int lane = 0; lane < 64; lane++) {                       // lane l of a tile is pixel (l >> 3, l & 7), as in the kernels
        const int x = bx * 8 + (lane >> 3), y = by * 8 + (lane & 7);
        if (traversal == DR_TRAVERSAL_WIDE && P.wide) {
          auto closest = [&](V3 o, V3 d, Ctr& cc) { return closest_hit_wide<COUNT>(wide, P.wide_pmax, o, d, cc, stack.data()); };
          render_pixel<COUNT>(P, closest, x, y, c);
        } else if (traversal == DR_TRAVERSAL_ORDERED) {
          auto closest = [&](V3 o, V3 d, Ctr& cc) { return closest_hit_ordered<COUNT>(P.pairs, P.prims, o, d, cc, stack.data()); };
          render_pixel<COUNT>(P, closest, x, y, c);
        } else {
          auto closest = [&](V3 o, V3 d, Ctr& cc) { return closest_hit_threaded<COUNT>(walk, o, d, cc); };
          render_pixel<COUNT>(P, closest, x, y, c);
        }
      }
  }
  total = c;
}
}  // namespace

extern "C" {

const char* hk_last_error() { return hk_err.c_str(); }

void* hk_scene_load(const char* rts_path, const char* texdir) {
  HkScene* h = new HkScene();
  if (dr_scene_load(rts_path, texdir ? texdir : "", &h->scene) != DR_OK || dr_scene_build_bvh(h->scene, 0) != DR_OK) {
    hk_err = dr_last_error();
    if (h->scene) dr_scene_free(h->scene);
    delete h;
    return nullptr;
  }
  try {
    if (linearise(h->scene->host, h->img, 1) != DR_OK) throw std::string(dr_last_error());
  } catch (...) {
    hk_err = "scene could not be linearised";
    dr_scene_free(h->scene);
    delete h;
    return nullptr;
  }
  return h;
}

void hk_scene_free(void* hv) {
  HkScene* h = (HkScene*)hv;
  if (!h) return;
  if (h->scene) dr_scene_free(h->scene);
  delete h;
}

int hk_has_wide(void* hv) { return ((HkScene*)hv)->img.wide.empty() ? 0 : 1; }

// One frame (dr_render_frame's arguments): int32[W * H * 3], pixel (x, y) at (x * H + y) * 3, unrendered margins 0; only the block
// columns bx % col_mod == col_rem are rendered (a bounded sample for the bench).  counters: rays, V, L, S, T, samples (6 words).
int hk_render(void* hv, const float* settings13, int W, int H, float background, uint64_t frame_seed, int traversal, int nthreads, int col_mod, int col_rem,
              int32_t* out, uint64_t* counters) {
  HkScene* h = (HkScene*)hv;
  if (!h || !settings13 || !out || col_mod < 1 || col_rem < 0 || col_rem >= col_mod) { hk_err = "bad argument"; return -1; }
  RenderParams P;
  memset(&P, 0, sizeof(P));
  if (const char* why = fill_view_params(settings13, W, H, background, frame_seed, col_mod, col_rem, P)) { hk_err = why; return -1; }
  const DeviceImage& img = h->img;
  if (P.backtex >= (int)img.tex.size()) { hk_err = "backtex refers to a texture that is not loaded"; return -1; }
  P.walk = img.walk.data(); P.walk_bytes = (uint32_t)(img.walk.size() * sizeof(DevUnit));
  P.wide = img.wide.empty() ? nullptr : img.wide.data(); P.wide_bytes = (uint32_t)(img.wide.size() * sizeof(DevUnit)); P.wide_pmax = img.wide_pmax;
  P.pairs = img.pairs.data(); P.prims = img.prims.data(); P.shade = img.shade.data(); P.tex = img.tex.data(); P.texels = img.texels.data();
  P.out = out;
  P.accumulate = 0;
  memset(out, 0, (size_t)W * H * 3 * sizeof(int32_t));
  if (nthreads < 1) nthreads = 1;
  std::vector<Ctr> part((size_t)nthreads);
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++)
    th.emplace_back([&, t] { if (counters) render_columns<true>(P, traversal, t, nthreads, part[(size_t)t]); else render_columns<false>(P, traversal, t, nthreads, part[(size_t)t]); });
  for (std::thread& t : th) t.join();
  if (counters) {
    for (int k = 0; k < 6; k++) counters[k] = 0;
    for (const Ctr& c : part) { counters[0] += c.rays; counters[1] += c.V; counters[2] += c.L; counters[3] += c.S; counters[4] += c.T; counters[5] += c.samples; }
  }
  return 0;
}

}  // extern "C"
