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
#include <cmath>
#include <cstring>
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
          auto closest = [&](V3 o, V3 d, Ctr& cc) { return closest_hit_wide<COUNT>(wide, P.wide_pmax, P.wide_mu.e, P.wide_mu.l, P.wide_mu.v, o, d, cc, stack.data()); };
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

// The restated uniform draws of device_core.hpp (Xorwow::uniform_double, uniform_pm1, outside_unit) against the plain expressions of
// curand_uniform_double (CUDA 11.2 curand_kernel.h) and of kernel.cu K:640-648: n random generator outputs plus the corners; returns mismatches.
long long hk_check_uniform(long long n, unsigned long long seed) {
  long long bad = 0;
  auto one = [&bad](uint32_t x, uint32_t y) {
    const uint64_t z = (uint64_t)x ^ ((uint64_t)y << 21);
    const volatile double u_plain = (double)z * 1.1102230246251565e-16 + 5.5511151231257827e-17;
    const volatile double t = u_plain * 2;
    const float f_plain = (float)(t - 1);
    const uint32_t lo = x ^ (y << 21), hi = y >> 11;
    const double zh = ((bits_double(0x45300000u, hi) - 19342813118337666422669312.0) + bits_double(0x43300000u, lo)) + 0.5;
    const double u_new = zh * 0x1p-53;
    const float f_new = (float)__builtin_fma(zh, 0x1p-52, -1.0);
    if (memcmp((const void*)&u_plain, &u_new, 8) != 0 || memcmp(&f_plain, &f_new, 4) != 0) bad++;
  };
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
  auto rnd = [&s]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (long long i = 0; i < n; i++) { const uint64_t r = rnd(); one((uint32_t)r, (uint32_t)(r >> 32)); }
  const uint32_t corners[] = {0u, 1u, 2u, 3u, 0x7ffu, 0x800u, 0x801u, 0x3ffu, 0x400u, 0x401u, 0x7fffffffu, 0x80000000u, 0x80000001u, 0xfffffffeu, 0xffffffffu, 0x001fffffu, 0x00200000u, 0xffe00000u, 0xffdfffffu};
  for (uint32_t a : corners) for (uint32_t b : corners) one(a, b);
  // a texel byte over 255 (byte_over_255) against the division
  for (uint32_t b8 = 0; b8 < 256; b8++) { const volatile float num = (float)b8, den = 255.0f; const float plain = num / den, fast = byte_over_255(b8); if (memcmp(&plain, &fast, 4) != 0) bad++; }
  // the rejection test: floats around 1 (every float within 2^-18 of 1) and a sweep
  auto test = [&bad](float d2) {
    const volatile float l = sqrtf(d2);
    const bool plain = l * l >= 1;
    if (plain != outside_unit(d2)) bad++;
  };
  for (int k = -(1 << 18); k <= (1 << 18); k++) { uint32_t b = 0x3f800000u + (uint32_t)k; float f; memcpy(&f, &b, 4); test(f); }
  for (long long i = 0; i < n / 4; i++) { const uint64_t r = rnd(); test((float)(r >> 40) * (3.0f / 16777216.0f)); }
  return bad;
}

// The merged rejection loop (device_core.hpp rand_points_merged: candidates classified from 32 bits, the accepted one converted exactly from the
// generator's state) against rand_in_unit_sphere / rand_in_unit_disk on n generator states: the same point, bit for bit, and the same state afterwards.
// Returns mismatches; *max_d2_gap (optional) = the largest |d2~ - d2| seen over all candidates, to set beside the bound of the proof (2^-17).
long long hk_check_reject(long long n, unsigned long long seed, double* max_d2_gap) {
  long long bad = 0;
  double gap = 0;
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
  auto rnd = [&s]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (long long i = 0; i < n; i++) {
    Xorwow a; a.init(rnd());
    const int warm = (int)(rnd() % 7); for (int k = 0; k < warm; k++) a.next();
    for (int kind = 2; kind <= 3; kind++) {
      Xorwow plain = a, merged = a;
      const V3 want = kind == 3 ? rand_in_unit_sphere(plain) : rand_in_unit_disk(plain);
      const V3 got = rand_points_merged<false>(merged, kind);
      if (memcmp(&want, &got, sizeof(V3)) != 0 || memcmp(&plain, &merged, sizeof(Xorwow)) != 0) bad++;
    }
    // the gap between the approximate and the exact squared length of one candidate of each kind
    Xorwow g = a;
    const uint32_t o[6] = {g.next(), g.next(), g.next(), g.next(), g.next(), g.next()};
    const float xs = (float)__builtin_fma(z_plus_half_of(o[0], o[1]), 0x1p-52, -1.0), ys = (float)__builtin_fma(z_plus_half_of(o[2], o[3]), 0x1p-52, -1.0), zs = (float)__builtin_fma(z_plus_half_of(o[4], o[5]), 0x1p-52, -1.0);
    const float xd = (float)(z_plus_half_of(o[0], o[1]) * 0x1p-53) * 2 - 1, yd = (float)(z_plus_half_of(o[2], o[3]) * 0x1p-53) * 2 - 1;
    const float xa = reject_coord(o[0], o[1]), ya = reject_coord(o[2], o[3]), za = reject_coord(o[4], o[5]);
    const double g3 = fabs((double)dot(mk(xs, ys, zs), mk(xs, ys, zs)) - (double)__builtin_fmaf(za, za, __builtin_fmaf(ya, ya, xa * xa)));
    const double g2 = fabs((double)dot(mk(xd, yd, 0), mk(xd, yd, 0)) - (double)__builtin_fmaf(ya, ya, xa * xa));
    if (g3 > gap) gap = g3;
    if (g2 > gap) gap = g2;
  }
  if (max_d2_gap) *max_d2_gap = gap;
  return bad;
}

// device_core.hpp wide_ray_margin, for the test of its lemma (tests/test_margin_lemma.py): the position margin of n rays for the scene constants (e, l, v)
void hk_ray_margin(long long n, const float* o, const float* d, float e, float l, float v, float* out) {
  for (long long i = 0; i < n; i++) out[i] = wide_ray_margin(mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]), e, l, v);
}
// hit_tri (device_core.hpp tri_hit) on n ray / triangle pairs: t or -1
void hk_tri_hit(long long n, const float* o, const float* d, const float* v0, const float* e1, const float* e2, float* t) {
  for (long long i = 0; i < n; i++)
    t[i] = tri_hit(mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]), mk(v0[3 * i], v0[3 * i + 1], v0[3 * i + 2]),
                   mk(e1[3 * i], e1[3 * i + 1], e1[3 * i + 2]), mk(e2[3 * i], e2[3 * i + 1], e2[3 * i + 2]));
}

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
int hk_wide_depth(void* hv) { return ((HkScene*)hv)->img.wide.empty() ? 0 : ((HkScene*)hv)->img.wide_depth; }      // nodes on the longest root-to-leaf path of the wide tree

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
  P.wide = img.wide.empty() ? nullptr : img.wide.data(); P.wide_bytes = (uint32_t)(img.wide.size() * sizeof(DevUnit)); P.wide_pmax = img.wide_pmax; P.wide_mu = img.wide_mu;
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
