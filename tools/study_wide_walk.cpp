// Host emulation of the wide walk (device_core.hpp wide_* functions, same arithmetic and stack discipline) against the
// threaded reference-order walk, on camera rays + two diffuse bounces: checks that hits are identical and prints visits per ray,
// build time and tree statistics.  The folded node test is the kernel's own source (device_core.hpp compiled for the host, as in
// tools/host_kernel.cpp), checked at every node against the plain decode-and-slab test it must cover.  Build from the repo root:
//   /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O2 -ffp-contract=off -mfma -DDR_HOST_BUILD=1 -I tools/host_kernel -I dogeray_amd/csrc -I include -o /tmp/widewalk tools/study_wide_walk.cpp \
//       dogeray_amd/csrc/linearise.cpp dogeray_amd/csrc/wide_builder.cpp dogeray_amd/csrc/rts_reader.cpp \
//       dogeray_amd/csrc/bvh_builder.cpp dogeray_amd/csrc/capi_host.cpp -pthread && /tmp/widewalk scene.rts 20000 [tree_mode]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "linearise.hpp"
#include "device_core.hpp"
using namespace dr;

static bool slab(const float o[3], const float inv[3], const float mn[3], const float mx[3], float& dist) {
  float t0[3], t1[3];
  for (int a = 0; a < 3; a++) { float n = inv[a] < 0 ? mx[a] : mn[a], f = inv[a] < 0 ? mn[a] : mx[a]; t0[a] = (n - o[a]) * inv[a]; t1[a] = (f - o[a]) * inv[a]; }
  float tmin = fmaxf(fmaxf(fmaxf(t0[0], 0.0f), t0[1]), t0[2]), tmax = fminf(fminf(fminf(t1[0], 10000.0f), t1[1]), t1[2]);
  dist = tmin; return tmax > tmin;
}
static float tri(const float o[3], const float d[3], const float* v0, const float* e1, const float* e2) {
  float h[3] = {d[1]*e2[2]-d[2]*e2[1], d[2]*e2[0]-d[0]*e2[2], d[0]*e2[1]-d[1]*e2[0]};
  float a = e1[0]*h[0]+e1[1]*h[1]+e1[2]*h[2]; if (a > -1e-4f && a < 1e-4f) return -1;
  float f = 1/a, s[3] = {o[0]-v0[0], o[1]-v0[1], o[2]-v0[2]};
  float u = f*(s[0]*h[0]+s[1]*h[1]+s[2]*h[2]); if (u < 0 || u > 1) return -1;
  float q[3] = {s[1]*e1[2]-s[2]*e1[1], s[2]*e1[0]-s[0]*e1[2], s[0]*e1[1]-s[1]*e1[0]};
  float v = f*(d[0]*q[0]+d[1]*q[1]+d[2]*q[2]); if (v < 0 || u+v > 1) return -1;
  float t = f*(e2[0]*q[0]+e2[1]*q[1]+e2[2]*q[2]); return t > 1e-4f ? t : -1;
}

struct WideStats { long nodes = 0, leaves = 0, tests = 0, maxsp = 0, cull_nodes = 0, cull_leaves = 0, cull_entry = 0, extra_children = 0, leafkids[5] = {0, 0, 0, 0, 0}; };

static void wide_hit(const std::vector<DevUnit>& rec, float pmax, WideMu mu, const float o[3], const float d[3], const float inv[3], float& best_t, int& best_slot, WideStats& ws) {
  best_t = 10000.0f; best_slot = -1;      // trav_begin
  const WideRay wr = wide_ray(mk(o[0], o[1], o[2]), mk(d[0], d[1], d[2]), mk(inv[0], inv[1], inv[2]), pmax, mu.e, mu.l, mu.v);
  unsigned stack[WIDE_STACK]; int sp = 0; unsigned top = 0;
  float sd[WIDE_STACK + 1][4]; float topd[4] = {0, 0, 0, 0};   // study only: entry distance of every pending child
  unsigned cur = 0;   // index << 1 | leaf
  for (;;) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&rec[(size_t)(cur >> 1) * WIDE_UNITS]);
    const float* f = reinterpret_cast<const float*>(w);
    bool descend = false;
    if (cur & 1) {
      ws.leaves++;
      float dist;
      if (slab(o, inv, f, f + 4, dist) && dist <= best_t) {
        ws.tests++;
        const int info = (int)w[3];
        const float v0[3] = {f[7], f[8], f[9]}, e1[3] = {f[10], f[11], f[12]}, e2[3] = {f[13], f[14], f[15]};
        float t = ((info >> WALK_SLOT_BITS) & 3) == WALK_KIND_TRIANGLE ? tri(o, d, v0, e1, e2) : -1;
        const int slot = info & ((1 << WALK_SLOT_BITS) - 1);
        if (t > 0 && t < 10000.0f && (t < best_t || (t == best_t && (unsigned)slot < (unsigned)best_slot))) { best_t = t; best_slot = slot; }
      }
    } else {
      ws.nodes++;
      const unsigned base = wide_node_first_child(w), valid = wide_node_valid(w), leafmask = wide_node_leaf_mask(w);
      // the kernel's folded test (wide_node_test) and, as a check, the plain decode-and-slab test it must cover
      unsigned mask_plain = 0, key = 0xffffffffu; float dist[4] = {0, 0, 0, 0};
      u32x4 RA, RB, RC; memcpy(&RA, w, 16); memcpy(&RB, w + 4, 16); memcpy(&RC, w + 8, 16);
      unsigned mask = wide_node_test(RA, RB, RC, mk(inv[0], inv[1], inv[2]), wr, best_t, key);      // the kernel's test
      float nearest_plain = INFINITY;
      for (int k = 0; k < 4; k++) {
        float mn[3], mx[3];
        for (int a = 0; a < 3; a++) {
          const float ql = (float)((wide_node_lo(w, a) >> (8 * k)) & 255u), qh = (float)((wide_node_hi(w, a) >> (8 * k)) & 255u);
          const uint32_t sb = wide_node_scale24_bits(w, a); float s24; memcpy(&s24, &sb, 4);
          const float sc = s24 * 0x1p-24f;      // the record holds scale * 2^24
          mn[a] = fmaf(ql, sc, f[a]); mx[a] = fmaf(qh, sc, f[a]);
        }
        float dplain;
        if (slab(o, inv, mn, mx, dplain) && dplain <= best_t) { mask_plain |= 1u << k; if ((valid >> k) & 1u) nearest_plain = fminf(nearest_plain, dplain); }
        dist[k] = dplain;      // (study only: the culling statistics below)
      }
      if ((mask_plain & valid) && (mask & valid)) {      // the nearest key's entry distance may not be later than the nearest plain one
        const uint32_t kb = key & ~3u; float kt; memcpy(&kt, &kb, 4);
        if (kt > nearest_plain) { printf("folded entry distance later than the plain one: %g > %g\n", kt, nearest_plain); exit(3); }
      }
      mask_plain &= valid;
      if (mask_plain & ~mask) { printf("FOLDED TEST NOT CONSERVATIVE: plain %x folded %x\n", mask_plain, mask); exit(3); }
      ws.extra_children += __builtin_popcount(mask & ~mask_plain);
      if (leafmask & valid) ws.leafkids[__builtin_popcount(mask & leafmask)]++;
      if (mask) {
        int near = (int)(key & 3u);
        if (!((mask >> near) & 1u)) near = __builtin_ctz(mask);
        const unsigned rest = mask & ~(1u << near);
        if (rest) { if (top) { if (sp >= WIDE_STACK) { printf("STACK OVERFLOW\n"); exit(2); } memcpy(sd[sp], topd, 16); stack[sp++] = top; } top = (base << 8) | (leafmask << 4) | rest; memcpy(topd, dist, 16); if (sp > ws.maxsp) ws.maxsp = sp; }
        cur = ((base + (unsigned)near) << 1) | ((leafmask >> near) & 1u);
        descend = true;
      }
    }
    if (descend) continue;
    if (!top) { if (sp == 0) break; top = stack[--sp]; memcpy(topd, sd[sp], 16); }
    const int j = __builtin_ctz(top & 15u);
    if (topd[j] > best_t) { if ((top >> (4 + j)) & 1u) ws.cull_leaves++; else ws.cull_nodes++; }
    { bool all = true; for (int k = 0; k < 4; k++) if ((top >> k & 1) && !(topd[k] > best_t)) all = false; if (all) ws.cull_entry++; }
    cur = (((top >> 8) + (unsigned)j) << 1) | ((top >> (4 + j)) & 1u);
    top &= top - 1;
    if (!(top & 15u)) top = 0;
  }
}

// exact (unquantised) box of every record: a leaf's own, a node's = union of its children's -- to price the 8-bit quantisation
static void exact_boxes(const std::vector<DevUnit>& rec, size_t idx, bool leaf, std::vector<float>& bx) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(&rec[idx * WIDE_UNITS]);
  const float* f = reinterpret_cast<const float*>(w);
  float* b = &bx[idx * 6];
  if (leaf) { for (int a = 0; a < 3; a++) { b[a] = f[a]; b[3 + a] = f[4 + a]; } return; }
  const unsigned base = wide_node_first_child(w), valid = wide_node_valid(w), leafmask = wide_node_leaf_mask(w);
  for (int a = 0; a < 3; a++) { b[a] = INFINITY; b[3 + a] = -INFINITY; }
  for (int k = 0; k < 4; k++) if (valid >> k & 1) {
    exact_boxes(rec, base + k, leafmask >> k & 1, bx);
    for (int a = 0; a < 3; a++) { b[a] = fminf(b[a], bx[(base + k) * 6 + a]); b[3 + a] = fmaxf(b[3 + a], bx[(base + k) * 6 + 3 + a]); }
  }
}
static void exact_hit(const std::vector<DevUnit>& rec, const std::vector<float>& bx, const float o[3], const float d[3], const float inv[3], long& nodes, long& leaves) {
  float best_t = 1e7f; unsigned best_slot = ~0u;
  struct E { unsigned idx; bool leaf; };
  std::vector<E> st{{0, false}};
  while (!st.empty()) {
    E e = st.back(); st.pop_back();
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&rec[(size_t)e.idx * WIDE_UNITS]);
    const float* f = reinterpret_cast<const float*>(w);
    float dist;
    if (!(slab(o, inv, &bx[(size_t)e.idx * 6], &bx[(size_t)e.idx * 6 + 3], dist) && dist <= best_t)) continue;      // pruned at pop time too: the ideal
    if (e.leaf) {
      leaves++;
      const int info = (int)w[3];
      const float v0[3] = {f[7], f[8], f[9]}, e1[3] = {f[10], f[11], f[12]}, e2[3] = {f[13], f[14], f[15]};
      float t = ((info >> WALK_SLOT_BITS) & 3) == WALK_KIND_TRIANGLE ? tri(o, d, v0, e1, e2) : -1;
      const unsigned slot = (unsigned)(info & ((1 << WALK_SLOT_BITS) - 1));
      if (t > 0 && t < 10000.0f && (t < best_t || (t == best_t && slot < best_slot))) { best_t = t; best_slot = slot; }
      continue;
    }
    nodes++;
    const unsigned base = wide_node_first_child(w), valid = wide_node_valid(w), leafmask = wide_node_leaf_mask(w);
    struct C { float d; unsigned idx; bool leaf; } c[4]; int n = 0;
    for (int k = 0; k < 4; k++) if (valid >> k & 1) {
      float dk;
      if (slab(o, inv, &bx[(size_t)(base + k) * 6], &bx[(size_t)(base + k) * 6 + 3], dk) && dk <= best_t) c[n++] = {dk, base + (unsigned)k, (bool)(leafmask >> k & 1)};
    }
    std::sort(c, c + n, [](const C& x, const C& y) { return x.d > y.d; });      // farthest first onto the stack
    for (int k = 0; k < n; k++) st.push_back({c[k].idx, c[k].leaf});
  }
}

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  dr_scene sc; sc.host.settings = default_settings();
  auto T0 = std::chrono::steady_clock::now();
  if (read_rts(argv[1], sc.host) != DR_OK || build_bvh(sc.host, 0) != DR_OK) { printf("load failed: %s\n", get_error().c_str()); return 1; }
  auto T1 = std::chrono::steady_clock::now();
  const int mode = argc > 3 ? atoi(argv[3]) : 1;
  DeviceImage img; if (linearise(sc.host, img, mode) != DR_OK) { printf("linearise failed: %s\n", get_error().c_str()); return 1; }
  auto T2 = std::chrono::steady_clock::now();
  printf("load+reference build %.2f s, linearise incl. wide build %.2f s; N = %d, wide: %zu records (%d nodes), depth %d\n",
         std::chrono::duration<double>(T1 - T0).count(), std::chrono::duration<double>(T2 - T1).count(), sc.host.n, img.wide.size() / WIDE_UNITS, img.wide_nodes, img.wide_depth);
  if (img.wide.empty()) { printf("scene not representable\n"); return 1; }
  const HostScene& S = sc.host;
  std::vector<int> slot(S.bvh.size(), -1); { std::vector<int> st{0}; int s = 0; while (!st.empty()) { int n = st.back(); st.pop_back(); const dr_bvh_node& b = S.bvh[n]; if (b.end) slot[n] = s++; else { st.push_back(b.children[1]); st.push_back(b.children[0]); } } }
  const dr_settings& st = S.settings; int W = st.width, H = st.height;
  float from[3] = {st.campos[0], st.campos[1], st.campos[2]}, at[3] = {st.look[0], st.look[1], st.look[2]};
  auto norm = [](float* v) { float l = sqrtf(v[0]*v[0]+v[1]*v[1]+v[2]*v[2]); v[0]/=l; v[1]/=l; v[2]/=l; };
  float w[3] = {from[0]-at[0], from[1]-at[1], from[2]-at[2]}; norm(w);
  float up[3] = {0,1,0}, u[3] = {up[1]*w[2]-up[2]*w[1], up[2]*w[0]-up[0]*w[2], up[0]*w[1]-up[1]*w[0]}; norm(u);
  float v[3] = {w[1]*u[2]-w[2]*u[1], w[2]*u[0]-w[0]*u[2], w[0]*u[1]-w[1]*u[0]};
  float vh = 2*tanf(st.fov * 3.14159265f / 360), vw = vh * W / H;
  std::mt19937 rng2(3); std::uniform_real_distribution<float> U(0, 1);
  int nr = argc > 2 ? atoi(argv[2]) : 20000;
  long bin_int = 0, bin_leaf = 0, bin_tests = 0, rays = 0; WideStats ws;
  std::vector<float> bx(img.wide.size() / WIDE_UNITS * 6); exact_boxes(img.wide, 0, false, bx);
  long ex_nodes = 0, ex_leaves = 0;
  for (int r = 0; r < nr; r++) {
    float o[3], d[3], inv[3];
    float s = U(rng2), t = U(rng2);
    for (int a = 0; a < 3; a++) { o[a] = from[a]; d[a] = (s - 0.5f) * vw * u[a] + (t - 0.5f) * vh * v[a] - w[a]; }
    norm(d);
    for (int bounce = 0; bounce < 3; bounce++) {
      for (int a = 0; a < 3; a++) inv[a] = 1.0f / d[a];
      float best = 1e7f; int bs = -1;
      { int node = 0; while (node >= 0) { const dr_bvh_node& b = S.bvh[node]; float dist; bool h = slab(o, inv, b.min, b.max, dist) && dist < best;
          if (b.end) { bin_leaf++; if (h) { bin_tests++; const DevPrim& p = img.prims[slot[node]]; float e1[3] = {p.e1x, p.e1y, p.e1z}, e2[3] = {p.e2x, p.e2y, p.e2z};
              float tt = p.type == 2 ? tri(o, d, p.v0, e1, e2) : -1; if (tt > 0 && tt < 10000.0f && tt < best) { best = tt; bs = slot[node]; } } node = b.miss_node; }
          else { bin_int++; node = h ? b.hit_node : b.miss_node; } } }
      float wb; int wsl;
      wide_hit(img.wide, img.wide_pmax, img.wide_mu, o, d, inv, wb, wsl, ws);
      exact_hit(img.wide, bx, o, d, inv, ex_nodes, ex_leaves);
      if (wsl != bs || (bs >= 0 && wb != best)) { printf("MISMATCH ray %d bounce %d: wide %d %g vs reference %d %g\n", r, bounce, wsl, wb, bs, best); return 1; }
      rays++;
      if (bs < 0) break;
      for (int a = 0; a < 3; a++) o[a] += best * d[a];
      float nd[3]; do { nd[0] = 2*U(rng2)-1; nd[1] = 2*U(rng2)-1; nd[2] = 2*U(rng2)-1; } while (nd[0]*nd[0]+nd[1]*nd[1]+nd[2]*nd[2] > 1 || nd[0]*nd[0]+nd[1]*nd[1]+nd[2]*nd[2] < 1e-3f);
      if (nd[1] > 0) nd[1] = -nd[1]; norm(nd);
      if (r % 50 == 7) nd[(r / 50) % 3] = (r & 64) ? 0.0f : -0.0f;      // exactly axis-parallel components: 1 / 0 = +-inf in slab()
      if (r % 50 == 9) nd[(r / 50) % 3] = 1e-30f; for (int a = 0; a < 3; a++) { d[a] = nd[a]; o[a] += 1e-3f * nd[a]; }
    }
  }
  printf("%ld rays, hits identical.  reference walk: %.1f internal + %.1f leaf visits, %.2f primitive tests per ray;  wide walk (tree mode %d): %.1f node + %.1f leaf records, %.2f primitive tests per ray, deepest stack %ld; fetched although already farther than the best t when popped: %.2f nodes + %.2f leaves per ray; children entered by the folded test only: %.3f per ray\n",
         rays, (double)bin_int / rays, (double)bin_leaf / rays, (double)bin_tests / rays, mode, (double)ws.nodes / rays, (double)ws.leaves / rays, (double)ws.tests / rays, ws.maxsp, (double)ws.cull_nodes / rays, (double)ws.cull_leaves / rays, (double)ws.extra_children / rays);
  printf("nodes with leaf children, by how many of them the ray enters: 0: %ld  1: %ld  2: %ld  3: %ld  4: %ld\n", ws.leafkids[0], ws.leafkids[1], ws.leafkids[2], ws.leafkids[3], ws.leafkids[4]);
  printf("the same tree with exact child boxes, fully sorted children and pruning at pop time (the ideal this layout approximates): %.1f node + %.1f leaf records per ray\n", (double)ex_nodes / rays, (double)ex_leaves / rays);
  return 0;
}
