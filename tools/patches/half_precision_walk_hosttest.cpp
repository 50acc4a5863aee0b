#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include "linearise.hpp"
using namespace dr;
static float h2f(unsigned short h) { const int e = (h >> 10) & 31, m = h & 1023; float v = e == 0 ? std::ldexp((float)m, -24) : (e == 31 ? (m ? NAN : INFINITY) : std::ldexp((float)(m | 1024), e - 25)); return (h & 0x8000) ? -v : v; }
static bool slab(const float o[3], const float inv[3], const float mn[3], const float mx[3], float& dist) {
  float t0[3], t1[3];
  for (int a = 0; a < 3; a++) { float n = inv[a] < 0 ? mx[a] : mn[a], f = inv[a] < 0 ? mn[a] : mx[a]; t0[a] = (n - o[a]) * inv[a]; t1[a] = (f - o[a]) * inv[a]; }
  float tmin = fmaxf(fmaxf(fmaxf(t0[0], 0.0f), t0[1]), t0[2]), tmax = fminf(fminf(fminf(t1[0], 10000.0f), t1[1]), t1[2]);
  dist = tmin; return tmax > tmin;
}
int main(int argc, char** argv) {
  dr_scene sc; sc.host.settings = default_settings();
  if (read_rts(argv[1], sc.host) != DR_OK || build_bvh(sc.host, 0) != DR_OK) return 1;
  DeviceImage img; if (linearise(sc.host, img) != DR_OK) { printf("linearise: %s\n", get_error().c_str()); return 1; }
  const HostScene& S = sc.host;
  printf("walk units %zu, hwalk units %zu\n", img.walk.size(), img.hwalk.size());
  if (img.hwalk.empty()) return 0;
  
  std::mt19937 rng(5); std::uniform_real_distribution<float> U(-1, 1);
  int nrays = argc > 2 ? atoi(argv[2]) : 2000; long se = 0, sh = 0;
  for (int r = 0; r < nrays; r++) {
    float o[3] = {U(rng) * 12, 6 + U(rng) * 6, U(rng) * 12}, d[3] = {U(rng), -fabsf(U(rng)) - 0.05f, U(rng)}, inv[3];
    if (r % 3 == 0) { o[0] = 0; o[1] = -12; o[2] = 18; d[0] = U(rng) * 0.5f; d[1] = 0.6f + U(rng) * 0.3f; d[2] = -1; }
    for (int a = 0; a < 3; a++) inv[a] = 1.0f / d[a];
    std::vector<int> la, lb;
    { int node = 0; while (node >= 0) { const dr_bvh_node& b = S.bvh[node]; float dist; bool h = slab(o, inv, b.min, b.max, dist); se++; if (b.end) { if (h) la.push_back(b.under); node = b.miss_node; } else node = h ? b.hit_node : b.miss_node; } }
    { int node = 0; long guard = 0;
      while (node >= 0) {
        if (++guard > 100000000) { printf("hwalk does not end\n"); return 1; }
        bool leaf = node & 1; size_t u = (size_t)(node >> 1); sh++;
        if (u + (leaf ? 4 : 1) > img.hwalk.size()) { printf("OOB link %d\n", node); return 1; }
        if (leaf) {
          float mn[3], mx[3]; int w0; memcpy(mn, img.hwalk[u].f, 12); memcpy(&w0, &img.hwalk[u].f[3], 4); memcpy(mx, img.hwalk[u + 1].f, 12);
          float dist; bool h = slab(o, inv, mn, mx, dist);
          if (h) lb.push_back(img.slot_to_orig[(size_t)(w0 & ((1 << 26) - 1))]);
          node = node + 7 + ((w0 >> 28) & 1);
        } else {
          unsigned w[4]; memcpy(w, img.hwalk[u].f, 16);
          float mn[3] = {h2f(w[0] & 0xffff), h2f(w[0] >> 16), h2f(w[1] & 0xffff)}, mx[3] = {h2f(w[1] >> 16), h2f(w[2] & 0xffff), h2f(w[2] >> 16)};
          float dist; bool h = slab(o, inv, mn, mx, dist);
          node = h ? node + 2 + (int)(w[3] >> 31) : (int)(w[3] & 0x7fffffffu) - 1;
        }
      } }
    if (la != lb) { printf("ray %d: leaf sequences differ (%zu vs %zu)\n", r, la.size(), lb.size()); return 1; }
  }
  printf("%d rays: hwalk == host BVH (leaf sequences identical); steps exact %ld, half %ld (+%.2f%%)\n", nrays, se, sh, 100.0 * (sh - se) / se);
  return 0;
}
