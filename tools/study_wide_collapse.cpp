// Host study for the next round: node and leaf visits per ray if the reference binary tree is collapsed into 4- or 8-wide
// nodes over the SAME leaves (hits are checked to be identical), against the threaded binary walk.  Build from the repo root:
//   g++ -std=c++17 -O2 -ffp-contract=off -I dogeray_amd/csrc -I include -o /tmp/wide tools/study_wide_collapse.cpp \
//       dogeray_amd/csrc/linearise.cpp dogeray_amd/csrc/rts_reader.cpp dogeray_amd/csrc/bvh_builder.cpp -pthread && /tmp/wide scene.rts 20000
#include <cmath>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include "linearise.hpp"
using namespace dr;
static bool slab(const float o[3], const float inv[3], const float mn[3], const float mx[3], float& dist) {
  float t0[3], t1[3];
  for (int a = 0; a < 3; a++) { float n = inv[a] < 0 ? mx[a] : mn[a], f = inv[a] < 0 ? mn[a] : mx[a]; t0[a] = (n - o[a]) * inv[a]; t1[a] = (f - o[a]) * inv[a]; }
  float tmin = fmaxf(fmaxf(fmaxf(t0[0], 0.0f), t0[1]), t0[2]), tmax = fminf(fminf(fminf(t1[0], 10000.0f), t1[1]), t1[2]);
  dist = tmin; return tmax > tmin;
}
static float tri(const float o[3], const float d[3], const DevPrim& p) {
  float e1[3] = {p.e1x, p.e1y, p.e1z}, e2[3] = {p.e2x, p.e2y, p.e2z};
  float h[3] = {d[1]*e2[2]-d[2]*e2[1], d[2]*e2[0]-d[0]*e2[2], d[0]*e2[1]-d[1]*e2[0]};
  float a = e1[0]*h[0]+e1[1]*h[1]+e1[2]*h[2]; if (fabsf(a) < 1e-4f) return -1;
  float f = 1/a, s[3] = {o[0]-p.v0[0], o[1]-p.v0[1], o[2]-p.v0[2]};
  float u = f*(s[0]*h[0]+s[1]*h[1]+s[2]*h[2]); if (u < 0 || u > 1) return -1;
  float q[3] = {s[1]*e1[2]-s[2]*e1[1], s[2]*e1[0]-s[0]*e1[2], s[0]*e1[1]-s[1]*e1[0]};
  float v = f*(d[0]*q[0]+d[1]*q[1]+d[2]*q[2]); if (v < 0 || u+v > 1) return -1;
  float t = f*(e2[0]*q[0]+e2[1]*q[1]+e2[2]*q[2]); return t > 1e-4f ? t : -1;
}
int main(int argc, char** argv) {
  dr_scene sc; sc.host.settings = default_settings();
  if (read_rts(argv[1], sc.host) != DR_OK || build_bvh(sc.host, 0) != DR_OK) return 1;
  DeviceImage img; if (linearise(sc.host, img) != DR_OK) return 1;
  const HostScene& S = sc.host;
  std::vector<int> slot(S.bvh.size(), -1); { std::vector<int> st{0}; int s = 0; while (!st.empty()) { int n = st.back(); st.pop_back(); const dr_bvh_node& b = S.bvh[n]; if (b.end) slot[n] = s++; else { st.push_back(b.children[1]); st.push_back(b.children[0]); } } }
  auto wide_children = [&](int n, int levels, std::vector<int>& out) {      // descendants `levels` binary levels down (leaves stop early), DFS order
    std::vector<std::pair<int,int>> st{{n, 0}}; out.clear();
    std::vector<std::pair<int,int>> tmp;
    // recursive expansion in order
    std::function<void(int,int)> rec = [&](int m, int l) { const dr_bvh_node& b = S.bvh[m]; if (l == levels || b.end) { out.push_back(m); return; } rec(b.children[0], l + 1); rec(b.children[1], l + 1); };
    const dr_bvh_node& b = S.bvh[n]; rec(b.children[0], 1); rec(b.children[1], 1);
  };
  const dr_settings& st = S.settings; int W = st.width, H = st.height;
  float from[3] = {st.campos[0], st.campos[1], st.campos[2]}, at[3] = {st.look[0], st.look[1], st.look[2]};
  auto norm = [](float* v) { float l = sqrtf(v[0]*v[0]+v[1]*v[1]+v[2]*v[2]); v[0]/=l; v[1]/=l; v[2]/=l; };
  float w[3] = {from[0]-at[0], from[1]-at[1], from[2]-at[2]}; norm(w);
  float up[3] = {0,1,0}, u[3] = {up[1]*w[2]-up[2]*w[1], up[2]*w[0]-up[0]*w[2], up[0]*w[1]-up[1]*w[0]}; norm(u);
  float v[3] = {w[1]*u[2]-w[2]*u[1], w[2]*u[0]-w[0]*u[2], w[0]*u[1]-w[1]*u[0]};
  float vh = 2*tanf(st.fov * 3.14159265f / 360), vw = vh * W / H;
  std::mt19937 rng(3); std::uniform_real_distribution<float> U(0, 1);
  int nr = argc > 2 ? atoi(argv[2]) : 20000;
  for (int levels = 2; levels <= 3; levels++) {
    std::mt19937 rng2(3); long bin_int = 0, bin_leaf = 0, wide_nodes = 0, wide_leaf = 0, wide_tests = 0, rays = 0, maxstack = 0;
    for (int r = 0; r < nr; r++) {
      float o[3], d[3], inv[3];
      float s = U(rng2), t = U(rng2);
      for (int a = 0; a < 3; a++) { o[a] = from[a]; d[a] = (s - 0.5f) * vw * u[a] + (t - 0.5f) * vh * v[a] - w[a]; }
      norm(d);
      for (int bounce = 0; bounce < 3; bounce++) {
        for (int a = 0; a < 3; a++) inv[a] = 1.0f / d[a];
        // binary threaded reference
        float best = 1e7f; int bs = -1;
        { int node = 0; while (node >= 0) { const dr_bvh_node& b = S.bvh[node]; float dist; bool h = slab(o, inv, b.min, b.max, dist) && dist < best;
            if (b.end) { bin_leaf++; if (h) { float tt = tri(o, d, img.prims[slot[node]]); if (tt > 0 && tt < best) { best = tt; bs = slot[node]; } } node = b.miss_node; }
            else { bin_int++; node = h ? b.hit_node : b.miss_node; } } }
        // wide: DFS with an explicit stack of (node) in child order
        float wb = 1e7f; int ws = -1;
        { std::vector<int> stack{0}; std::vector<int> ch;
          // root is an internal binary node: visit it as a wide node
          std::vector<std::pair<int,float>> pend;
          std::function<void(int)> visit = [&](int n) {
            wide_nodes++;
            wide_children(n, levels, ch);
            std::vector<std::pair<int,float>> hits;
            for (int c : ch) { float dist; wide_tests++; if (slab(o, inv, S.bvh[c].min, S.bvh[c].max, dist) && dist < wb) hits.push_back({c, dist}); }
            if ((long)hits.size() > maxstack) maxstack = (long)hits.size();
            for (auto& hc : hits) {
              if (!(hc.second < wb)) continue;                      // pruned since the node was tested
              const dr_bvh_node& b = S.bvh[hc.first];
              if (b.end) { wide_leaf++; float dist; if (slab(o, inv, b.min, b.max, dist) && dist < wb) { float tt = tri(o, d, img.prims[slot[hc.first]]); if (tt > 0 && tt < wb) { wb = tt; ws = slot[hc.first]; } } }
              else visit(hc.first);
            }
          };
          visit(0);
        }
        if (ws != bs || wb != best) { printf("MISMATCH levels %d ray %d: %d %g vs %d %g\n", levels, r, ws, wb, bs, best); return 1; }
        rays++;
        if (bs < 0) break;
        for (int a = 0; a < 3; a++) o[a] += best * d[a];
        float nd[3]; do { nd[0] = 2*U(rng2)-1; nd[1] = 2*U(rng2)-1; nd[2] = 2*U(rng2)-1; } while (nd[0]*nd[0]+nd[1]*nd[1]+nd[2]*nd[2] > 1 || nd[0]*nd[0]+nd[1]*nd[1]+nd[2]*nd[2] < 1e-3f);
        if (nd[1] > 0) nd[1] = -nd[1]; norm(nd); for (int a = 0; a < 3; a++) { d[a] = nd[a]; o[a] += 1e-3f * nd[a]; }
      }
    }
    printf("%d-wide collapse (%d binary levels per node), %ld rays, hits identical: binary walk %.1f internal + %.1f leaf visits per ray; wide walk %.1f node visits (%.1f box tests) + %.1f leaf visits per ray\n",
           1 << levels, levels, rays, (double)bin_int / rays, (double)bin_leaf / rays, (double)wide_nodes / rays, (double)wide_tests / rays, (double)wide_leaf / rays);
  }
  return 0;
}
