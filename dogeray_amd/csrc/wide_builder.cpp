// Traversal structure of the wide walk: a 4-way tree of 64-byte records over the REFERENCE's leaves.
//
// What hit() (kernel.cu K:468-512) returns is fixed by the leaves alone: the primitive with the smallest t
// among the leaves whose own box the ray enters (an ancestor's box encloses the leaf's, and the slab test is
// monotone in the box, so "every ancestor passes" follows from "the leaf passes"), ties going to the leaf the
// reference's walk reaches first (= the lower slot).  The internal levels only decide which leaves are
// LOOKED AT; any tree whose internal boxes enclose the leaf boxes below them looks at a superset of the
// leaves the reference accepts, and the leaf's own test uses the reference's exact box and arithmetic.
// So the internal levels are free, and the reference's median split on vertex 0 (K:1678-1717) is a poor
// traversal structure (266 node visits per ray on the city scene).  This file builds a better one:
//
//   1. a binary tree over the leaf boxes, either by binned surface-area heuristic (default) or by copying
//      the reference's topology (option wide_tree = 0, for comparison); wide_tree = 2 (default): the SAH tree over the small triangles' OWN bounds
//      instead of their padded leaf boxes, with a per-ray margin in the kernel that keeps the set of accepted hits the reference's (step 0 below);
//   2. collapsed into nodes with up to four children (largest child opened first);
//   3. laid out as 64-byte records, children of a node contiguous, child boxes quantised to 8 bits on a
//      per-node grid {origin, power-of-two scale; the record stores scale * 2^24} and rounded OUTWARD; every decoded plane is checked here
//      with the same fmaf the kernel uses, so enclosure is a verified fact, not an error estimate.
//
// Leaves keep the reference's exact box and {v0, e1, e2}; slot = rank of the leaf in the reference's
// child-0-first order.  Scenes this cannot represent (non-finite leaf boxes, more than 2^24 records, depth
// beyond the kernel's stack) make build_wide() return false and the context falls back to the threaded walk.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <new>
#include <system_error>
#include <thread>

#include "linearise.hpp"

namespace dr {
namespace {

struct Box {
  float mn[3], mx[3];
  void clear() { for (int a = 0; a < 3; a++) { mn[a] = INFINITY; mx[a] = -INFINITY; } }
  void grow(const Box& b) { for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], b.mn[a]); mx[a] = fmaxf(mx[a], b.mx[a]); } }
  float half_area() const {
    float d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
  }
};

struct BNode {          // binary tree: children < 0 encode a leaf (~slot)
  Box box;
  int child[2];
};

#ifndef DR_WIDE_BINS
#define DR_WIDE_BINS 16
#endif
constexpr int BINS = DR_WIDE_BINS;
constexpr int MAX_BINARY_DEPTH = 2 * WIDE_MAX_DEPTH;     // a 2-levels-per-node collapse then fits the kernel's stack

struct SahBuilder {
  const std::vector<Box>& leaf;      // by slot
  std::vector<float> cen;            // 3 per slot
  std::vector<int> idx;              // slots, partitioned in place
  std::vector<BNode> nodes;          // node k's subtree with c leaves occupies k .. k + c - 2 (internal nodes only)
  int par_depth = 0;

  explicit SahBuilder(const std::vector<Box>& l) : leaf(l) {}

  // internal node `id` covers idx[first, first + count), count >= 2; `levels` = binary levels still allowed below it
  void subtree(int id, int first, int count, int levels, int depth) {
    struct Item { int id, first, count, levels, depth; };
    std::vector<Item> stack;
    std::vector<std::thread> spawned;
    // whatever happens below (an allocation that throws, a thread that cannot be created), the threads already running are joined
    // before this frame goes away: a joinable std::thread destroyed during unwinding would end the process
    struct Joiner { std::vector<std::thread>& v; ~Joiner() { for (std::thread& t : v) if (t.joinable()) t.join(); } } joiner{spawned};
    stack.push_back({id, first, count, levels, depth});
    while (!stack.empty()) {
      const Item it = stack.back();
      stack.pop_back();
      int* list = idx.data() + it.first;
      Box nb, cb;
      nb.clear(); cb.clear();
      for (int i = 0; i < it.count; i++) {
        nb.grow(leaf[(size_t)list[i]]);
        const float* c = &cen[(size_t)list[i] * 3];
        for (int a = 0; a < 3; a++) { cb.mn[a] = fminf(cb.mn[a], c[a]); cb.mx[a] = fmaxf(cb.mx[a], c[a]); }
      }
      nodes[(size_t)it.id].box = nb;
      int nl = -1;
      const long long cap = it.levels - 1 >= 31 ? (1ll << 31) : (1ll << (it.levels - 1));   // most leaves one child may hold
      if (it.count > 2) {
        // binned SAH over the three axes
        float best_cost = INFINITY; int best_axis = -1, best_split = 0;
        float scale[3];
        for (int a = 0; a < 3; a++) {
          const float ext = cb.mx[a] - cb.mn[a];
          scale[a] = ext > 0 ? (float)BINS * (1.0f - 1e-6f) / ext : 0.0f;
          if (!std::isfinite(scale[a])) scale[a] = 0.0f;
        }
        Box bb[3][BINS]; int bc[3][BINS];
        for (int a = 0; a < 3; a++) for (int b = 0; b < BINS; b++) { bb[a][b].clear(); bc[a][b] = 0; }
        for (int i = 0; i < it.count; i++) {
          const int s = list[i];
          const float* c = &cen[(size_t)s * 3];
          for (int a = 0; a < 3; a++) {
            int b = (int)((c[a] - cb.mn[a]) * scale[a]);
            b = b < 0 ? 0 : (b > BINS - 1 ? BINS - 1 : b);
            bb[a][b].grow(leaf[(size_t)s]); bc[a][b]++;
          }
        }
        for (int a = 0; a < 3; a++) {
          if (!(scale[a] > 0)) continue;
          float right_area[BINS]; int right_n[BINS];
          Box acc; acc.clear(); int n = 0;
          for (int b = BINS - 1; b > 0; b--) { acc.grow(bb[a][b]); n += bc[a][b]; right_area[b] = n ? acc.half_area() : 0.0f; right_n[b] = n; }
          acc.clear(); n = 0;
          for (int b = 1; b < BINS; b++) {
            acc.grow(bb[a][b - 1]); n += bc[a][b - 1];
            if (n == 0 || right_n[b] == 0) continue;
            if (n > cap || right_n[b] > cap) continue;                     // would not fit the depth budget
            const float cost = acc.half_area() * (float)n + right_area[b] * (float)right_n[b];
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; }
          }
        }
        if (best_axis >= 0) {
          const int a = best_axis;
          int* mid = std::partition(list, list + it.count, [&](int s) {
            int b = (int)((cen[(size_t)s * 3 + a] - cb.mn[a]) * scale[a]);
            b = b < 0 ? 0 : (b > BINS - 1 ? BINS - 1 : b);
            return b < best_split;
          });
          nl = (int)(mid - list);
        }
      }
      if (nl <= 0 || nl >= it.count) {
        // no usable SAH split (coincident centroids, two leaves, depth budget): median on the widest centroid axis
        int a = 0;
        for (int k = 1; k < 3; k++) if (cb.mx[k] - cb.mn[k] > cb.mx[a] - cb.mn[a]) a = k;
        nl = it.count / 2;
        std::nth_element(list, list + nl, list + it.count, [&](int x, int y) {
          const float cx = cen[(size_t)x * 3 + a], cy = cen[(size_t)y * 3 + a];
          return cx < cy || (cx == cy && x < y);
        });
      }
      const int nr = it.count - nl;
      BNode& me = nodes[(size_t)it.id];
      // left subtree's internal nodes: id + 1 .. id + nl - 1; right subtree's: id + nl ..
      const int lid = it.id + 1, rid = it.id + nl;
      me.child[0] = nl == 1 ? ~list[0] : lid;
      me.child[1] = nr == 1 ? ~list[nl] : rid;
      const bool fork = it.depth < par_depth && it.count > 8192;
      if (nr > 1) stack.push_back({rid, it.first + nl, nr, it.levels - 1, it.depth + 1});
      if (nl > 1) {
        if (fork) {
          const int f = it.first, lv = it.levels - 1, d = it.depth + 1;
          try {
            // (an exception must not leave a thread function: it is noted, and build() rethrows after everybody has been joined)
            spawned.emplace_back([this, lid, f, nl, lv, d]() { try { subtree(lid, f, nl, lv, d); } catch (...) { failed.store(true); } });
          } catch (const std::system_error&) {
            stack.push_back({lid, f, nl, lv, d});                          // no more threads to be had: this subtree on our own stack
          }
        } else {
          stack.push_back({lid, it.first, nl, it.levels - 1, it.depth + 1});
        }
      }
    }
    for (std::thread& t : spawned) t.join();
  }
  std::atomic<bool> failed{false};
};

struct WNode {
  int child[4];       // >= 0: binary node that becomes a wide node; < 0: leaf ~slot
  int n;
};

// children of the wide node rooted at binary node b
void collapse(const std::vector<BNode>& bn, const std::vector<Box>& leaf, int b, bool fixed_levels, WNode& out) {
  out.n = 2;
  out.child[0] = bn[(size_t)b].child[0];
  out.child[1] = bn[(size_t)b].child[1];
  if (fixed_levels) {
    // exactly two binary levels per wide node: depth(wide) <= ceil(depth(binary) / 2)
    int c[4], n = 0;
    for (int k = 0; k < 2; k++) {
      const int ch = out.child[k];
      if (ch >= 0) { c[n++] = bn[(size_t)ch].child[0]; c[n++] = bn[(size_t)ch].child[1]; }
      else c[n++] = ch;
    }
    out.n = n;
    memcpy(out.child, c, sizeof(int) * (size_t)n);
    return;
  }
  while (out.n < 4) {
    int pick = -1; float area = -1.0f;
    for (int k = 0; k < out.n; k++) {
      if (out.child[k] < 0) continue;
      const float a = bn[(size_t)out.child[k]].box.half_area();
      if (a > area) { area = a; pick = k; }
    }
    if (pick < 0) break;
    const int ch = out.child[pick];
    for (int k = out.n; k > pick + 1; k--) out.child[k] = out.child[k - 1];
    out.child[pick] = bn[(size_t)ch].child[0];
    out.child[pick + 1] = bn[(size_t)ch].child[1];
    out.n++;
  }
  (void)leaf;
}

int wide_depth(const std::vector<BNode>& bn, const std::vector<Box>& leaf, bool fixed_levels) {
  struct Item { int b, d; };
  std::vector<Item> st;
  st.push_back({0, 1});
  int deepest = 0;
  WNode w;
  while (!st.empty()) {
    const Item it = st.back();
    st.pop_back();
    if (it.d > deepest) deepest = it.d;
    collapse(bn, leaf, it.b, fixed_levels, w);
    for (int k = 0; k < w.n; k++) if (w.child[k] >= 0) st.push_back({w.child[k], it.d + 1});
  }
  return deepest;
}

inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline bool finite_box(const float* mn, const float* mx) {
  for (int a = 0; a < 3; a++) if (!std::isfinite(mn[a]) || !std::isfinite(mx[a])) return false;
  return true;
}

// Quantise the boxes of a node's children.  On return lo/hi hold the bytes and every decoded plane encloses:
// fmaf(q_lo, scale, origin) <= box.mn and fmaf(q_hi, scale, origin) >= box.mx, evaluated with the kernel's own fmaf.
bool quantise(const Box* cb, int n, float origin[3], float scale[3], uint8_t lo[4][3], uint8_t hi[4][3]) {
  for (int a = 0; a < 3; a++) {
    float o = INFINITY, top = -INFINITY;
    for (int k = 0; k < n; k++) { o = fminf(o, cb[k].mn[a]); top = fmaxf(top, cb[k].mx[a]); }
    const float ext = top - o;
    if (!std::isfinite(o) || !std::isfinite(ext)) return false;
    int e;
    (void)frexpf(ext > 0 ? ext / 255.0f : 0.0f, &e);       // ext / 255 = m * 2^e, m in [0.5, 1): 2^e >= ext / 255
    if (!(ext > 0)) e = -60;
    if (e < -60) e = -60;                                   // scale in [2^-60, 2^36]: scale * (1 / direction) * 2^24 stays exact in the kernel (wide_node_test)
    for (;; e++) {
      if (e > 36) return false;
      const float s = ldexpf(1.0f, e);
      bool ok = true;
      for (int k = 0; k < n && ok; k++) {
        long ql = (long)floorf((cb[k].mn[a] - o) / s);
        if (ql > 255) ql = 255;
        if (ql < 0) ql = 0;
        while (ql > 0 && !(fmaf((float)ql, s, o) <= cb[k].mn[a])) ql--;
        if (!(fmaf((float)ql, s, o) <= cb[k].mn[a])) { ok = false; break; }
        long qh = (long)ceilf((cb[k].mx[a] - o) / s);
        if (qh < 0) qh = 0;
        while (qh <= 255 && !(fmaf((float)qh, s, o) >= cb[k].mx[a])) qh++;
        if (qh > 255) { ok = false; break; }
        lo[k][a] = (uint8_t)ql; hi[k][a] = (uint8_t)qh;
      }
      if (ok) { origin[a] = o; scale[a] = s; break; }
    }
  }
  return true;
}

}  // namespace

bool build_wide(const HostScene& sc, const std::vector<int>& leaf_node_of_slot, const std::vector<DevPrim>& prims, int tree_mode,
                int nthreads, WideImage& out) {
  out.rec.clear(); out.depth = 0; out.nodes = 0; out.leaves = 0;
  const int N = (int)leaf_node_of_slot.size();
  if (N < 2) return false;
  std::vector<Box> leaf((size_t)N);
  for (int s = 0; s < N; s++) {
    const dr_bvh_node& b = sc.bvh[(size_t)leaf_node_of_slot[(size_t)s]];
    if (!finite_box(b.min, b.max)) return false;
    memcpy(leaf[(size_t)s].mn, b.min, 12); memcpy(leaf[(size_t)s].mx, b.max, 12);
  }
  // ---- 0. the boxes the tree is built over and culls with.  tree_mode 2: a small triangle enters with its OWN bounds (the record's v0, v0 + e1, v0 + e2,
  // rounded outward) instead of the reference's leaf box, which is those bounds padded by 0.01 (K:353-354) -- 2.9 times the footprint of a 0.028-wide triangle --;
  // the rays make up for it with a margin that is a multiple of E = max |e1| |e2| over these triangles (device_core.hpp wide_ray_margin).
  // "Small": |e1| |e2| at or below a cut chosen per scene (below); the tree keeps the reference's boxes altogether unless at least half of the leaves qualify.
  // Other primitives keep the reference's box.
  std::vector<Box> cull(leaf);
  out.mu = WideMu{0, 0, 0}; out.own_bounds = 0;
  if (tree_mode == 2) {
    double vmax_all = 0;
    for (int s = 0; s < N; s++) {
      const DevPrim& p = prims[(size_t)s];
      if (p.type == 2) vmax_all = std::max(vmax_all, std::sqrt((double)p.v0[0] * p.v0[0] + (double)p.v0[1] * p.v0[1] + (double)p.v0[2] * p.v0[2]));
    }
    // The cut: the rays' margin is a multiple of the LARGEST |e1| |e2| that gets in, so one medium triangle taxes every small one.  Of the values of |e1| |e2|
    // in the scene (a sample of them), take the one that maximises (triangles at or below it) x (padding it leaves them: 0.01 minus a unit ray's margin).
    const double s_ref = 2.0 * vmax_all + 1.0, l_cut = std::max(1.0, vmax_all);
    double e_cut = 0;
    {
      std::vector<double> es;
      es.reserve((size_t)N);
      for (int s = 0; s < N; s++) {
        const DevPrim& p = prims[(size_t)s];
        if (p.type != 2) continue;
        const double n1 = std::sqrt((double)p.e1x * p.e1x + (double)p.e1y * p.e1y + (double)p.e1z * p.e1z), n2 = std::sqrt((double)p.e2x * p.e2x + (double)p.e2y * p.e2y + (double)p.e2z * p.e2z);
        if (n1 * n2 == n1 * n2 && n1 + n2 <= l_cut) es.push_back(n1 * n2);
      }
      std::sort(es.begin(), es.end());
      double best = 0;
      for (int q = 1; q <= 64 && !es.empty(); q++) {
        const size_t k = std::min(es.size() - 1, es.size() * (size_t)q / 64);
        const double e = es[k], left = 0.01 - 0.021 * e * s_ref;
        const size_t count = (size_t)(std::upper_bound(es.begin(), es.end(), e) - es.begin());
        if (left > 0 && (double)count * left > best) { best = (double)count * left; e_cut = e; }
      }
    }
    double e_s = 0, l_s = 0, v_s = 0;
    int chosen = 0;
    if (vmax_all < 0x1p30) {
      for (int s = 0; s < N; s++) {
        const DevPrim& p = prims[(size_t)s];
        if (p.type != 2) continue;
        const double v0[3] = {p.v0[0], p.v0[1], p.v0[2]}, e1[3] = {p.e1x, p.e1y, p.e1z}, e2[3] = {p.e2x, p.e2y, p.e2z};
        const double n1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), n2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
        if (!(n1 * n2 <= e_cut && n1 + n2 <= l_cut)) continue;      // (a NaN edge stays with the reference's box)
        Box t;
        bool inside = true;
        for (int a = 0; a < 3; a++) {
          const double lo = std::min(v0[a], std::min(v0[a] + e1[a], v0[a] + e2[a])), hi = std::max(v0[a], std::max(v0[a] + e1[a], v0[a] + e2[a]));      // (exact: sums of two floats)
          float fl = (float)lo, fh = (float)hi;
          if ((double)fl > lo) fl = nextafterf(fl, -INFINITY);
          if ((double)fh < hi) fh = nextafterf(fh, INFINITY);
          t.mn[a] = fl; t.mx[a] = fh;
          // only ever SHRINK the reference's box (v0 + e1 is v1 up to a rounding of the edge: it cannot leave the padding)
          if (!(fl >= leaf[(size_t)s].mn[a] && fh <= leaf[(size_t)s].mx[a])) inside = false;
        }
        if (!inside) continue;
        cull[(size_t)s] = t;
        chosen++;
        e_s = std::max(e_s, n1 * n2); l_s = std::max(l_s, n1 + n2);
        v_s = std::max(v_s, std::sqrt(v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2]));
      }
    }
    if (chosen * 2 < N) {
      // mostly large primitives: the margin every ray would carry on top of THEIR (padded) boxes costs more records than the few own bounds save
      // (3 892-triangle bolter2: +1 % time) -- the tree over the reference's boxes, no margin
      cull = leaf;
      chosen = 0;
    }
    if (chosen > 0) {
      // (rounded up into floats; e > 0 is what switches the rays' margin on)
      auto up = [](double x) { float f = (float)x; if ((double)f < x) f = nextafterf(f, INFINITY); return f; };
      out.mu = WideMu{std::max(up(e_s), 0x1p-100f), up(l_s), up(v_s)};
      out.own_bounds = chosen;
    }
  }

  // ---- 1. binary tree
  std::vector<BNode> bn;
  if (tree_mode == 0) {
    // the reference's topology (K:1745-1861); its internal boxes are recomputed as unions of the leaf boxes
    std::vector<int> slot_of(sc.bvh.size(), -1);
    for (int s = 0; s < N; s++) slot_of[(size_t)leaf_node_of_slot[(size_t)s]] = s;
    std::vector<int> id_of(sc.bvh.size(), -1);
    struct Item { int ref; bool expanded; };
    std::vector<Item> st;
    std::vector<int> order;                          // internal reference nodes, pre-order
    st.push_back({0, false});
    while (!st.empty()) {
      Item it = st.back(); st.pop_back();
      const dr_bvh_node& b = sc.bvh[(size_t)it.ref];
      if (b.end) continue;
      id_of[(size_t)it.ref] = (int)order.size();
      order.push_back(it.ref);
      st.push_back({b.children[1], false});
      st.push_back({b.children[0], false});
    }
    bn.resize(order.size());
    for (size_t k = order.size(); k-- > 0;) {        // reverse pre-order: children before parents
      const dr_bvh_node& b = sc.bvh[(size_t)order[k]];
      BNode& me = bn[k];
      me.box.clear();
      for (int c = 0; c < 2; c++) {
        const int ch = b.children[c];
        if (sc.bvh[(size_t)ch].end) { me.child[c] = ~slot_of[(size_t)ch]; me.box.grow(cull[(size_t)slot_of[(size_t)ch]]); }
        else { me.child[c] = id_of[(size_t)ch]; me.box.grow(bn[(size_t)id_of[(size_t)ch]].box); }
      }
    }
    // the reference tree is a median split: its depth is ceil(log2 N) <= 26
  } else {
    SahBuilder sb(cull);
    sb.cen.resize((size_t)N * 3);
    sb.idx.resize((size_t)N);
    for (int s = 0; s < N; s++) {
      sb.idx[(size_t)s] = s;
      for (int a = 0; a < 3; a++) sb.cen[(size_t)s * 3 + a] = 0.5f * cull[(size_t)s].mn[a] + 0.5f * cull[(size_t)s].mx[a];
    }
    sb.nodes.resize((size_t)N - 1);
    if (nthreads <= 0) nthreads = usable_threads();
    while ((1 << sb.par_depth) < nthreads && sb.par_depth < 6) sb.par_depth++;
    if (nthreads <= 1) sb.par_depth = 0;
    sb.subtree(0, 0, N, MAX_BINARY_DEPTH, 0);
    if (sb.failed.load()) throw std::bad_alloc();      // a worker ran out of memory: dr_context_upload_scene reports it
    bn.swap(sb.nodes);
  }

  // ---- 2. collapse; keep the area-guided collapse if it fits the kernel's stack, else two levels per node
  bool fixed_levels = false;
  int depth = wide_depth(bn, cull, false);
  if (depth > WIDE_MAX_DEPTH) {
    fixed_levels = true;
    depth = wide_depth(bn, cull, true);
    if (depth > WIDE_MAX_DEPTH) return false;
  }

  // ---- 3. records: root at 0; a node's children are contiguous; a child subtree follows its siblings' block
  std::vector<DevUnit>& rec = out.rec;
  rec.reserve(((size_t)N + (size_t)N / 2 + 16) * WIDE_UNITS);
  rec.resize(WIDE_UNITS);
  struct Item { int b; size_t at; };
  std::vector<Item> st;
  st.push_back({0, 0});
  int n_nodes = 0, n_leaves = 0;
  float pmax = 0.0f;
  WNode w;
  while (!st.empty()) {
    const Item it = st.back();
    st.pop_back();
    collapse(bn, cull, it.b, fixed_levels, w);
    const size_t base = rec.size() / WIDE_UNITS;
    if (base + (size_t)w.n > ((size_t)1 << WIDE_INDEX_BITS)) return false;
    rec.resize(rec.size() + (size_t)w.n * WIDE_UNITS);
    Box cb[4];
    uint32_t leafmask = 0;
    for (int k = 0; k < w.n; k++) {
      if (w.child[k] < 0) { cb[k] = cull[(size_t)~w.child[k]]; leafmask |= 1u << k; }
      else cb[k] = bn[(size_t)w.child[k]].box;
    }
    float origin[3], scale[3];
    uint8_t lo[4][3], hi[4][3];
    memset(lo, 0, sizeof(lo)); memset(hi, 0, sizeof(hi));
    if (!quantise(cb, w.n, origin, scale, lo, hi)) return false;
    for (int a = 0; a < 3; a++) pmax = fmaxf(pmax, fmaxf(fabsf(origin[a]), fabsf(fmaf(255.0f, scale[a], origin[a]))));
    uint32_t words[WIDE_UNITS * 4];
    memset(words, 0, sizeof(words));
    // (device_layout.h: the node is the record's first three units)
    uint32_t s24[3];
    for (int a = 0; a < 3; a++) {
      words[a] = fbits(origin[a]);
      s24[a] = fbits(ldexpf(scale[a], 24));      // the record holds scale * 2^24 (device_core.hpp wide_node_test)
      if (s24[a] & 0xffffu) return false;        // (a power of two: nothing below the upper half)
    }
    words[3] = (uint32_t)base | (((1u << w.n) - 1u) << 24) | (leafmask << 28);
    for (int a = 0; a < 3; a++) {
      uint32_t wl = 0, wh = 0;
      for (int k = 0; k < w.n; k++) { wl |= (uint32_t)lo[k][a] << (8 * k); wh |= (uint32_t)hi[k][a] << (8 * k); }
      for (int k = w.n; k < 4; k++) wl |= 255u << (8 * k);      // unused child: inverted box (and cleared valid bit)
      words[4 + a] = wl; words[7 + a] = wh;
    }
    words[10] = (s24[0] >> 16) | (s24[1] & 0xffff0000u);
    words[11] = s24[2];
    memcpy(&rec[it.at * WIDE_UNITS], words, sizeof(words));
    n_nodes++;
    for (int k = w.n - 1; k >= 0; k--) {
      if (w.child[k] >= 0) { st.push_back({w.child[k], base + (size_t)k}); continue; }
      const int slot = ~w.child[k];
      const DevPrim& p = prims[(size_t)slot];
      const int kind = p.type == 0 ? WALK_KIND_SPHERE : (p.type == 2 ? WALK_KIND_TRIANGLE : WALK_KIND_NONE);
      DevUnit* u = &rec[(base + (size_t)k) * WIDE_UNITS];
      const int32_t info = slot | (kind << WALK_SLOT_BITS);
      memcpy(u[0].f, leaf[(size_t)slot].mn, 12); memcpy(&u[0].f[3], &info, 4);
      memcpy(u[1].f, leaf[(size_t)slot].mx, 12); u[1].f[3] = p.v0[0];
      u[2].f[0] = p.v0[1]; u[2].f[1] = p.v0[2]; u[2].f[2] = p.e1x; u[2].f[3] = p.e1y;
      u[3].f[0] = p.e1z; u[3].f[1] = p.e2x; u[3].f[2] = p.e2y; u[3].f[3] = p.e2z;
      n_leaves++;
    }
  }
  if (n_leaves != N) return false;
  out.depth = depth; out.nodes = n_nodes; out.leaves = n_leaves; out.pmax = pmax;
  return true;
}

}  // namespace dr
