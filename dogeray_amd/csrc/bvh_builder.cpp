// Host BVH builder: produces, bit for bit, the tree the reference builds
// (kernel.cu build_bvh K:1864-1909, bvhr K:1745-1861, split K:1678-1717, sorto K:1626-1674,
// calculateSD K:1560-1623, pairsort K:1534-1557, arraybound K:383-406, bounding_box K:335-364,
// build_links K:1720-1742) -- same node numbering, same float bounds, same links -- but
// without the reference's per-node new/delete and recursion-only structure:
//
//   * one index array is partitioned in place (a node's primitives are a span of it);
//   * node numbers are computed, not allocated: a subtree with c primitives whose root already
//     has a number uses exactly 2c-2 more, so when a node's children are (k, k+1) the left
//     subtree's descendants start at k+2 and the right subtree's at k+2*count(left).  That makes
//     subtrees independent, and the top of the tree fans out over std::threads;
//   * per-primitive boxes are computed once instead of once per level.
//
// Numerics that must not change (they decide the split axis and therefore the tree):
// float sum / mean, (float)((double)sd + (double)diff*(double)diff) per element in list order,
// sqrtf(sd/len), ties between axes resolved to the highest axis, sort key = (vertex-0
// coordinate, object index) ascending.  Compile with -ffp-contract=off.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <utility>

#include "scene_host.hpp"

namespace dr {
namespace {

struct PrimBox {
  float mn[3], mx[3];
  bool written;   // bounding_box() writes nothing for types other than 0 and 2 (K:339-357)
};

struct Builder {
  HostScene& sc;
  std::vector<PrimBox> boxes;   // N + 1
  std::vector<int> index;       // the `under` lists, partitioned in place
  int max_par_depth = 0;

  explicit Builder(HostScene& s) : sc(s) {}

  void prim_boxes() {
    const size_t n = sc.objects.size();
    boxes.resize(n);
    for (size_t i = 0; i < n; i++) {
      const dr_object& o = sc.objects[i];
      PrimBox& b = boxes[i];
      b.written = true;
      if (o.type == 0) {
        for (int k = 0; k < 3; k++) { b.mn[k] = o.pos[k] - o.dim[0]; b.mx[k] = o.pos[k] + o.dim[0]; }
      } else if (o.type == 2) {
        for (int k = 0; k < 3; k++) {
          // K:353-354: float min/max of the three vertices, -/+ 0.01 in double, narrowed to float
          b.mn[k] = (float)((double)fminf(o.pos[k], fminf(o.dim[k], o.rot[k])) - 0.01);
          b.mx[k] = (float)((double)fmaxf(o.pos[k], fmaxf(o.dim[k], o.rot[k])) + 0.01);
        }
      } else {
        b.written = false;
      }
    }
  }

  // arraybound over list[0..len): running fmin/fmax in list order; an unwritten box repeats
  // the previous primitive's (or -1 if it is the first), as the reference's temporaries do.
  void bound(const int* list, int len, float mn[3], float mx[3]) const {
    float tmn[3] = {-1, -1, -1}, tmx[3] = {-1, -1, -1};
    for (int g = 0; g < len; g++) {
      const PrimBox& b = boxes[(size_t)list[g]];
      if (b.written) { memcpy(tmn, b.mn, sizeof(tmn)); memcpy(tmx, b.mx, sizeof(tmx)); }
      if (g == 0) { memcpy(mn, tmn, sizeof(tmn)); memcpy(mx, tmx, sizeof(tmx)); }
      else for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], tmn[k]); mx[k] = fmaxf(mx[k], tmx[k]); }
    }
  }

  float deviation(const int* list, int len, int axis) const {
    float sum = 0.0f;
    for (int i = 0; i < len; i++) sum += sc.objects[(size_t)list[i]].pos[axis];
    float mean = sum / len;
    float sd = 0.0f;
    for (int i = 0; i < len; i++) {
      float diff = sc.objects[(size_t)list[i]].pos[axis] - mean;
      double d = (double)diff;
      sd = (float)((double)sd + d * d);   // pow(float,int) promotes to double; d*d is exact
    }
    return sqrtf(sd / len);
  }

  // Sort list[0..len) by (pos[axis], index); returns nothing, the halves are list[0..len/2) and the rest.
  void split(int* list, int len, std::vector<std::pair<float, int>>& scratch) const {
    float dx = deviation(list, len, 0), dy = deviation(list, len, 1), dz = deviation(list, len, 2);
    float mx = fmaxf(dx, fmaxf(dy, dz));
    int axis = 0;
    if (mx == dx) axis = 0;
    if (mx == dy) axis = 1;
    if (mx == dz) axis = 2;
    scratch.resize((size_t)len);
    for (int i = 0; i < len; i++) scratch[(size_t)i] = std::make_pair(sc.objects[(size_t)list[i]].pos[axis], list[i]);
    std::sort(scratch.begin(), scratch.end());
    for (int i = 0; i < len; i++) list[i] = scratch[(size_t)i].second;
  }

  void make_child(int id, const int* list, int count) {
    dr_bvh_node& n = sc.bvh[(size_t)id];
    n.active = 1;
    n.count = count;
    n.end = 0;
    if (count == 1) { n.under = list[0]; n.end = 1; }
    bound(list, count, n.min, n.max);
  }

  // `node` is numbered and bounded; its primitives are list[0..count); its children get the
  // numbers first_free and first_free + 1.
  void subtree(int node, int* list, int count, int first_free, int depth) {
    std::vector<std::pair<float, int>> scratch;
    // explicit stack instead of recursion for the sequential part
    struct Item { int node; int* list; int count; int first_free; int depth; };
    std::vector<Item> stack;
    std::vector<std::thread> spawned;
    stack.push_back({node, list, count, first_free, depth});
    while (!stack.empty()) {
      Item it = stack.back();
      stack.pop_back();
      if (it.count < 2) continue;   // leaf
      split(it.list, it.count, scratch);
      int p1 = it.count / 2, p2 = it.count - p1;
      int an = it.first_free, bn = it.first_free + 1;
      make_child(an, it.list, p1);
      make_child(bn, it.list + p1, p2);
      sc.bvh[(size_t)it.node].children[0] = an;
      sc.bvh[(size_t)it.node].children[1] = bn;
      int left_free = it.first_free + 2;
      int right_free = it.first_free + 2 * p1;   // left subtree uses 2*p1 - 2 further numbers
      if (it.depth < max_par_depth && it.count > 4096) {
        int* l = it.list; int d = it.depth + 1;
        spawned.emplace_back([this, an, l, p1, left_free, d]() { subtree(an, l, p1, left_free, d); });
        stack.push_back({bn, it.list + p1, p2, right_free, it.depth + 1});
      } else {
        stack.push_back({bn, it.list + p1, p2, right_free, it.depth + 1});
        stack.push_back({an, it.list, p1, left_free, it.depth + 1});
      }
    }
    for (std::thread& t : spawned) t.join();
  }

  // build_links without recursion: hit = first child, miss = whatever follows the subtree.
  void links() {
    struct Item { int self; int next_right; };
    std::vector<Item> stack;
    stack.push_back({0, -1});
    while (!stack.empty()) {
      Item it = stack.back();
      stack.pop_back();
      dr_bvh_node& n = sc.bvh[(size_t)it.self];
      if (!n.end) {
        n.hit_node = n.children[0];
        n.miss_node = it.next_right;
        stack.push_back({n.children[1], it.next_right});
        stack.push_back({n.children[0], n.children[1]});
      } else {
        n.hit_node = it.next_right;
        n.miss_node = it.next_right;
      }
    }
  }
};

}  // namespace

int build_bvh(HostScene& sc, int nthreads) {
  const int N = sc.n;
  if (N < 2) {
    set_error("BVH build needs at least 2 objects (the reference recurses without bound below that, K:1756)");
    return DR_ERR_SCENE;
  }
  if ((int)sc.objects.size() != N + 1) { set_error("scene object array has the wrong size"); return DR_ERR_INVALID; }
  if (nthreads <= 0) nthreads = usable_threads();
  if (nthreads < 1) nthreads = 1;

  Builder b(sc);
  while ((1 << b.max_par_depth) < nthreads && b.max_par_depth < 6) b.max_par_depth++;
  if (nthreads == 1) b.max_par_depth = 0;

  dr_bvh_node blank;
  memset(&blank, 0, sizeof(blank));
  sc.bvh.assign((size_t)(N + 1) * 2, blank);       // bvhnum = nanum * 2 (K:2073), all inactive (K:1873)
  b.prim_boxes();
  b.index.resize((size_t)N + 1);
  for (int i = 0; i <= N; i++) b.index[(size_t)i] = i;

  dr_bvh_node& root = sc.bvh[0];
  root.active = 1;
  root.count = N;                                   // nanum - 1 (K:1895)
  root.end = 0;
  b.bound(b.index.data(), N + 1, root.min, root.max);   // the root box also covers slot N (K:1899)
  b.subtree(0, b.index.data(), N, 1, 0);            // actualbvhnum starts at 1 (K:1874)
  b.links();
  sc.bvh_used = 2 * N - 1;
  return DR_OK;
}

}  // namespace dr
