// Host scene -> device image (see device_layout.h for the why of each array).
#include <cstring>

#include "linearise.hpp"

namespace dr {

int linearise(const HostScene& sc, DeviceImage& img, int wide_tree_mode) {
  const int N = sc.n;
  if (N < 2) { set_error("scene needs at least 2 objects (K:1756)"); return DR_ERR_SCENE; }
  if (sc.bvh.empty() || sc.bvh_used != 2 * N - 1 || (int)sc.objects.size() < N) { set_error("BVH not built"); return DR_ERR_INVALID; }
  const int used = sc.bvh_used;
  // arrays that came from a caller or from a .rtsb file are not trusted: every link must stay inside the array
  for (const dr_bvh_node& b : sc.bvh) {
    if (!b.active) continue;
    const int lim = (int)sc.bvh.size();
    if (b.hit_node < -1 || b.hit_node >= lim || b.miss_node < -1 || b.miss_node >= lim ||
        (!b.end && (b.children[0] < 0 || b.children[0] >= lim || b.children[1] < 0 || b.children[1] >= lim))) {
      set_error("corrupt BVH (link outside the node array)");
      return DR_ERR_INVALID;
    }
  }
  img.pairs.assign((size_t)(N - 1), DevPair());
  img.prims.assign((size_t)N, DevPrim());
  img.shade.assign((size_t)N, DevShade());
  img.slot_to_orig.assign((size_t)N, -1);

  // pre-order walk (child 0 first = the reference's hit-link order, K:1727-1731)
  std::vector<int> new_id(sc.bvh.size(), -1);     // reference node number -> pre-order number
  std::vector<int> pair_id(sc.bvh.size(), -1);    // reference node number -> internal rank
  std::vector<int> slot_of(sc.bvh.size(), -1);    // reference leaf number -> leaf rank
  std::vector<int> order;
  order.reserve((size_t)used);
  {
    std::vector<int> stack;
    stack.push_back(0);
    int next_pair = 0, next_slot = 0;
    while (!stack.empty()) {
      int n = stack.back();
      stack.pop_back();
      if (n < 0 || n >= (int)sc.bvh.size() || !sc.bvh[(size_t)n].active || new_id[(size_t)n] >= 0) {
        set_error("corrupt BVH (bad child link)");
        return DR_ERR_INVALID;
      }
      new_id[(size_t)n] = (int)order.size();
      order.push_back(n);
      const dr_bvh_node& b = sc.bvh[(size_t)n];
      if (b.end) {
        slot_of[(size_t)n] = next_slot++;
      } else {
        pair_id[(size_t)n] = next_pair++;
        stack.push_back(b.children[1]);
        stack.push_back(b.children[0]);
      }
    }
    if ((int)order.size() != used || next_slot != N || next_pair != N - 1) {
      set_error("corrupt BVH (node count mismatch)");
      return DR_ERR_INVALID;
    }
  }

  // pre-order checks: an internal node's first child is the next node, a leaf's hit link equals
  // its miss link (K:1738-1739) and is the next node too
  for (int k = 0; k < used; k++) {
    const dr_bvh_node& b = sc.bvh[(size_t)order[(size_t)k]];
    if (!b.end && new_id[(size_t)b.children[0]] != k + 1) { set_error("pre-order numbering broken"); return DR_ERR_INVALID; }
    if (b.end) {
      int expect = (k + 1 < used) ? k + 1 : -1;
      int miss = b.miss_node < 0 ? -1 : new_id[(size_t)b.miss_node];
      if (miss != expect) { set_error("leaf link is not the pre-order successor"); return DR_ERR_INVALID; }
    }
  }

  auto child_code = [&](int ref_node) -> int32_t {
    const dr_bvh_node& c = sc.bvh[(size_t)ref_node];
    return c.end ? ~slot_of[(size_t)ref_node] : pair_id[(size_t)ref_node];
  };
  for (int k = 0; k < used; k++) {
    int ref = order[(size_t)k];
    const dr_bvh_node& b = sc.bvh[(size_t)ref];
    if (b.end) continue;
    DevPair& p = img.pairs[(size_t)pair_id[(size_t)ref]];
    const dr_bvh_node& c0 = sc.bvh[(size_t)b.children[0]];
    const dr_bvh_node& c1 = sc.bvh[(size_t)b.children[1]];
    memcpy(p.mn0, c0.min, sizeof(p.mn0)); memcpy(p.mx0, c0.max, sizeof(p.mx0));
    memcpy(p.mn1, c1.min, sizeof(p.mn1)); memcpy(p.mx1, c1.max, sizeof(p.mx1));
    p.c0 = child_code(b.children[0]);
    p.c1 = child_code(b.children[1]);
    p.pad0 = p.pad1 = 0;
  }
  memcpy(img.root_mn, sc.bvh[0].min, sizeof(img.root_mn));
  memcpy(img.root_mx, sc.bvh[0].max, sizeof(img.root_mx));

  for (int k = 0; k < used; k++) {
    int ref = order[(size_t)k];
    const dr_bvh_node& b = sc.bvh[(size_t)ref];
    if (!b.end) continue;
    int slot = slot_of[(size_t)ref];
    int oi = b.under;
    if (oi < 0 || oi >= N) { set_error("leaf refers to an object outside the scene"); return DR_ERR_INVALID; }
    const dr_object& o = sc.objects[(size_t)oi];
    img.slot_to_orig[(size_t)slot] = oi;
    DevPrim& p = img.prims[(size_t)slot];
    memset(&p, 0, sizeof(p));
    p.type = o.type;
    memcpy(p.v0, o.pos, sizeof(p.v0));
    if (o.type == 2) {
      // edge1 = vertex1 - vertex0, edge2 = vertex2 - vertex0 (K:287-288, K:721-722): the same
      // float subtractions the kernel would do per test, done once
      p.e1x = o.dim[0] - o.pos[0]; p.e1y = o.dim[1] - o.pos[1]; p.e1z = o.dim[2] - o.pos[2];
      p.e2x = o.rot[0] - o.pos[0]; p.e2y = o.rot[1] - o.pos[1]; p.e2z = o.rot[2] - o.pos[2];
    } else {
      p.e1x = o.dim[0];   // sphere radius (K:441)
    }
    DevShade& s = img.shade[(size_t)slot];
    memset(&s, 0, sizeof(s));
    memcpy(s.norm, o.norm, sizeof(s.norm));
    memcpy(s.n1, o.n1, sizeof(s.n1)); memcpy(s.n2, o.n2, sizeof(s.n2)); memcpy(s.n3, o.n3, sizeof(s.n3));
    s.t1[0] = o.t1[0]; s.t1[1] = o.t1[1];
    s.t2[0] = o.t2[0]; s.t2[1] = o.t2[1];
    s.t3[0] = o.t3[0]; s.t3[1] = o.t3[1];
    memcpy(s.col, o.col, sizeof(s.col));
    s.add_x = o.addional[0]; s.add_y = o.addional[1];
    s.mat = o.mat;
    const int ntex = (int)sc.textures.size();
    s.texnum = (o.texnum >= 0 && o.texnum < ntex) ? o.texnum : -1;
    s.rtexnum = (o.rtexnum >= 0 && o.rtexnum < ntex) ? o.rtexnum : -1;
    s.flags = (o.smooth ? 1 : 0) | (o.tex ? 2 : 0);
    s.type = o.type;
    s.orig = oi;
  }

  // the walk array: records in pre-order, links carry the target's leaf flag
  {
    std::vector<int64_t> unit_of((size_t)used + 1, 0);
    for (int k = 0; k < used; k++)
      unit_of[(size_t)k + 1] = unit_of[(size_t)k] + (sc.bvh[(size_t)order[(size_t)k]].end ? WALK_UNITS_LEAF : WALK_UNITS_INTERNAL);
    if (unit_of[(size_t)used] + WALK_UNITS_INTERNAL >= ((int64_t)1 << 28)) { set_error("scene too large: the walk array must stay below 4 GiB (32-bit buffer offsets)"); return DR_ERR_SCENE; }
    auto link_to = [&](int pre) -> int32_t {
      if (pre < 0) return -1;
      return (int32_t)((unit_of[(size_t)pre] << 1) | (sc.bvh[(size_t)order[(size_t)pre]].end ? 1 : 0));
    };
    img.walk.assign((size_t)unit_of[(size_t)used] + WALK_UNITS_INTERNAL, DevUnit());
    {
      // terminator: an internal record no ray enters (min > max), both links -1.  The last leaf's successor is
      // "the next record" like every other leaf's, so the kernel's step needs no end-of-walk test.
      DevUnit* t = &img.walk[(size_t)unit_of[(size_t)used]];
      const int32_t end = -1;
      for (int a = 0; a < 3; a++) { t[0].f[a] = 3.0e38f; t[1].f[a] = -3.0e38f; }
      memcpy(&t[0].f[3], &end, 4); memcpy(&t[1].f[3], &end, 4);
    }
    if (N > (1 << WALK_SLOT_BITS)) { set_error("scene too large for the walk array's slot field"); return DR_ERR_SCENE; }
    for (int k = 0; k < used; k++) {
      const dr_bvh_node& b = sc.bvh[(size_t)order[(size_t)k]];
      DevUnit* u = &img.walk[(size_t)unit_of[(size_t)k]];
      if (!b.end) {
        int32_t w0 = link_to(k + 1);
        int32_t w1 = link_to(b.miss_node < 0 ? -1 : new_id[(size_t)b.miss_node]);
        memcpy(u[0].f, b.min, 12); memcpy(&u[0].f[3], &w0, 4);
        memcpy(u[1].f, b.max, 12); memcpy(&u[1].f[3], &w1, 4);
      } else {
        const int slot = slot_of[(size_t)order[(size_t)k]];
        const DevPrim& p = img.prims[(size_t)slot];
        const int kind = p.type == 0 ? WALK_KIND_SPHERE : (p.type == 2 ? WALK_KIND_TRIANGLE : WALK_KIND_NONE);
        const bool last = k + 1 >= used;
        const bool next_leaf = !last && sc.bvh[(size_t)order[(size_t)k + 1]].end;
        int32_t info = slot | (kind << WALK_SLOT_BITS) | ((next_leaf ? 1 : 0) << 28) | ((last ? 1 : 0) << 29);
        memcpy(u[0].f, b.min, 12); memcpy(&u[0].f[3], &info, 4);
        memcpy(u[1].f, b.max, 12); u[1].f[3] = p.v0[0];
        u[2].f[0] = p.v0[1]; u[2].f[1] = p.v0[2]; u[2].f[2] = p.e1x; u[2].f[3] = p.e1y;
        u[3].f[0] = p.e1z; u[3].f[1] = p.e2x; u[3].f[2] = p.e2y; u[3].f[3] = p.e2z;
      }
    }
  }

  // the wide walk: a 4-way tree over the same leaves (wide_builder.cpp); scenes it cannot represent keep the threaded walk
  {
    std::vector<int> leaf_node_of_slot((size_t)N, -1);
    for (int k = 0; k < used; k++) {
      const int ref = order[(size_t)k];
      if (sc.bvh[(size_t)ref].end) leaf_node_of_slot[(size_t)slot_of[(size_t)ref]] = ref;
    }
    WideImage w;
    img.wide.clear(); img.wide_depth = 0; img.wide_nodes = 0;
    if (wide_tree_mode >= 0 && build_wide(sc, leaf_node_of_slot, img.prims, wide_tree_mode, 0, w)) {
      img.wide.swap(w.rec);
      img.wide_depth = w.depth;
      img.wide_nodes = w.nodes;
      img.wide_pmax = w.pmax; img.wide_mu = w.mu; img.wide_own_bounds = w.own_bounds;
    }
  }

  img.tex.clear();
  img.texels.clear();
  for (const HostTexture& t : sc.textures) {
    DevTex d;
    d.offset = (uint32_t)img.texels.size();
    d.w = t.w; d.h = t.h; d.pad = 0;
    size_t px = (size_t)t.w * (size_t)t.h;
    if (img.texels.size() + px > 0xffffffffull) { set_error("textures exceed 4G texels"); return DR_ERR_NOMEM; }
    size_t base = img.texels.size();
    img.texels.resize(base + px);
    memcpy(img.texels.data() + base, t.rgba.data(), px * 4);   // little-endian: R in bits 0-7
    img.tex.push_back(d);
  }
  return DR_OK;
}

}  // namespace dr
