// Device-resident scene layout (HBM), shared by the host lineariser and the HIP kernels.
//
// The reference uploads its host structs as they are: `bvh` (56 B, fields used by the kernel
// at offsets 16..52) and `singleobject` (164 B, the three vertices at offsets 4, 16 and 120),
// every frame (kernel.cu K:2618-2629).  Here the scene is uploaded once, renumbered and split
// by access frequency:
//
//   walk    16-B units, one record per node in DFS pre-order (a subtree is one contiguous range):
//             internal  2 units  {min.xyz, hit link} {max.xyz, miss link}
//             leaf      4 units  {min.xyz, info} {max.xyz, v0.x} {v0.yz, e1.xy} {e1.z, e2.xyz}   (64 B)
//           link = (unit index << 1) | target-is-leaf, -1 = end of the walk.  Knowing from the
//           link that the target is a leaf lets the kernel fetch the box AND the triangle with
//           one round trip instead of two dependent ones.  A leaf needs no miss link: its hit and
//           miss links are equal (K:1738-1739) and in pre-order that successor is the next
//           record, so `info` = slot | kind << 26 | next-is-leaf << 28 | last-node << 29.
//   pairs   64 B   both children of an internal node     (ordered traversal: one fetch per step)
//   prims   48 B   {v0.xyz, e1.x} {e1.yz, e2.xy} {e2.z, type, -, -}   hot intersection data,
//                                                        in leaf (DFS) order: slot = leaf rank
//   shade  128 B   normals, uvs, material                cold, read once per shaded hit
//   texels  4 B    RGBA8, all textures back to back + {offset, w, h} table
//
// Renumbering changes no result: node numbers never leave the BVH, and the slot order is the
// order in which the reference's traversal reaches the leaves, which is what breaks ties.
#pragma once
#include <stdint.h>

namespace dr {

struct DevUnit {       // 16 B; the walk array is made of these (read as float4 on the device)
  float f[4];
};
constexpr int WALK_UNITS_INTERNAL = 2;
constexpr int WALK_UNITS_LEAF = 4;
constexpr int WALK_SLOT_BITS = 26;                    // slots < 2^26 (the walk array itself is limited to 2^28 units)
constexpr int WALK_KIND_SPHERE = 0, WALK_KIND_TRIANGLE = 1, WALK_KIND_NONE = 2;   // object type 0 / 2 / anything else

// Wide walk (wide_builder.cpp): 64-byte records, a 4-way tree over the same leaves.
//   node  {origin.xyz, index of first child | valid mask << 24 | leaf mask << 28} {lo.x, lo.y, lo.z, hi.x} {hi.y, hi.z, scale.x | scale.y << 16, scale.z} {-}
//         each lo/hi word = one byte per child: plane = fmaf(byte, scale, origin); scale is a power of two in [2^-60, 2^36] and is stored times 2^24
//         (the kernel reads a plane byte as the f16 denormal byte * 2^-24, device_core.hpp wide_node_test): x and y as the upper halves of their floats
//         (a power of two has no other bits), z as the float.  A node is its first THREE units: the kernel does not fetch the fourth (a node step costs
//         three 16-byte loads per lane instead of four; every load instruction of a step is ~1 % of the frame, profiles/r4_o_node_three_units.txt)
//   leaf  {min.xyz, slot | kind << 26} {max.xyz, v0.x} {v0.yz, e1.xy} {e1.z, e2.xyz}      (the walk array's leaf record)
// A node's children are contiguous records.  The kernel keeps, per lane, the children of a node that were entered
// but not yet visited as ONE stack word: first child's index << 8 | leaf mask << 4 | pending mask.
// (host-side readers of a node record's words: wide_builder.cpp writes them, tools/study_wide_walk.cpp reads them)
inline uint32_t wide_node_first_child(const uint32_t* w) { return w[3] & 0xffffffu; }
inline uint32_t wide_node_valid(const uint32_t* w) { return (w[3] >> 24) & 15u; }
inline uint32_t wide_node_leaf_mask(const uint32_t* w) { return w[3] >> 28; }
inline uint32_t wide_node_lo(const uint32_t* w, int axis) { return w[4 + axis]; }
inline uint32_t wide_node_hi(const uint32_t* w, int axis) { return w[7 + axis]; }
inline uint32_t wide_node_scale24_bits(const uint32_t* w, int axis) { return axis == 0 ? w[10] << 16 : (axis == 1 ? w[10] & 0xffff0000u : w[11]); }      // bits of scale * 2^24
// wide_tree = 2 (default): small triangles enter the tree with their OWN bounds instead of the reference's leaf box (their bounds padded by 0.01), and every ray
// carries the position margin that makes up for it (device_core.hpp wide_ray; DESIGN.md 4.10 has the proof).  e = 0: every leaf entered with the reference's box.
struct WideMu {
  float e;      // largest |e1| |e2| of the triangles that entered with their own bounds (Euclidean norms)
  float l;      // largest |e1| + |e2| of them
  float v;      // largest |v0| of them
};
constexpr int WIDE_UNITS = 4;
constexpr int WIDE_INDEX_BITS = 24;                   // records < 2^24 (1 GiB of them)
constexpr int WIDE_STACK = 16;                        // stack words per lane kept in LDS (one more lives in a register)
constexpr int WIDE_MAX_DEPTH = WIDE_STACK + 1;        // most nodes on a root-to-leaf path the kernel can walk
constexpr int ORDERED_STACK = 24;                     // ordered traversal: levels of its per-lane stack (median split: depth <= ceil(log2 N))

// Ordered traversal: record k describes the two children of internal node k (pre-order index
// of the internal node among ALL nodes is kept in `DevNode`; pairs are indexed by node id).
struct DevPair {       // 64 B, 64-B aligned
  float mn0[3];
  int32_t c0;          // child 0: >= 0 internal node id, < 0: leaf, slot = ~c0
  float mx0[3];
  int32_t c1;          // child 1, same encoding
  float mn1[3];
  int32_t pad0;
  float mx1[3];
  int32_t pad1;
};

struct DevPrim {       // 48 B, 16-B aligned
  float v0[3];         // triangle: vertex 0; sphere: centre
  float e1x;           // triangle: e1.x;     sphere: radius
  float e1y, e1z, e2x, e2y;
  float e2z;
  int32_t type;        // 0 sphere, 2 triangle, anything else: never hit
  int32_t pad[2];
};

struct DevShade {      // 128 B
  float norm[3];
  float n1[3], n2[3], n3[3];
  float t1[2], t2[2], t3[2];
  float col[3];
  float add_x, add_y;  // addional.x (diffuse mode), addional.y (roughness / IOR)
  int32_t mat;
  int32_t texnum, rtexnum;
  int32_t flags;       // bit 0 smooth, bit 1 tex (checker)
  int32_t type;
  int32_t orig;        // index of the object in the .rts file
  int32_t pad[3];
};

struct DevTex {
  uint32_t offset;     // first texel in the texel array
  int32_t w, h;
  int32_t pad;
};

static_assert(sizeof(DevUnit) == 16, "DevUnit");
static_assert(sizeof(DevPair) == 64, "DevPair");
static_assert(sizeof(DevPrim) == 48, "DevPrim");
static_assert(sizeof(DevShade) == 128, "DevShade");
static_assert(sizeof(DevTex) == 16, "DevTex");

// Everything one launch needs.  The camera basis (kernel.cu K:1016-1052) does not depend on
// the pixel, so it is computed once on the host with the reference's arithmetic.
struct RenderParams {
  const DevUnit* walk;
  uint32_t walk_bytes;            // size of the walk array (buffer descriptor range; < 4 GiB)
  uint32_t pad_;
  const DevUnit* wide;            // wide walk records (null: scene not representable, threaded walk is used)
  uint32_t wide_bytes;
  float wide_pmax;                // largest |plane coordinate| of the wide walk's nodes (margin of the folded node test)
  const DevPair* pairs;
  const DevPrim* prims;
  const DevShade* shade;
  const DevTex* tex;
  const uint32_t* texels;
  int32_t* out;                   // int32[W*H*3], pixel (x, y) at (x*H + y)*3
  unsigned long long* counters;   // 8 words or null: rays, V, L, S, T, samples, trav_slots, ray_slots
  unsigned long long* wave_log;   // null, or 16 words per wave of the persistent kernel: begin, queue empty, end (100 MHz ticks), loop iterations after the queue ran empty
  float from[3], llc[3], hor[3], ver[3], uu[3], vu[3];
  float lens_radius;
  float bgint;
  float scale;                    // (float)(1.0 / spp)
  float spp_f;                    // settings[10]
  double den_w, den_h;            // (double)float(W / divisor), (double)float(H / divisor)
  uint64_t seed;
  int32_t W, H;
  int32_t gx, gy;                 // block grid of the reference launch (K:2636)
  int32_t ncols;                  // block columns this context renders
  int32_t stripe_mod, stripe_rem;
  uint32_t seed_stride;           // blockDim.x * gridDim.x = 8 * gx (K:1065)
  int32_t max_depth;
  int32_t backtex;
  int32_t accumulate;             // 0: store, 1: add into out, 2: atomic add (several frames in one launch)
  int32_t batch;                  // frames rendered by this launch (persistent kernel), >= 1
  uint64_t batch_seed_stride;     // frame f of the batch uses seed + f * batch_seed_stride
  int32_t coop_steps;             // drain phase: a ray older than this many node steps is finished cooperatively (0 = never)
  int32_t coop_lanes;             // ... in waves with at most this many lanes still walking
  int32_t split_parts;            // work-sharing build: parts in which the tiles at the head of the order are handed out (1 = whole)
  int32_t coop_rounds;            // work sharing: hand-over rounds per loop iteration
  int32_t regions;                // persistent kernel: number of tile queues (1, or 8 = one per XCD)
  uint32_t out_frame_stride;      // int32 words between the output buffers of two frames of a batch (0: one buffer for all: accumulation)
  int32_t region_start[9];        // identity order: region r owns tiles [region_start[r], region_start[r+1])
  WideMu wide_mu;                 // wide walk: the margin's scene constants (e = 0: none)
};

}  // namespace dr
