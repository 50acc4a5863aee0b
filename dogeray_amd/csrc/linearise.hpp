#pragma once
#include <vector>

#include "device_layout.h"
#include "scene_host.hpp"

namespace dr {

struct DeviceImage {
  std::vector<DevUnit> walk;       // 2(N-1) + 5N units: the threaded walk (device_layout.h)
  std::vector<DevPair> pairs;      // N - 1, pre-order among internal nodes; pair 0 = root's children
  std::vector<DevPrim> prims;      // N, leaf order
  std::vector<DevShade> shade;     // N, leaf order
  std::vector<DevTex> tex;
  std::vector<uint32_t> texels;
  std::vector<int> slot_to_orig;   // leaf rank -> object index in the file
  float root_mn[3], root_mx[3];
  std::vector<DevUnit> wide;       // wide walk records, empty if the scene is not representable
  int wide_depth = 0, wide_nodes = 0;
  float wide_pmax = 0;
  WideMu wide_mu = {0, 0, 0};      // wide_tree = 2: the margin's scene constants (device_layout.h)
  int wide_own_bounds = 0;         // ... and how many triangles entered the tree with their own bounds
};

// wide_builder.cpp: the 4-way traversal structure over the reference's leaves (device_layout.h "wide walk")
struct WideImage {
  std::vector<DevUnit> rec;        // WIDE_UNITS per record, root = record 0
  int depth = 0, nodes = 0, leaves = 0;
  float pmax = 0;                  // largest |decoded plane coordinate| over all nodes
  WideMu mu = {0, 0, 0};           // tree_mode 2: constants of the rays' margin (e = 0: every leaf entered with the reference's box)
  int own_bounds = 0;              // ... number of triangles entered with their own bounds
};
// tree_mode 2 (default): binned-SAH tree, small triangles entered with their own bounds (the rays carry a margin); 1: the same over the reference's leaf boxes;
// 0: the reference's topology collapsed.  false = not representable.
bool build_wide(const HostScene& sc, const std::vector<int>& leaf_node_of_slot, const std::vector<DevPrim>& prims, int tree_mode,
                int nthreads, WideImage& out);

int linearise(const HostScene& sc, DeviceImage& img, int wide_tree_mode = 2);

}  // namespace dr
