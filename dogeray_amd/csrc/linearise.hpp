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
};

int linearise(const HostScene& sc, DeviceImage& img);

}  // namespace dr
