// Host-side scene: what the reference keeps in `allobjects`, `nbvhtree`, the texture path
// list and its settings globals (kernel.cu K:119-132, K:2055-2094).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/dogeray_amd.h"

namespace dr {

struct HostTexture {
  int w = 0, h = 0;
  std::string path;
  std::vector<uint8_t> rgba;  // RGBA8, A = 0, rows as stored in the file
};

struct HostScene {
  std::vector<dr_object> objects;  // N + 1 entries; slot N is never written by the reader (K:2061)
  int n = 0;                       // object lines
  dr_settings settings;
  std::vector<HostTexture> textures;
  std::vector<dr_bvh_node> bvh;    // 2 * (N + 1) entries once built
  int bvh_used = 0;
};

void set_error(const std::string& msg);
const std::string& get_error();

int usable_threads();           // CPUs this process may use: affinity mask capped by the cgroup CPU quota
dr_object default_object();    // struct defaults K:55-71, everything else 0
dr_settings default_settings();  // K:29-30,109,123-132

// rts_reader.cpp
int scan_texture_dir(const char* dir, std::vector<std::string>& paths);
int load_ppm_rgba(const std::string& path, HostTexture& out);
int resolve_texture(const std::vector<HostTexture>& tex, const char* query, size_t len);
int read_rts(const char* path, HostScene& scene);

// bvh_builder.cpp
int build_bvh(HostScene& scene, int nthreads);

}  // namespace dr

struct dr_scene {
  dr::HostScene host;
};
