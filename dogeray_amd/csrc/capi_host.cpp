// Host half of the C ABI (include/dogeray_amd.h): scene ingest and BVH build.
// Replaces main()'s start-up sequence K:2055-2094.
#include <cstring>
#include <new>

#include "scene_host.hpp"

using namespace dr;

extern "C" {

const char* dr_last_error(void) { return get_error().c_str(); }
int dr_abi_version(void) { return DR_ABI_VERSION; }

int dr_scene_load(const char* rts_path, const char* texture_dir, dr_scene** out) {
  if (!rts_path || !out) { set_error("null argument"); return DR_ERR_INVALID; }
  *out = nullptr;
  dr_scene* s = new (std::nothrow) dr_scene();
  if (!s) { set_error("out of memory"); return DR_ERR_NOMEM; }
  s->host.settings = default_settings();
  int rc = DR_OK;
  try {
    std::vector<std::string> paths;
    rc = scan_texture_dir(texture_dir, paths);                 // getppmnum/getppmpaths K:2063-2068
    for (size_t i = 0; rc == DR_OK && i < paths.size(); i++) { // sdkLoadPPM4 K:1926
      HostTexture t;
      rc = load_ppm_rgba(paths[i], t);
      if (rc == DR_OK) s->host.textures.push_back(std::move(t));
    }
    if (rc == DR_OK) rc = read_rts(rts_path, s->host);         // getnum + read K:2055,2071
  } catch (std::bad_alloc&) {
    set_error("out of memory");
    rc = DR_ERR_NOMEM;
  }
  if (rc != DR_OK) { delete s; return rc; }
  *out = s;
  return DR_OK;
}

void dr_scene_free(dr_scene* s) { delete s; }

int dr_scene_create_from_arrays(const dr_object* objects, int n_objects, const dr_settings* settings,
                                const dr_bvh_node* bvh, int bvhnum, dr_scene** out) {
  if (!objects || !out || n_objects < 0) { set_error("bad argument"); return DR_ERR_INVALID; }
  *out = nullptr;
  if (bvh && bvhnum != 2 * (n_objects + 1)) { set_error("bvhnum must be 2 * (n_objects + 1) (K:2073)"); return DR_ERR_INVALID; }
  dr_scene* s = new (std::nothrow) dr_scene();
  if (!s) { set_error("out of memory"); return DR_ERR_NOMEM; }
  try {
    s->host.n = n_objects;
    s->host.objects.assign(objects, objects + n_objects + 1);
    s->host.settings = settings ? *settings : default_settings();
    if (bvh) {
      s->host.bvh.assign(bvh, bvh + bvhnum);
      s->host.bvh_used = 2 * n_objects - 1;
    }
  } catch (std::bad_alloc&) {
    delete s;
    set_error("out of memory");
    return DR_ERR_NOMEM;
  }
  *out = s;
  return DR_OK;
}

int dr_scene_add_texture(dr_scene* s, const uint8_t* rgba, int width, int height, const char* name) {
  if (!s || !rgba || width <= 0 || height <= 0) { set_error("bad argument"); return DR_ERR_INVALID; }
  HostTexture t;
  t.w = width; t.h = height; t.path = name ? name : "";
  t.rgba.assign(rgba, rgba + (size_t)width * height * 4);
  s->host.textures.push_back(std::move(t));
  return (int)s->host.textures.size() - 1;
}

int dr_scene_num_objects(const dr_scene* s) { return s ? s->host.n : DR_ERR_INVALID; }

int dr_scene_get_objects(const dr_scene* s, dr_object* out) {
  if (!s || !out) { set_error("null argument"); return DR_ERR_INVALID; }
  memcpy(out, s->host.objects.data(), s->host.objects.size() * sizeof(dr_object));
  return DR_OK;
}

int dr_scene_get_settings(const dr_scene* s, dr_settings* out) {
  if (!s || !out) { set_error("null argument"); return DR_ERR_INVALID; }
  *out = s->host.settings;
  return DR_OK;
}

int dr_scene_set_settings(dr_scene* s, const dr_settings* in) {
  if (!s || !in) { set_error("null argument"); return DR_ERR_INVALID; }
  if (in->backtex >= (int)s->host.textures.size()) { set_error("backtex out of range"); return DR_ERR_INVALID; }
  s->host.settings = *in;
  return DR_OK;
}

int dr_scene_num_textures(const dr_scene* s) { return s ? (int)s->host.textures.size() : DR_ERR_INVALID; }

int dr_scene_texture_info(const dr_scene* s, int i, int* width, int* height) {
  if (!s || i < 0 || i >= (int)s->host.textures.size()) { set_error("texture index out of range"); return DR_ERR_INVALID; }
  if (width) *width = s->host.textures[(size_t)i].w;
  if (height) *height = s->host.textures[(size_t)i].h;
  return DR_OK;
}

int dr_scene_texture_data(const dr_scene* s, int i, uint8_t* rgba) {
  if (!s || !rgba || i < 0 || i >= (int)s->host.textures.size()) { set_error("texture index out of range"); return DR_ERR_INVALID; }
  const HostTexture& t = s->host.textures[(size_t)i];
  memcpy(rgba, t.rgba.data(), t.rgba.size());
  return DR_OK;
}

int dr_scene_build_bvh(dr_scene* s, int nthreads) {
  if (!s) { set_error("null argument"); return DR_ERR_INVALID; }
  try {
    return build_bvh(s->host, nthreads);
  } catch (std::bad_alloc&) {
    set_error("out of memory");
    return DR_ERR_NOMEM;
  }
}

int dr_scene_bvh_size(const dr_scene* s) { return s ? (int)s->host.bvh.size() : DR_ERR_INVALID; }
int dr_scene_bvh_used(const dr_scene* s) { return s ? s->host.bvh_used : DR_ERR_INVALID; }

int dr_scene_get_bvh(const dr_scene* s, dr_bvh_node* out) {
  if (!s || !out) { set_error("null argument"); return DR_ERR_INVALID; }
  memcpy(out, s->host.bvh.data(), s->host.bvh.size() * sizeof(dr_bvh_node));
  return DR_OK;
}

}  // extern "C"
