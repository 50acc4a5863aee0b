// Host half of the C ABI (include/dogeray_amd.h): scene ingest and BVH build.
// Replaces main()'s start-up sequence K:2055-2094.
#include <cstdio>
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

// ------------------------------------------------------------------ .rtsb sidecar
namespace {
struct RtsbHeader {
  char magic[8];                       // "DRTSB\0\0\0"
  uint32_t version, abi;
  uint32_t sizeof_object, sizeof_node, sizeof_settings, reserved;
  int32_t n, bvh_count, bvh_used, ntex;
  uint64_t payload_bytes, checksum;    // of everything after the header
};
const char RTSB_MAGIC[8] = {'D', 'R', 'T', 'S', 'B', 0, 0, 0};
const uint32_t RTSB_VERSION = 1;

// 64-bit multiply-xor over 8-byte words (tail bytes zero-padded): a transfer/truncation check, not cryptography
struct Checksum {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  uint64_t carry = 0; int ncarry = 0;
  void word(uint64_t w) { h = (h ^ w) * 0x100000001B3ull; h ^= h >> 29; }
  void add(const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    while (n > 0 && ncarry > 0) { carry |= (uint64_t)*b++ << (8 * ncarry); n--; if (++ncarry == 8) { word(carry); carry = 0; ncarry = 0; } }
    for (; n >= 8; n -= 8, b += 8) { uint64_t w; memcpy(&w, b, 8); word(w); }
    for (; n > 0; n--) { carry |= (uint64_t)*b++ << (8 * ncarry); ncarry++; }
  }
  uint64_t done() { if (ncarry) word(carry); return h; }
};

struct Writer {
  FILE* f = nullptr; Checksum ck; uint64_t bytes = 0; bool ok = true;
  void put(const void* p, size_t n) { if (n == 0) return; ck.add(p, n); bytes += n; if (ok && fwrite(p, 1, n, f) != n) ok = false; }
};
struct Reader {
  FILE* f = nullptr; Checksum ck; uint64_t left = 0; bool ok = true;
  bool get(void* p, size_t n) {
    if (n == 0) return ok;
    if (!ok || n > left || fread(p, 1, n, f) != n) { ok = false; return false; }
    ck.add(p, n); left -= n;
    return true;
  }
};
}  // namespace

extern "C" {

int dr_scene_save_binary(const dr_scene* s, const char* path) {
  if (!s || !path) { set_error("null argument"); return DR_ERR_INVALID; }
  const HostScene& h = s->host;
  FILE* f = fopen(path, "wb");
  if (!f) { set_error(std::string("cannot create ") + path); return DR_ERR_IO; }
  RtsbHeader hd;
  memset(&hd, 0, sizeof hd);
  memcpy(hd.magic, RTSB_MAGIC, 8);
  hd.version = RTSB_VERSION; hd.abi = DR_ABI_VERSION;
  hd.sizeof_object = sizeof(dr_object); hd.sizeof_node = sizeof(dr_bvh_node); hd.sizeof_settings = sizeof(dr_settings);
  hd.n = h.n; hd.bvh_count = (int32_t)h.bvh.size(); hd.bvh_used = h.bvh_used; hd.ntex = (int32_t)h.textures.size();
  Writer w;
  w.f = f;
  if (fwrite(&hd, 1, sizeof hd, f) != sizeof hd) w.ok = false;       // placeholder, rewritten below with size + checksum
  w.put(&h.settings, sizeof h.settings);
  w.put(h.objects.data(), h.objects.size() * sizeof(dr_object));
  w.put(h.bvh.data(), h.bvh.size() * sizeof(dr_bvh_node));
  for (const HostTexture& t : h.textures) {
    int32_t meta[3] = {t.w, t.h, (int32_t)t.path.size()};
    w.put(meta, sizeof meta);
    w.put(t.path.data(), t.path.size());
    w.put(t.rgba.data(), t.rgba.size());
  }
  hd.payload_bytes = w.bytes;
  hd.checksum = w.ck.done();
  if (w.ok && (fseek(f, 0, SEEK_SET) != 0 || fwrite(&hd, 1, sizeof hd, f) != sizeof hd)) w.ok = false;
  if (fclose(f) != 0) w.ok = false;
  if (!w.ok) { remove(path); set_error(std::string("write failed: ") + path); return DR_ERR_IO; }
  return DR_OK;
}

int dr_scene_load_binary(const char* path, dr_scene** out) {
  if (!path || !out) { set_error("null argument"); return DR_ERR_INVALID; }
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) { set_error(std::string("cannot open ") + path); return DR_ERR_IO; }
  dr_scene* s = nullptr;
  int rc = DR_ERR_PARSE;
  try {
    RtsbHeader hd;
    const char* why = nullptr;
    if (fread(&hd, 1, sizeof hd, f) != sizeof hd || memcmp(hd.magic, RTSB_MAGIC, 8) != 0) why = "not a .rtsb file";
    else if (hd.version != RTSB_VERSION || hd.abi != (uint32_t)DR_ABI_VERSION) why = ".rtsb written by another version";
    else if (hd.sizeof_object != sizeof(dr_object) || hd.sizeof_node != sizeof(dr_bvh_node) || hd.sizeof_settings != sizeof(dr_settings))
      why = ".rtsb record sizes differ from this ABI";
    else if (hd.n < 0 || hd.ntex < 0 || (hd.bvh_count != 0 && hd.bvh_count != 2 * (hd.n + 1))) why = ".rtsb header is inconsistent";
    if (!why) {
      s = new dr_scene();
      HostScene& h = s->host;
      Reader r;
      r.f = f; r.left = hd.payload_bytes;
      h.n = hd.n; h.bvh_used = hd.bvh_used;
      r.get(&h.settings, sizeof h.settings);
      // sizes are checked against the declared payload before anything is allocated
      const uint64_t need = ((uint64_t)hd.n + 1) * sizeof(dr_object) + (uint64_t)hd.bvh_count * sizeof(dr_bvh_node);
      if (!r.ok || need > r.left) why = ".rtsb is truncated";
      if (!why) {
        h.objects.resize((size_t)hd.n + 1);
        h.bvh.resize((size_t)hd.bvh_count);
        r.get(h.objects.data(), h.objects.size() * sizeof(dr_object));
        r.get(h.bvh.data(), h.bvh.size() * sizeof(dr_bvh_node));
        for (int i = 0; r.ok && i < hd.ntex; i++) {
          int32_t meta[3];
          if (!r.get(meta, sizeof meta)) break;
          if (meta[0] <= 0 || meta[1] <= 0 || meta[2] < 0 || (uint64_t)meta[2] + (uint64_t)meta[0] * (uint64_t)meta[1] * 4 > r.left) { r.ok = false; break; }
          HostTexture t;
          t.w = meta[0]; t.h = meta[1];
          t.path.resize((size_t)meta[2]);
          t.rgba.resize((size_t)t.w * (size_t)t.h * 4);
          r.get(&t.path[0], t.path.size());
          r.get(t.rgba.data(), t.rgba.size());
          h.textures.push_back(std::move(t));
        }
        if (!r.ok || r.left != 0 || fgetc(f) != EOF) why = ".rtsb is truncated or has trailing data";
        else if (r.ck.done() != hd.checksum) why = ".rtsb checksum mismatch";
        else if (h.settings.backtex >= (int)h.textures.size()) why = ".rtsb settings name a texture it does not hold";
      }
    }
    if (why) set_error(std::string(why) + ": " + path);
    else rc = DR_OK;
  } catch (std::bad_alloc&) {
    set_error("out of memory");
    rc = DR_ERR_NOMEM;
  }
  fclose(f);
  if (rc != DR_OK) { delete s; return rc; }
  *out = s;
  return DR_OK;
}

}  // extern "C"
