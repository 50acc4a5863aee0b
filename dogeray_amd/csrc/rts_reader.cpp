// .rts scene reader and .ppm texture loader.
//
// Behaviour follows the reference reader (kernel.cu getnum K:1113-1169, read K:1186-1530,
// gettexnum K:1172-1183, getppmnum/getppmpaths K:1979-2018, readtextures' sdkLoadPPM4 call
// K:1926), re-designed for large files: the whole file is read once and fields are converted
// in place, instead of one stringstream per line.  A 1M-triangle, 38-column scene is ~330 MB.
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <unistd.h>

#include "scene_host.hpp"

namespace dr {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
const std::string& get_error() { return g_error; }

dr_object default_object() {
  dr_object o;
  memset(&o, 0, sizeof(o));
  const float absent[3] = {-2, -3, -20};                  // K:55-60
  memcpy(o.norm, absent, sizeof(absent));
  memcpy(o.n1, absent, sizeof(absent));
  memcpy(o.n2, absent, sizeof(absent));
  memcpy(o.n3, absent, sizeof(absent));
  o.t1[0] = 0; o.t1[1] = 1;                                // K:62-64
  o.t2[0] = 0; o.t2[1] = 0;
  o.t3[0] = 1; o.t3[1] = 0;
  o.texnum = -1; o.rtexnum = -1;                           // K:70-71
  return o;
}

dr_settings default_settings() {
  dr_settings s;
  memset(&s, 0, sizeof(s));
  s.campos[2] = 2;          // K:125
  s.aperture = 0.01f;       // K:127
  s.focus_dist = 3;         // K:128
  s.fov = 45;               // K:132
  s.max_depth = 50;         // K:130
  s.spp = 1;                // K:131
  s.background = 1;         // K:109
  s.backtex = -1;           // K:123
  s.width = 1280;           // K:29
  s.height = 720;           // K:30
  return s;
}

// ---------------------------------------------------------------------------- textures
int scan_texture_dir(const char* dir, std::vector<std::string>& paths) {
  std::string base;
  if (dir == nullptr) {
    char buf[4096];
    if (!getcwd(buf, sizeof(buf))) { set_error("getcwd failed"); return DR_ERR_IO; }
    base = buf;
  } else {
    base = dir;
  }
  if (base.empty()) return DR_OK;   // "" = no textures
  DIR* d = opendir(base.c_str());
  if (!d) { set_error("cannot open texture directory " + base); return DR_ERR_IO; }
  std::vector<std::string> names;
  while (dirent* e = readdir(d)) {
    if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
    names.emplace_back(e->d_name);
  }
  closedir(d);
  std::sort(names.begin(), names.end());
  for (const std::string& n : names) {
    std::string full = base + "/" + n;
    // the reference matches on the whole path, not the file name (K:1986)
    if (full.find("ppm") != std::string::npos || full.find("PPM") != std::string::npos) paths.push_back(full);
  }
  return DR_OK;
}

namespace {
struct FileCloser {
  FILE* f;
  ~FileCloser() { if (f) fclose(f); }
};

bool ppm_token(FILE* f, std::string& out) {
  out.clear();
  int c = fgetc(f);
  for (;;) {
    while (c != EOF && isspace(c)) c = fgetc(f);
    if (c != '#') break;
    while (c != EOF && c != '\n') c = fgetc(f);
  }
  while (c != EOF && !isspace(c)) {
    out.push_back((char)c);
    c = fgetc(f);
  }
  return !out.empty();
}
}  // namespace

// P6 (or P5) -> RGBA8 with A = 0, rows in file order (what sdkLoadPPM4 hands to readtextures).
int load_ppm_rgba(const std::string& path, HostTexture& out) {
  FileCloser fc{fopen(path.c_str(), "rb")};
  if (!fc.f) { set_error("cannot open texture " + path); return DR_ERR_IO; }
  std::string magic, tw, th, tmax;
  if (!ppm_token(fc.f, magic) || !ppm_token(fc.f, tw) || !ppm_token(fc.f, th) || !ppm_token(fc.f, tmax)) {
    set_error("truncated PPM header in " + path);
    return DR_ERR_PARSE;
  }
  int channels = magic == "P6" ? 3 : (magic == "P5" ? 1 : 0);
  long w = strtol(tw.c_str(), nullptr, 10), h = strtol(th.c_str(), nullptr, 10);
  if (channels == 0 || w <= 0 || h <= 0 || w > 65536 || h > 65536) {
    set_error("unsupported PPM (need P6/P5 with sane size): " + path);
    return DR_ERR_PARSE;
  }
  size_t px = (size_t)w * (size_t)h;
  std::vector<uint8_t> raw(px * channels);
  if (fread(raw.data(), 1, raw.size(), fc.f) != raw.size()) {
    set_error("truncated PPM payload in " + path);
    return DR_ERR_PARSE;
  }
  out.w = (int)w; out.h = (int)h; out.path = path;
  out.rgba.assign(px * 4, 0);
  uint8_t* dst = out.rgba.data();
  const uint8_t* src = raw.data();
  if (channels == 3) {
    for (size_t i = 0; i < px; i++, dst += 4, src += 3) { dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; }
  } else {
    for (size_t i = 0; i < px; i++, dst += 4, src += 1) { dst[0] = dst[1] = dst[2] = src[0]; }
  }
  return DR_OK;
}

// gettexnum K:1172-1183: first texture whose lower-cased path contains the query verbatim.
int resolve_texture(const std::vector<HostTexture>& tex, const char* query, size_t len) {
  std::string q(query, len);
  for (size_t i = 0; i < tex.size(); i++) {
    std::string p = tex[i].path;
    for (char& c : p) c = (char)tolower((unsigned char)c);
    if (p.find(q) != std::string::npos) return (int)i;
  }
  return -1;
}

// ---------------------------------------------------------------------------- .rts
namespace {

struct Field { const char* p; size_t n; };

struct LineError { std::string msg; };

// std::stof / std::stoi on one field: leading whitespace skipped, longest numeric prefix
// converted, trailing characters ignored; nothing convertible (or out of range) is an error
// where the reference would have thrown.
float field_float(Field f, long line) {
  char buf[64];
  std::string big;
  const char* s;
  if (f.n < sizeof(buf)) { memcpy(buf, f.p, f.n); buf[f.n] = 0; s = buf; }
  else { big.assign(f.p, f.n); s = big.c_str(); }
  char* end = nullptr;
  errno = 0;
  float v = strtof(s, &end);
  if (end == s) throw LineError{"line " + std::to_string(line) + ": stof: no conversion in '" + std::string(f.p, f.n) + "'"};
  if (errno == ERANGE) throw LineError{"line " + std::to_string(line) + ": stof: out of range '" + std::string(f.p, f.n) + "'"};
  return v;
}
int field_int(Field f, long line) {
  char buf[64];
  std::string big;
  const char* s;
  if (f.n < sizeof(buf)) { memcpy(buf, f.p, f.n); buf[f.n] = 0; s = buf; }
  else { big.assign(f.p, f.n); s = big.c_str(); }
  char* end = nullptr;
  errno = 0;
  long v = strtol(s, &end, 10);
  if (end == s) throw LineError{"line " + std::to_string(line) + ": stoi: no conversion in '" + std::string(f.p, f.n) + "'"};
  if (errno == ERANGE || v < INT32_MIN || v > INT32_MAX) throw LineError{"line " + std::to_string(line) + ": stoi: out of range"};
  return (int)v;
}
inline bool is_no(Field f) { return f.n == 2 && f.p[0] == 'n' && f.p[1] == 'o'; }

// The reference replaces a field "r" by rand()/(RAND_MAX+1.0) seeded from the tick count
// (K:1098-1102,1308-1311), printed with to_string (6 decimals).  That is not reproducible;
// here the value comes from a fixed-seed generator so that a scene loads the same way twice.
struct FieldRandom {
  uint64_t s = 0x853c49e6748fea9bull;
  float next() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double r = (double)(s >> 11) * (1.0 / 9007199254740992.0);
    char buf[32];
    snprintf(buf, sizeof(buf), "%f", r);
    return strtof(buf, nullptr);
  }
};

void apply_setting(HostScene& sc, int col, Field f, long line) {
  dr_settings& g = sc.settings;
  switch (col) {                                   // K:1230-1293
    case 1: g.campos[0] = field_float(f, line); break;
    case 2: g.campos[1] = field_float(f, line); break;
    case 3: g.campos[2] = field_float(f, line); break;
    case 4: g.aperture = field_float(f, line); break;
    case 5: g.look[0] = field_float(f, line); break;
    case 6: g.look[1] = field_float(f, line); break;
    case 7: g.look[2] = field_float(f, line); break;
    case 8: g.focus_dist = field_float(f, line); break;
    case 9: g.fov = field_int(f, line); break;
    case 10: g.max_depth = field_int(f, line); break;
    case 11: g.spp = field_int(f, line); break;
    case 12: g.background = field_float(f, line); break;
    case 13: if (!is_no(f)) g.backtex = resolve_texture(sc.textures, f.p, f.n); break;
    case 14: g.width = field_int(f, line); break;
    case 15: g.height = field_int(f, line); break;
    default: break;
  }
}

void apply_object(HostScene& sc, dr_object& o, int col, Field f, long line, FieldRandom& rnd) {
  float fv = 0;
  int iv = 0;
  const bool is_r = (f.n == 1 && f.p[0] == 'r');
  // which columns are floats / ints / names (K:1316-1503)
  const bool int_col = (col == 3 || col == 12 || col == 34 || col == 35);
  const bool name_col = (col == 36 || col == 37);
  if (col > 37) return;
  if (!name_col) {
    if (is_r) {
      fv = rnd.next();
      iv = (int)fv;   // stoi("0.xxxxxx") == 0
    } else if (int_col) {
      iv = field_int(f, line);
    } else {
      fv = field_float(f, line);
    }
  }
  switch (col) {
    case 0: o.pos[0] = fv; break;
    case 1: o.pos[1] = fv; break;
    case 2: o.pos[2] = fv; break;
    case 3: o.type = iv; break;
    case 4: o.col[0] = fv; break;
    case 5: o.col[1] = fv; break;
    case 6: o.col[2] = fv; break;
    case 7: o.addional[1] = fv; break;
    case 8: o.addional[0] = fv; break;
    case 9: o.dim[0] = fv; break;
    case 10: o.dim[1] = fv; break;
    case 11: o.dim[2] = fv; break;
    case 12: o.mat = iv; break;
    case 13: o.rot[0] = fv; break;
    case 14: o.rot[1] = fv; break;
    case 15: o.rot[2] = fv; break;
    case 16: o.norm[0] = fv; break;
    case 17: o.norm[1] = fv; break;
    case 18: o.norm[2] = fv; break;
    case 19: o.n1[0] = fv; break;
    case 20: o.n1[1] = fv; break;
    case 21: o.n1[2] = fv; break;
    case 22: o.n2[0] = fv; break;
    case 23: o.n2[1] = fv; break;
    case 24: o.n2[2] = fv; break;
    case 25: o.n3[0] = fv; break;
    case 26: o.n3[1] = fv; break;
    case 27: o.n3[2] = fv; break;
    case 28: o.t1[0] = fv; break;
    case 29: o.t1[1] = fv; break;
    case 30: o.t2[0] = fv; break;
    case 31: o.t2[1] = fv; break;
    case 32: o.t3[0] = fv; break;
    case 33: o.t3[1] = fv; break;
    case 34: if (iv == 1) o.smooth = 1; break;
    case 35: if (iv == 1) o.tex = 1; break;
    case 36: if (!is_no(f)) o.texnum = resolve_texture(sc.textures, f.p, f.n); break;
    case 37: if (!is_no(f)) o.rtexnum = resolve_texture(sc.textures, f.p, f.n); break;
    default: break;
  }
}

}  // namespace

int read_rts(const char* path, HostScene& sc) {
  FileCloser fc{fopen(path, "rb")};
  if (!fc.f) { set_error(std::string("cannot open scene ") + path); return DR_ERR_IO; }
  std::string data;
  {
    char chunk[1 << 16];
    size_t got;
    fseek(fc.f, 0, SEEK_END);
    long sz = ftell(fc.f);
    fseek(fc.f, 0, SEEK_SET);
    if (sz > 0) data.reserve((size_t)sz);
    while ((got = fread(chunk, 1, sizeof(chunk), fc.f)) > 0) data.append(chunk, got);
  }
  const char* p = data.data();
  const char* endp = p + data.size();

  // getnum: lines whose first byte is neither '/' nor '*' (an empty line counts, K:1143-1152)
  size_t nobj = 0;
  for (const char* q = p; q < endp;) {
    const char* nl = (const char*)memchr(q, '\n', (size_t)(endp - q));
    const char* le = nl ? nl : endp;
    char first = (le > q) ? *q : '\0';
    if (first != '/' && first != '*') nobj++;
    q = nl ? nl + 1 : endp;
  }
  if (nobj > (size_t)INT32_MAX / 4) { set_error("scene too large"); return DR_ERR_SCENE; }
  sc.n = (int)nobj;
  sc.objects.assign(nobj + 1, default_object());   // objnum = count + 1 (K:1158); slot N untouched

  FieldRandom rnd;
  size_t obj = 0;
  long line = 0;
  try {
    for (const char* q = p; q < endp;) {
      const char* nl = (const char*)memchr(q, '\n', (size_t)(endp - q));
      const char* le = nl ? nl : endp;
      line++;
      char first = (le > q) ? *q : '\0';
      if (first != '/') {
        const bool is_settings = first == '*';
        // one iteration per comma-separated field, a trailing empty one included (K:1224,1303)
        int col = 0;
        const char* f = q;
        for (;;) {
          const char* comma = (const char*)memchr(f, ',', (size_t)(le - f));
          const char* fe = comma ? comma : le;
          Field fld{f, (size_t)(fe - f)};
          if (is_settings) apply_setting(sc, col, fld, line);
          else apply_object(sc, sc.objects[obj], col, fld, line, rnd);
          col++;
          if (!comma) break;
          f = comma + 1;
        }
        if (!is_settings) obj++;
      }
      q = nl ? nl + 1 : endp;
    }
  } catch (LineError& e) {
    set_error(std::string(path) + ": " + e.msg);
    return DR_ERR_PARSE;
  }
  return DR_OK;
}

}  // namespace dr
