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
#include <sched.h>
#include <thread>
#include <unistd.h>
#include <utility>

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

int usable_threads() {
  long n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (long)std::thread::hardware_concurrency();
  // cgroup v2 quota ("<quota> <period>" or "max <period>"): a container may see every CPU of the host and own a few
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    long period = 0;
    if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      long lim = (atol(q) + period / 2) / period;
      if (lim >= 1 && lim < n) n = lim;
    }
    fclose(f);
  }
  if (n < 1) n = 1;
  if (n > 256) n = 256;
  return (int)n;
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
  // a function of (line, column) only, so that the file can be parsed in pieces, in any order
  static float at(long line, int col) {
    uint64_t s = 0x853c49e6748fea9bull ^ ((uint64_t)line * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(col + 1) * 0xC2B2AE3D27D4EB4Full);
    s ^= s >> 33; s *= 0xff51afd7ed558ccdull; s ^= s >> 33; s *= 0xc4ceb9fe1a85ec53ull; s ^= s >> 33;
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

void apply_object(const HostScene& sc, dr_object& o, int col, Field f, long line) {
  float fv = 0;
  int iv = 0;
  const bool is_r = (f.n == 1 && f.p[0] == 'r');
  // which columns are floats / ints / names (K:1316-1503)
  const bool int_col = (col == 3 || col == 12 || col == 34 || col == 35);
  const bool name_col = (col == 36 || col == 37);
  if (col > 37) return;
  if (!name_col) {
    if (is_r) {
      fv = FieldRandom::at(line, col);
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

  // The file is cut into pieces at line boundaries and the pieces are converted by std::threads: object
  // lines are independent of each other, and a piece knows where its objects go once the object lines of
  // the pieces before it have been counted.  '*' lines are applied afterwards in file order (a later
  // settings line overrides an earlier one, field by field).
  struct Piece {
    const char* b; const char* e;
    size_t lines = 0, objects = 0;       // pass 1
    size_t first_line = 0, first_object = 0;
    std::vector<std::pair<const char*, const char*>> settings;   // '*' lines found
    std::vector<long> settings_line;
    bool failed = false; long err_line = 0; std::string err;
  };
  unsigned nthreads = (unsigned)usable_threads();
  size_t want = data.size() / (1 << 20) + 1;              // at least ~1 MB per piece
  if (want < nthreads) nthreads = (unsigned)want;
  std::vector<Piece> pieces(nthreads);
  {
    const char* b = p;
    for (unsigned i = 0; i < nthreads; i++) {
      const char* e = (i + 1 == nthreads) ? endp : p + data.size() * (i + 1) / nthreads;
      if (e < b) e = b;
      if (e < endp) { const char* nl = (const char*)memchr(e, '\n', (size_t)(endp - e)); e = nl ? nl + 1 : endp; }
      pieces[i].b = b; pieces[i].e = e;
      b = e;
    }
  }
  auto for_each_piece = [&](auto&& fn) {
    if (nthreads == 1) { fn(pieces[0]); return; }
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nthreads; i++) th.emplace_back([&, i]() { fn(pieces[i]); });
    for (auto& t : th) t.join();
  };
  // pass 1 = getnum: lines whose first byte is neither '/' nor '*' (an empty line counts, K:1143-1152)
  for_each_piece([&](Piece& pc) {
    for (const char* q = pc.b; q < pc.e;) {
      const char* nl = (const char*)memchr(q, '\n', (size_t)(pc.e - q));
      const char* le = nl ? nl : pc.e;
      char first = (le > q) ? *q : '\0';
      pc.lines++;
      if (first != '/' && first != '*') pc.objects++;
      q = nl ? nl + 1 : pc.e;
    }
  });
  size_t nobj = 0, nlines = 0;
  for (Piece& pc : pieces) { pc.first_object = nobj; pc.first_line = nlines; nobj += pc.objects; nlines += pc.lines; }
  if (nobj > (size_t)INT32_MAX / 4) { set_error("scene too large"); return DR_ERR_SCENE; }
  sc.n = (int)nobj;
  sc.objects.assign(nobj + 1, default_object());   // objnum = count + 1 (K:1158); slot N untouched

  // pass 2 = read: one iteration per comma-separated field, a trailing empty one included (K:1224,1303)
  for_each_piece([&](Piece& pc) {
    size_t obj = pc.first_object;
    long line = (long)pc.first_line;
    try {
      for (const char* q = pc.b; q < pc.e;) {
        const char* nl = (const char*)memchr(q, '\n', (size_t)(pc.e - q));
        const char* le = nl ? nl : pc.e;
        line++;
        char first = (le > q) ? *q : '\0';
        if (first == '*') {
          pc.settings.emplace_back(q, le);
          pc.settings_line.push_back(line);
        } else if (first != '/') {
          int col = 0;
          const char* f = q;
          for (;;) {
            const char* comma = (const char*)memchr(f, ',', (size_t)(le - f));
            const char* fe = comma ? comma : le;
            apply_object(sc, sc.objects[obj], col, Field{f, (size_t)(fe - f)}, line);
            col++;
            if (!comma) break;
            f = comma + 1;
          }
          obj++;
        }
        q = nl ? nl + 1 : pc.e;
      }
    } catch (LineError& e) {
      pc.failed = true; pc.err_line = line; pc.err = e.msg;
    }
  });
  // settings lines, in file order; an error on an earlier line wins over one on a later line
  long first_err_line = -1; std::string first_err;
  for (Piece& pc : pieces) {
    for (size_t k = 0; k < pc.settings.size() && first_err_line < 0; k++) {
      if (pc.failed && pc.settings_line[k] > pc.err_line) break;
      try {
        int col = 0;
        const char* f = pc.settings[k].first; const char* le = pc.settings[k].second;
        for (;;) {
          const char* comma = (const char*)memchr(f, ',', (size_t)(le - f));
          const char* fe = comma ? comma : le;
          apply_setting(sc, col, Field{f, (size_t)(fe - f)}, pc.settings_line[k]);
          col++;
          if (!comma) break;
          f = comma + 1;
        }
      } catch (LineError& e) {
        first_err_line = pc.settings_line[k]; first_err = e.msg;
      }
    }
    if (first_err_line >= 0) break;
    if (pc.failed) { first_err_line = pc.err_line; first_err = pc.err; break; }
  }
  if (first_err_line >= 0) {
    set_error(std::string(path) + ": " + first_err);
    return DR_ERR_PARSE;
  }
  return DR_OK;
}

}  // namespace dr
