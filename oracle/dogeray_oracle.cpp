// =====================================================================================
// dogeray_oracle.cpp  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
//
// A scalar CPU restatement of the DOGERAY render path (reference: raygpu/kernel.cu,
// cited below as K:<line>).  It exists so that tests/, __graft_entry__.smoke() and the
// `cpu_baseline` leg of bench.py can check / time the HIP path against an independent
// statement of the reference algorithm.  Nothing under dogeray_amd/ may include, link,
// dlopen or call anything in this directory.
//
// PARITY STATUS: pinned statistically, not bit for bit.  The reference ships no tests and no golden
// vectors, seeds cuRAND from clock() (K:1065), and cannot be built in this image (it needs
// cuda_runtime.h, curand_kernel.h, SDL.h, Windows.h and the CUDA-samples helper headers, none of
// which exist here; writing stand-ins for them is not allowed).  What it does hold is two frames it
// saved itself (SDL_SaveBMP, K:2505-2513) together with their inputs: images/eorovan.blend.rts.bmp
// and images/bolter2.blend.rts.bmp for samples/eorovan.blend.rts and samples/bolter2.blend.rts
// (+ boltersmall.ppm, env.ppm).  This oracle reproduces both to Monte-Carlo noise
// (tests/test_oracle.py::test_oracle_reproduces_the_reference_image*, fixtures in
// tests/golden/reference_image/): sky pixels to a grey level, silhouette IoU 0.998, the textured
// scene with per-pixel mean |diff| 1.4 of 255 and 16x16-block correlation 0.99999, no bias.
// What those two frames pin: ingest (.rts, .ppm, texture names), the BVH and intersection code (silhouettes), the camera
// including the window's key-step offsets, the sky gradient, the environment-map lookup, the albedo texture lookup, the
// METAL material (3) with smooth normals -- the only material the compared regions contain -- accumulation and the display
// divide.  The van of eorovan.blend.rts is diffuse (material 0) but its texture blob is missing from the reference tree, so
// the van is excluded from the comparison: DIFFUSE IS NOT PINNED.  Also not pinned by anything the reference holds: mirror,
// glass, glossy and emissive materials, spheres, roughness maps, the checker flag, the cuRAND bit stream (only its
// statistics), CUDA libm ulps, nvcc's default FMA contraction (-fmad=true contracts a*b+c in aabb2 / hit_tri / the dot
// products; this restatement follows the source text, -ffp-contract=off).  For all of those the oracle rests on the reading
// of the source, line by line.
// The third-party pieces it restates from their published definitions are:
//   * cuRAND XORWOW (CUDA 11.2 curand_kernel.h): curand_init(seed,0,0), curand(),
//     curand_uniform_double()                         -> struct Xorwow below
//   * CUDA texture unit, point filter / wrap / normalised coordinates (K:1959-1964)
//                                                     -> tex_fetch() below
//   * sdkLoadPPM4 (CUDA samples helper_image.h), RGB -> RGBA with A = 0
//                                                     -> load_ppm4() below
//
// Arithmetic contract (where C++ leaves room, or CUDA and host C++ would differ):
//   C1  pow(x, 2.0f) on floats (K:320,324,645,681,991) is restated as the correctly
//       rounded product x*x; pow(d, 2.) on doubles (K:958) as d*d.  (Define
//       ORACLE_LIBM_POW to call libm's powf/pow instead: used by tests to show how far
//       a 1-ulp libm difference moves pixels.)
//   C2  pow(float, int 5) (K:690) follows CUDA's float overload (powif: square-and-
//       multiply in float), not host C++'s promotion to double.
//   C3  min(float, double) (K:679,920) follows CUDA's overload: computed in double.
//   C4  float -> int conversions (K:802,1076,1083-1085) saturate and map NaN to 0, as
//       CUDA's cvt.rzi.s32.f32 does (host C++ leaves them undefined).
//   C5  the three draws inside one make_float3(...) (K:644) and the two inside K:990 are
//       taken left to right (x first).  C++ leaves the order unspecified.
//   C6  texco is zero where the reference leaves it uninitialised (spheres, K:798,707).
//   C7  an object whose type is neither 0 nor 2 is never hit (the reference reads an
//       uninitialised value, K:438-447); the never-written object slot N (K:2061,1899)
//       and every field a short line leaves unwritten are zero.
//   C8  the RNG seed is  frame_seed + sample*SPP_SEED_STRIDE + (x + y*8*(W/div/8)),
//       i.e. clock() at K:1065 is replaced by a caller-supplied 64-bit value.
//   C9  pixels outside the rendered sub-rectangle (K:2633-2636) are 0.
//   C10 the viewport's tan() (K:1023) is the float overload (tanf), as in CUDA.
// Compile with -ffp-contract=off and without -ffast-math.
// =====================================================================================
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace {

static const uint64_t SPP_SEED_STRIDE = 0x9E3779B97F4A7C15ull;  // C8

// ------------------------------------------------------------------ vectors (K:144-232)
struct V3 { float x, y, z; };
inline V3 v3(float a, float b, float c) { V3 r; r.x = a; r.y = b; r.z = c; return r; }
inline V3 splat(float a) { return v3(a, a, a); }                      // K:153 make3
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }  // K:212
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }  // K:217
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }  // K:222
inline V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }  // K:227
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      // K:168
inline V3 cross(V3 a, V3 b) {                                                    // K:148
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length(V3 a) { return sqrtf(dot(a, a)); }                           // K:200
inline V3 normalized(V3 v) {                                                     // K:179
  float inv = 1.0f / sqrtf(dot(v, v));
  return v3(v.x * inv, v.y * inv, v.z * inv);
}

inline float sqf(float x) {  // C1
#ifdef ORACLE_LIBM_POW
  return powf(x, 2.0f);
#else
  return x * x;
#endif
}
inline double sqd(double x) {  // C1
#ifdef ORACLE_LIBM_POW
  return pow(x, 2.0);
#else
  return x * x;
#endif
}
inline float powi5(float a) {  // C2: square-and-multiply, exponent 5 = 0b101
  float r = a;        // bit 0
  a = a * a;          // a^2
  a = a * a;          // a^4 (bit 1 clear)
  r = r * a;          // bit 2
  return r;
}
inline int f2i(float f) {  // C4
  if (f != f) return 0;
  if (f >= 2147483648.0f) return INT32_MAX;
  if (f <= -2147483648.0f) return INT32_MIN;
  return (int)f;
}

// ------------------------------------------------------------------ scene types (K:48-96)
struct Obj {           // K:48-74 (defaults K:55-71)
  int type = 0;
  V3 pos = {0, 0, 0};
  V3 rot = {0, 0, 0};
  V3 norm = {-2, -3, -20};
  V3 n1 = {-2, -3, -20}, n2 = {-2, -3, -20}, n3 = {-2, -3, -20};
  V3 t1 = {0, 1, 0}, t2 = {0, 0, 0}, t3 = {1, 0, 0};
  bool smooth = false;
  bool tex = false;
  int mat = 0;
  V3 dim = {0, 0, 0};
  V3 col = {0, 0, 0};
  int texnum = -1;
  int rtexnum = -1;
  V3 addional = {0, 0, 0};
};

struct Node {          // K:79-96
  bool active = false;
  int children[2] = {0, 0};
  int count = 0;
  int hit_node = 0;
  int miss_node = 0;
  int under = 0;
  V3 min = {0, 0, 0};
  V3 max = {0, 0, 0};
  bool end = false;
};

struct Texture { int w = 0, h = 0; std::vector<uint8_t> rgba; };

struct Settings {      // globals K:29-30,109,123-132
  V3 campos = {0, 0, 2};
  V3 look = {0, 0, 0};
  float aperture = 0.01f;
  float focus_dist = 3;
  int fov = 45;
  int max_depth = 50;
  int spp = 1;
  float background = 1;
  int backtex = -1;
  int width = 1280, height = 720;
};

struct Scene {
  std::vector<Obj> objs;   // N+1 entries (K:1158,2061); slot N stays default/zero (C7)
  int nanum = 0;           // K:1518 = N+1
  std::vector<Node> bvh;   // 2*(N+1) entries (K:2073)
  int next_node = 0;       // actualbvhnum K:129
  Settings set;
  std::vector<std::string> texpaths;
  std::vector<Texture> tex;
  std::string err;
};

// ------------------------------------------------------------------ textures
// sdkLoadPPM4 restated: header tokens separated by whitespace, '#' comment
// lines skipped, then w*h*3 raw bytes; expanded to RGBA with A = 0.
bool load_ppm4(const std::string& path, Texture& t) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  auto token = [&](std::string& out) -> bool {
    out.clear();
    int c = fgetc(f);
    for (;;) {
      while (c != EOF && isspace(c)) c = fgetc(f);
      if (c == '#') { while (c != EOF && c != '\n') c = fgetc(f); continue; }
      break;
    }
    while (c != EOF && !isspace(c)) { out.push_back((char)c); c = fgetc(f); }
    return !out.empty();   // exactly one whitespace byte consumed after the token
  };
  std::string magic, sw, sh, smax;
  bool ok = token(magic) && token(sw) && token(sh) && token(smax);
  int ch = magic == "P6" ? 3 : (magic == "P5" ? 1 : 0);
  if (!ok || ch == 0) { fclose(f); return false; }
  t.w = atoi(sw.c_str()); t.h = atoi(sh.c_str());
  if (t.w <= 0 || t.h <= 0) { fclose(f); return false; }
  std::vector<uint8_t> raw((size_t)t.w * t.h * ch);
  size_t got = fread(raw.data(), 1, raw.size(), f);
  fclose(f);
  if (got != raw.size()) return false;
  t.rgba.assign((size_t)t.w * t.h * 4, 0);
  for (size_t i = 0; i < (size_t)t.w * t.h; i++) {
    if (ch == 3) { t.rgba[4*i] = raw[3*i]; t.rgba[4*i+1] = raw[3*i+1]; t.rgba[4*i+2] = raw[3*i+2]; }
    else         { t.rgba[4*i] = raw[i]; t.rgba[4*i+1] = raw[i]; t.rgba[4*i+2] = raw[i]; }
  }
  return true;
}

// getppmnum/getppmpaths K:1979-2018: every directory entry whose full path contains
// "ppm" or "PPM".  The reference scans the process cwd; here the directory is a
// parameter and entries are taken in sorted name order (directory order is unspecified).
void scan_textures(Scene& s, const std::string& dir) {
  std::vector<std::string> names;
  if (DIR* d = opendir(dir.c_str())) {
    while (dirent* e = readdir(d)) {
      std::string n = e->d_name;
      if (n == "." || n == "..") continue;
      names.push_back(n);
    }
    closedir(d);
  }
  std::sort(names.begin(), names.end());
  for (auto& n : names) {
    std::string full = dir + "/" + n;
    if (full.find("ppm") != std::string::npos || full.find("PPM") != std::string::npos)
      s.texpaths.push_back(full);
  }
}

// gettexnum K:1172-1183: first path whose lower-cased form contains the (unmodified) query.
int gettexnum(const Scene& s, const std::string& query) {
  for (size_t i = 0; i < s.texpaths.size(); i++) {
    std::string p = s.texpaths[i];
    for (auto& c : p) c = (char)tolower((unsigned char)c);
    if (p.find(query) != std::string::npos) return (int)i;
  }
  return -1;
}

// ------------------------------------------------------------------ .rts reader (K:1113-1530)
struct ParseError { std::string what; };

float to_f(const std::string& s, int line) {   // std::stof
  const char* b = s.c_str(); char* e = nullptr;
  float v = strtof(b, &e);
  if (e == b) throw ParseError{"line " + std::to_string(line) + ": stof: no conversion in '" + s + "'"};
  return v;
}
int to_i(const std::string& s, int line) {     // std::stoi
  const char* b = s.c_str(); char* e = nullptr;
  long v = strtol(b, &e, 10);
  if (e == b) throw ParseError{"line " + std::to_string(line) + ": stoi: no conversion in '" + s + "'"};
  return (int)v;
}

bool read_rts(Scene& s, const std::string& file) {
  // getnum K:1113-1169: count lines not starting with '/' or '*', return count + 1
  {
    std::ifstream in(file);
    if (!in.is_open()) { s.err = "cannot open " + file; return false; }
    std::string line; int n = 0;
    while (std::getline(in, line)) {
      if (line[0] == '/') continue;
      if (line[0] == '*') continue;
      n++;
    }
    s.objs.assign((size_t)n + 1, Obj());
  }
  std::ifstream in(file);
  std::string text; int line = 0, fileline = 0;
  try {
    while (std::getline(in, text)) {          // K:1212
      fileline++;
      int col = 0;
      std::stringstream ss(text);
      if (text[0] == '/') continue;           // K:1218
      if (text[0] == '*') {                   // K:1223-1299
        Settings& g = s.set;
        while (ss.good()) {
          std::string f; std::getline(ss, f, ',');
          switch (col) {
            case 1: g.campos.x = to_f(f, fileline); break;
            case 2: g.campos.y = to_f(f, fileline); break;
            case 3: g.campos.z = to_f(f, fileline); break;
            case 4: g.aperture = to_f(f, fileline); break;
            case 5: g.look.x = to_f(f, fileline); break;
            case 6: g.look.y = to_f(f, fileline); break;
            case 7: g.look.z = to_f(f, fileline); break;
            case 8: g.focus_dist = to_f(f, fileline); break;
            case 9: g.fov = to_i(f, fileline); break;
            case 10: g.max_depth = to_i(f, fileline); break;
            case 11: g.spp = to_i(f, fileline); break;
            case 12: g.background = to_f(f, fileline); break;
            case 13: if (f != "no") g.backtex = gettexnum(s, f); break;
            case 14: g.width = to_i(f, fileline); break;
            case 15: g.height = to_i(f, fileline); break;
            default: break;
          }
          col++;
        }
        continue;
      }
      Obj& o = s.objs[(size_t)line];
      while (ss.good()) {                     // K:1303-1513
        std::string f; std::getline(ss, f, ',');
        if (f == "r") throw ParseError{"line " + std::to_string(fileline) + ": 'r' (time-seeded random, K:1308) is not reproducible"};
        switch (col) {
          case 0: o.pos.x = to_f(f, fileline); break;
          case 1: o.pos.y = to_f(f, fileline); break;
          case 2: o.pos.z = to_f(f, fileline); break;
          case 3: o.type = to_i(f, fileline); break;
          case 4: o.col.x = to_f(f, fileline); break;
          case 5: o.col.y = to_f(f, fileline); break;
          case 6: o.col.z = to_f(f, fileline); break;
          case 7: o.addional.y = to_f(f, fileline); break;
          case 8: o.addional.x = to_f(f, fileline); break;
          case 9: o.dim.x = to_f(f, fileline); break;
          case 10: o.dim.y = to_f(f, fileline); break;
          case 11: o.dim.z = to_f(f, fileline); break;
          case 12: o.mat = to_i(f, fileline); break;
          case 13: o.rot.x = to_f(f, fileline); break;
          case 14: o.rot.y = to_f(f, fileline); break;
          case 15: o.rot.z = to_f(f, fileline); break;
          case 16: o.norm.x = to_f(f, fileline); break;
          case 17: o.norm.y = to_f(f, fileline); break;
          case 18: o.norm.z = to_f(f, fileline); break;
          case 19: o.n1.x = to_f(f, fileline); break;
          case 20: o.n1.y = to_f(f, fileline); break;
          case 21: o.n1.z = to_f(f, fileline); break;
          case 22: o.n2.x = to_f(f, fileline); break;
          case 23: o.n2.y = to_f(f, fileline); break;
          case 24: o.n2.z = to_f(f, fileline); break;
          case 25: o.n3.x = to_f(f, fileline); break;
          case 26: o.n3.y = to_f(f, fileline); break;
          case 27: o.n3.z = to_f(f, fileline); break;
          case 28: o.t1.x = to_f(f, fileline); break;
          case 29: o.t1.y = to_f(f, fileline); break;
          case 30: o.t2.x = to_f(f, fileline); break;
          case 31: o.t2.y = to_f(f, fileline); break;
          case 32: o.t3.x = to_f(f, fileline); break;
          case 33: o.t3.y = to_f(f, fileline); break;
          case 34: if (to_i(f, fileline) == 1) o.smooth = true; break;
          case 35: if (to_i(f, fileline) == 1) o.tex = true; break;
          case 36: if (f != "no") o.texnum = gettexnum(s, f); break;
          case 37: if (f != "no") o.rtexnum = gettexnum(s, f); break;
          default: break;
        }
        col++;
      }
      line++;
    }
  } catch (ParseError& e) { s.err = e.what; return false; }
  s.nanum = line + 1;                          // K:1518
  return true;
}

// ------------------------------------------------------------------ BVH build (K:335-406,1534-1909)
// (the -DORACLE_FMA_KERNEL variant, liboracle_fma.so, contracts a*b+c into fused multiply-adds in the DEVICE functions only -- what nvcc's
// default -fmad=true does to kernel.cu's __device__ code; the BVH build is host code of the reference and stays uncontracted)
#ifdef ORACLE_FMA_KERNEL
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
#endif
struct Builder {
  Scene& s;
  explicit Builder(Scene& sc) : s(sc) {}

  // bounding_box K:335-364: writes nothing for other types (the caller's values persist)
  void bounding_box(int obj, V3& mn, V3& mx) {
    const Obj& o = s.objs[(size_t)obj];
    if (o.type == 0) {
      mn = o.pos - splat(o.dim.x);
      mx = o.pos + splat(o.dim.x);
    } else if (o.type == 2) {
      V3 a = o.pos, b = o.dim, c = o.rot;
      // K:353-354: float fmin/fmax, then "- 0.01" in double, narrowed by make_float3
      mn = v3((float)((double)fminf(a.x, fminf(b.x, c.x)) - 0.01),
              (float)((double)fminf(a.y, fminf(b.y, c.y)) - 0.01),
              (float)((double)fminf(a.z, fminf(b.z, c.z)) - 0.01));
      mx = v3((float)((double)fmaxf(a.x, fmaxf(b.x, c.x)) + 0.01),
              (float)((double)fmaxf(a.y, fmaxf(b.y, c.y)) + 0.01),
              (float)((double)fmaxf(a.z, fmaxf(b.z, c.z)) + 0.01));
    }
  }
  // arraybound K:383-406 (+ surrounding_box K:370-380)
  bool arraybound(V3& mn, V3& mx, const int* objs, int len) {
    if (len == 0) return false;
    V3 tmn = splat(-1), tmx = splat(-1);
    bool first = true;
    for (int g = 0; g < len; g++) {
      bounding_box(objs[g], tmn, tmx);
      if (first) { mn = tmn; mx = tmx; }
      else {
        mn = v3(fminf(mn.x, tmn.x), fminf(mn.y, tmn.y), fminf(mn.z, tmn.z));
        mx = v3(fmaxf(mx.x, tmx.x), fmaxf(mx.y, tmx.y), fmaxf(mx.z, tmx.z));
      }
      first = false;
    }
    return true;
  }
  // calculateSD K:1560-1623, one axis at a time; float accumulators, pow(float,int) in double
  static float sd_axis(const std::vector<V3>& d, int len, int axis) {
    auto at = [&](int i) { return axis == 0 ? d[(size_t)i].x : (axis == 1 ? d[(size_t)i].y : d[(size_t)i].z); };
    float sum = 0.0f, mean, sd = 0.0f;
    for (int i = 0; i < len; i++) sum += at(i);
    mean = sum / len;
    for (int i = 0; i < len; i++) {
      float diff = at(i) - mean;
      sd = (float)((double)sd + sqd((double)diff));   // K:1575  float += double
    }
    return sqrtf(sd / len);                            // K:1580
  }
  // split K:1678-1717 (+ sorto K:1626, pairsort K:1534)
  void split(const int* input, int* a, int* b, int num) {
    std::vector<V3> many((size_t)num);
    std::vector<std::pair<float, int>> pairs((size_t)num);
    for (int o = 0; o < num; o++) many[(size_t)o] = s.objs[(size_t)input[o]].pos;   // K:1686 vertex 0
    float dx = sd_axis(many, num, 0), dy = sd_axis(many, num, 1), dz = sd_axis(many, num, 2);
    int axis = 0;
    float mx = fmaxf(dx, fmaxf(dy, dz));        // K:1634
    if (mx == dx) axis = 0;
    if (mx == dy) axis = 1;
    if (mx == dz) axis = 2;                      // ties -> highest axis
    for (int o = 0; o < num; o++) {
      const V3& p = many[(size_t)o];
      pairs[(size_t)o] = std::make_pair(axis == 0 ? p.x : (axis == 1 ? p.y : p.z), input[o]);
    }
    std::sort(pairs.begin(), pairs.end());       // K:1547
    int part1 = num / 2;
    for (int o = 0; o < part1; o++) a[o] = pairs[(size_t)o].second;
    for (int o = part1; o < num; o++) b[o - part1] = pairs[(size_t)o].second;
  }
  // bvhr K:1745-1861
  void bvhr(int node, int* under) {
    std::vector<Node>& t = s.bvh;
    if (!(t[(size_t)node].active && !t[(size_t)node].end)) return;
    int size = t[(size_t)node].count;
    int p1 = size / 2, p2 = size - p1;
    std::vector<int> a((size_t)p1), b((size_t)p2);
    split(under, a.data(), b.data(), size);
    int an = s.next_node;                        // K:1770
    t[(size_t)an].active = true; t[(size_t)an].end = false;
    for (int e = 0; e < p1; e++) under[e] = a[(size_t)e];
    t[(size_t)an].count = p1;
    if (p1 == 1) { t[(size_t)an].under = under[0]; t[(size_t)an].end = true; }
    arraybound(t[(size_t)an].min, t[(size_t)an].max, a.data(), p1);
    t[(size_t)node].children[0] = an;
    s.next_node++;
    int bn = s.next_node;                        // K:1807
    t[(size_t)bn].active = true;
    for (int e = 0; e < p2; e++) under[e] = b[(size_t)e];
    t[(size_t)bn].end = false;
    t[(size_t)bn].count = p2;
    if (p2 == 1) { t[(size_t)bn].under = under[0]; t[(size_t)bn].end = true; }
    arraybound(t[(size_t)bn].min, t[(size_t)bn].max, b.data(), p2);
    t[(size_t)node].children[1] = bn;
    s.next_node++;
    bvhr(an, a.data());
    bvhr(bn, b.data());
  }
  // build_links K:1720-1742
  void links(int self, int next_right) {
    Node& n = s.bvh[(size_t)self];
    if (!n.end) {
      int c1 = n.children[0], c2 = n.children[1];
      n.hit_node = c1; n.miss_node = next_right;
      links(c1, c2);
      links(c2, next_right);
    } else {
      n.hit_node = next_right; n.miss_node = next_right;
    }
  }
  // build_bvh K:1864-1909
  bool build() {
    int N = s.nanum - 1;
    if (N < 2) { s.err = "BVH build needs at least 2 objects (the reference recurses without bound, K:1756)"; return false; }
    s.bvh.assign((size_t)s.nanum * 2, Node());   // K:2073, K:1873
    s.next_node = 1;
    std::vector<int> under((size_t)s.nanum);
    for (int o = 0; o < s.nanum; o++) under[(size_t)o] = o;
    Node& r = s.bvh[0];
    r.active = true; r.count = s.nanum - 1; r.end = false;
    arraybound(r.min, r.max, under.data(), s.nanum);     // K:1899 includes slot N
    bvhr(0, under.data());
    links(0, -1);
    return true;
  }
};
#ifdef ORACLE_FMA_KERNEL
#pragma GCC pop_options
#endif

// ------------------------------------------------------------------ RNG (cuRAND XORWOW)
struct Xorwow {
  uint32_t v[5]; uint32_t d;
  void init(uint64_t seed) {   // curand_init(seed, 0, 0): _curand_init_scratch, no skip-ahead
    uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    d = 6615241u + t1 + t0;
    v[0] = 123456789u + t0;
    v[1] = 362436069u ^ t0;
    v[2] = 521288629u + t1;
    v[3] = 88675123u ^ t1;
    v[4] = 5783321u + t0;
  }
  uint32_t next() {             // curand(curandStateXORWOW_t*)
    uint32_t t = v[0] ^ (v[0] >> 2);
    v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
    v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
    d += 362437u;
    return v[4] + d;
  }
  double uniform_double() {     // curand_uniform_double: _curand_uniform_double_hq(x, y)
    uint32_t x = next(), y = next();
    uint64_t z = (uint64_t)x ^ ((uint64_t)y << 21);
    return (double)z * 1.1102230246251565e-16 + 5.5511151231257827e-17;  // 2^-53, 2^-54
  }
};

// ------------------------------------------------------------------ counters
struct Counters {     // rays = hit() calls, V node visits, L primitive tests,
  uint64_t rays = 0, V = 0, L = 0, S = 0, T = 0, samples = 0;   // S shading events, T texels
  void add(const Counters& o) { rays += o.rays; V += o.V; L += o.L; S += o.S; T += o.T; samples += o.samples; }
};

// ------------------------------------------------------------------ device functions
// aabb2 K:244-274
inline bool aabb2(V3 o, V3 d, V3 a, V3 b, float& dist) {
  float t_min = 0, t_max = 10000;
  float org[3] = {o.x, o.y, o.z}, dir[3] = {d.x, d.y, d.z};
  float mn[3] = {a.x, a.y, a.z}, mx[3] = {b.x, b.y, b.z};
  for (int k = 0; k < 3; k++) {
    float invD = 1.0f / dir[k];
    float t0 = (mn[k] - org[k]) * invD;
    float t1 = (mx[k] - org[k]) * invD;
    if (invD < 0.0f) { float old = t0; t0 = t1; t1 = old; }
    t_min = t0 > t_min ? t0 : t_min;
    t_max = t1 < t_max ? t1 : t_max;
    if (t_max <= t_min) return false;
  }
  dist = t_min;
  return true;
}

// hit_tri K:277-313; returns t or -1
inline float hit_tri(V3 ro, V3 rd, V3 v0, V3 v1, V3 v2) {
  const float EPS = 0.0001;                      // K:283: float initialised from a double literal
  V3 e1 = v1 - v0, e2 = v2 - v0;
  V3 h = cross(rd, e2);
  float a = dot(e1, h);
  if (a > -EPS && a < EPS) return -1;
  float f = (float)(1.0 / (double)a);            // K:293
  V3 s = ro - v0;
  float u = f * dot(s, h);
  if (u < 0.0 || u > 1.0) return -1;
  V3 q = cross(s, e1);
  float v = f * dot(rd, q);
  if (v < 0.0 || u + v > 1.0) return -1;
  float t = f * dot(e2, q);
  if (t > EPS) return t;
  return -1;
}

// hit_sphere K:316-333; returns (-b - sqrt(disc))/a or -1
inline float hit_sphere(V3 c, float radius, V3 o, V3 d) {
  V3 oc = o - c;
  float a = sqf(length(d));
  float half_b = dot(oc, d);
  float cc = sqf(length(oc)) - radius * radius;
  float disc = half_b * half_b - a * cc;
  if (disc < 0) return -1.0f;
  return (-half_b - sqrtf(disc)) / a;
}

struct HitRec { float t; int idx; };

// singlehit K:432-464
inline HitRec singlehit(const Scene& s, V3 o, V3 d, int x, Counters& c) {
  const Obj& b = s.objs[(size_t)x];
  float dist;
  c.L++;
  if (b.type == 0) dist = hit_sphere(b.pos, b.dim.x, o, d);
  else if (b.type == 2) dist = hit_tri(o, d, b.pos, b.dim, b.rot);
  else dist = -1;                                  // C7
  HitRec r; r.t = -1; r.idx = 0;
  if (dist < 10000.0f && dist > -0.0) { r.t = dist; r.idx = x; }   // K:449
  return r;
}

// hit K:468-512
inline HitRec hit(const Scene& s, V3 o, V3 d, Counters& c) {
  HitRec out; out.t = 10000000.0f; out.idx = 0;
  bool none = true;
  int box = 0;
  c.rays++;
  while (box != -1) {
    const Node& n = s.bvh[(size_t)box];
    float dister = 0;
    c.V++;
    bool h = aabb2(o, d, n.min, n.max, dister);
    if (h && dister < out.t) {
      if (n.end) {
        HitRec tmp = singlehit(s, o, d, n.under, c);
        if ((double)tmp.t > -0.01 && tmp.t < out.t) { out = tmp; none = false; }   // K:488
      }
      box = n.hit_node;
    } else {
      box = n.miss_node;
    }
  }
  if (none) { out.t = -1; out.idx = 0; }
  return out;
}

inline V3 rand_in_unit_sphere(Xorwow& r) {       // K:640-648
  for (;;) {
    float x = (float)(r.uniform_double() * 2 - 1);   // C5: x, then y, then z
    float y = (float)(r.uniform_double() * 2 - 1);
    float z = (float)(r.uniform_double() * 2 - 1);
    V3 p = v3(x, y, z);
    if (sqf(length(p)) >= 1) continue;
    return p;
  }
}
inline float randy(Xorwow& r) { return (float)r.uniform_double(); }   // K:651-662
inline V3 rand_in_unit_disk(Xorwow& r) {         // K:988-994
  for (;;) {
    float x = randy(r) * 2 - 1;
    float y = randy(r) * 2 - 1;
    V3 p = v3(x, y, 0);
    if (sqf(length(p)) >= 1) continue;
    return p;
  }
}
inline V3 reflect(V3 v, V3 n) {                   // K:667-669 (2.0*dot narrows exactly)
  float k = (float)(2.0 * (double)dot(v, n));
  return v - splat(k) * n;
}
inline V3 refract(V3 uv, V3 n, float eta) {       // K:678-683
  float cos_theta = (float)fmin((double)dot(uv * splat(-1), n), 1.0);   // C3
  V3 perp = splat(eta) * (uv + splat(cos_theta) * n);
  float par = (float)(-sqrt(fabs(1.0 - (double)sqf(length(perp)))));
  return perp + splat(par) * n;
}
inline float reflectance(float cosine, float ref_idx) {   // K:686-691, C2
  float r0 = (1 - ref_idx) / (1 + ref_idx);
  r0 = r0 * r0;
  return r0 + (1 - r0) * powi5(1 - cosine);
}

// getnormal K:703-773
inline V3 getnormal(const Scene& s, int obj, V3 origin, V3 hitpoint, V3 dir, V3& texco) {
  const Obj& b = s.objs[(size_t)obj];
  if (b.type == 0) return (hitpoint - b.pos) / splat(b.dim.x);
  if (b.type == 2) {
    V3 v0 = b.pos, v1 = b.dim, v2 = b.rot;
    V3 v0v1 = v1 - v0, v0v2 = v2 - v0;
    V3 N = cross(v0v1, v0v2);
    V3 pvec = cross(dir, v0v2);
    float det = dot(v0v1, pvec);
    float invDet = 1 / det;
    V3 tvec = origin - v0;
    V3 uv;
    uv.x = dot(tvec, pvec) * invDet;
    V3 qvec = cross(tvec, v0v1);
    uv.y = dot(dir, qvec) * invDet;
    uv.z = 1 - uv.x - uv.y;
    texco = splat(uv.z) * b.t1 + splat(uv.x) * b.t2 + splat(uv.y) * b.t3;
    if (b.norm.z != -20) {
      N = b.norm;
      if (b.n1.z != -20 && b.smooth)
        N = splat(uv.z) * b.n1 + splat(uv.x) * b.n2 + splat(uv.y) * b.n3;
    }
    return normalized(N);
  }
  return normalized(hitpoint - b.pos);
}

// checker K:776-784
inline V3 checker(V3 uv, V3 c1, V3 c2) {
  float u2 = floorf(uv.x * 10), v2 = floorf(uv.y * 10);
  float yes = u2 + v2;
  return fmodf(yes, 2.0f) == 0 ? c1 : c2;
}

// tex2D<uchar4>, point / wrap / normalised (K:1959-1964): texel = floor(frac(u) * W)
inline void tex_fetch(const Texture& t, float u, float v, uint8_t out[4], Counters& c) {
  c.T++;
  float fu = u - floorf(u), fv = v - floorf(v);
  int i = f2i(floorf(fu * (float)t.w)), j = f2i(floorf(fv * (float)t.h));
  if (i > t.w - 1) i = t.w - 1;
  if (j > t.h - 1) j = t.h - 1;
  if (i < 0) i = 0;
  if (j < 0) j = 0;
  const uint8_t* p = &t.rgba[((size_t)j * t.w + i) * 4];
  out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3];
}

// raycolor K:787-982
V3 raycolor(const Scene& s, V3 origin, V3 dir, int max_depth, int backtex, float bgint, Xorwow& rng, Counters& c) {
  V3 raydir = dir, rayo = origin, atten = splat(1.0f);
  for (int i = 0; i < max_depth; i++) {
    V3 texco = v3(0, 0, 0);                       // C6
    HitRec h = hit(s, rayo, raydir, c);
    int g = h.idx;
    float t = h.t;
    if (t > 0.0) {
      c.S++;
      V3 hitpoint = rayo + splat(t) * raydir;
      V3 N = getnormal(s, g, rayo, hitpoint, raydir, texco);
      bool front = dot(raydir, N) < 0;             // K:235, 819
      N = front ? N : N * splat(-1);
      const Obj& b = s.objs[(size_t)g];
      V3 ocolor = b.col;
      float rough = b.addional.y;
      if (b.texnum >= 0) {                         // K:829-833
        uint8_t C[4]; tex_fetch(s.tex[(size_t)b.texnum], texco.x, -texco.y + 1, C, c);
        ocolor = v3(float(C[0]) / 255, float(C[1]) / 255, float(C[2]) / 255);
      } else if (b.tex) {
        ocolor = checker(texco, splat((float)0.8), b.col);
      }
      if (b.rtexnum >= 0) {                        // K:840-844
        uint8_t C[4]; tex_fetch(s.tex[(size_t)b.rtexnum], texco.x, -texco.y + 1, C, c);
        rough = float(C[0]) / 255 / 2;
      }
      if (b.mat == 0) {                            // diffuse K:848-866
        V3 target = hitpoint + N;
        if (b.addional.x == 0) target = target + rand_in_unit_sphere(rng);
        else target = target + normalized(rand_in_unit_sphere(rng));
        atten = atten * ocolor;
        rayo = hitpoint;
        raydir = normalized(target - hitpoint);
      } else if (b.mat == 2) {                     // mirror K:867-874
        atten = atten * ocolor;
        rayo = hitpoint;
        raydir = reflect(normalized(raydir), N);
      } else if (b.mat == 3) {                     // metal K:875-883
        V3 refl = reflect(normalized(raydir), N);
        atten = atten * ocolor;
        rayo = hitpoint;
        raydir = refl + splat(rough) * rand_in_unit_sphere(rng);
      } else if (b.mat == 5) {                     // glossy K:884-912
        float r = randy(rng);
        if (r > 0.8) {
          V3 refl = reflect(normalized(raydir), N);
          atten = atten * ocolor;
          rayo = hitpoint;
          raydir = refl + splat(rough) * rand_in_unit_sphere(rng);
        } else {
          V3 target = hitpoint + N;
          target = target + rand_in_unit_sphere(rng);
          atten = atten * ocolor;
          rayo = hitpoint;
          raydir = normalized(target - hitpoint);
        }
      } else if (b.mat == 4) {                     // glass K:914-939
        float ir = b.addional.y;
        float ratio = front ? (float)(1.0 / (double)ir) : ir;
        float cos_theta = (float)fmin((double)dot(normalized(raydir) * splat(-1), N), 1.0);   // C3
        float sin_theta = (float)sqrt(1.0 - (double)(cos_theta * cos_theta));
        bool cannot = (ratio * sin_theta) > 1.0;
        V3 out;
        if (cannot || reflectance(cos_theta, ratio) > randy(rng)) out = reflect(normalized(raydir), N);
        else out = refract(normalized(raydir), N, ratio);
        atten = atten * ocolor;
        rayo = hitpoint;
        raydir = out;
      } else {                                     // emissive K:941-944
        return ocolor * atten;
      }
    } else {
      if (backtex > -1) {                          // env map K:953-966
        V3 u = normalized(raydir);
        float m = (float)(2. * sqrt(sqd((double)u.x) + sqd((double)u.y) + sqd((double)u.z + 1.)));
        V3 tt = u / splat(m) + splat(.5f);
        tt.y = -tt.y;
        uint8_t C[4]; tex_fetch(s.tex[(size_t)backtex], tt.x, -tt.y + 1, C, c);
        V3 col = v3(float(C[0]) / 255, float(C[1]) / 255, float(C[2]) / 255);
        return atten * col * splat(bgint);
      }
      V3 u = normalized(raydir);                   // sky K:971-974
      float t2 = (float)(0.5 * ((double)u.y + 1.0));
      float omt = (float)(1.0 - (double)t2);
      V3 sky = splat(omt) * v3(1.0f, 1.0f, 1.0f) + splat(t2) * v3((float)0.5, (float)0.7, (float)1.0);
      return atten * sky * splat(bgint);
    }
  }
  return v3(0, 0, 0);                              // K:981
}

struct RenderArgs {    // settings[13] K:2581 + W,H + backgroundintensity + frame seed (C8)
  float settings[13];
  int W, H;
  float bgint;
  uint64_t frame_seed;
};

// Kernel K:998-1093 for one pixel
void pixel(const Scene& s, const RenderArgs& a, int x, int y, unsigned stride, int32_t* out, Counters& c, uint32_t* visits = nullptr) {
  const uint64_t v_before = c.V;
  const float* st = a.settings;
  int W = a.W, H = a.H;
  size_t w = (size_t)x * H + y;                    // K:1006
  out[3*w] = out[3*w+1] = out[3*w+2] = 0;
  float aspect = float(W / st[11]) / float(H / st[11]);            // K:1016
  float fov = (float)((double)st[8] * M_PI / 180);                 // K:1020
  float vh = (float)(2.0 * (double)tanf(fov / 2));                 // K:1023, C10
  float vw = aspect * vh;
  V3 from = v3(st[0], st[1], st[2]), at = v3(st[3], st[4], st[5]);
  float focus = st[7];
  V3 vup = v3(0, 1, 0);
  V3 wu = normalized(from - at);
  V3 uu = normalized(cross(vup, wu));
  V3 vu = cross(wu, uu);
  V3 horizontal = splat(focus) * splat(vw) * uu;                   // K:1047
  V3 vertical = splat(focus) * splat(vh) * vu;
  V3 llc = from - horizontal / splat(2) - vertical / splat(2) - splat(focus) * wu;
  float lens_radius = st[6] / 2;
  V3 color = v3(0, 0, 0);
  for (int sidx = 0; sidx < st[10]; ++sidx) {                       // K:1059
    Xorwow rng;
    rng.init(a.frame_seed + (uint64_t)sidx * SPP_SEED_STRIDE + (uint64_t)((unsigned)x + (unsigned)y * stride));  // K:1065, C8
    c.samples++;
    float nu = (float)(((double)float(x) + rng.uniform_double()) / (double)float(W / st[11]));   // K:1067
    float nv = (float)(((double)float(y) + rng.uniform_double()) / (double)float(H / st[11]));
    V3 rd = splat(lens_radius) * rand_in_unit_disk(rng);
    V3 offset = uu * splat(rd.x) + vu * splat(rd.y);
    V3 dir = llc + splat(nu) * horizontal + splat(nv) * vertical - from - offset;
    color = color + raycolor(s, from + offset, dir, f2i(st[9]), f2i(st[12]), a.bgint, rng, c);
  }
  float scale = (float)(1.0 / (double)st[10]);                      // K:1081
  out[3*w]   = f2i(color.x * 255 * scale);
  out[3*w+1] = f2i(color.y * 255 * scale);
  out[3*w+2] = f2i(color.z * 255 * scale);
  if (visits) visits[w] = (uint32_t)(c.V - v_before);   // node visits spent on this pixel (load-balance studies)
}

}  // namespace

// =====================================================================================
// C interface for ctypes (tests / smoke / bench cpu_baseline only)
// =====================================================================================
extern "C" {

struct OrcObj {   // flat mirror of Obj, 42 x 4 bytes
  int32_t type; float pos[3], rot[3], norm[3], n1[3], n2[3], n3[3], t1[3], t2[3], t3[3];
  int32_t smooth, tex, mat; float dim[3], col[3]; int32_t texnum, rtexnum; float addional[3];
};
struct OrcSettings {
  float campos[3], look[3], aperture, focus_dist; int32_t fov, max_depth, spp; float background;
  int32_t backtex, width, height;
};
struct OrcCounters { uint64_t rays, V, L, S, T, samples; };

static thread_local std::string g_err;
const char* orc_last_error() { return g_err.c_str(); }

// texdir == NULL or "" -> no textures are discovered (every name resolves to -1)
void* orc_scene_load(const char* rts_path, const char* texdir) {
  Scene* s = new Scene();
  if (texdir && texdir[0]) {
    scan_textures(*s, texdir);
    for (auto& p : s->texpaths) {
      Texture t;
      if (!load_ppm4(p, t)) { g_err = "cannot load texture " + p; delete s; return nullptr; }
      s->tex.push_back(std::move(t));
    }
  }
  if (!read_rts(*s, rts_path)) { g_err = s->err; delete s; return nullptr; }
  return s;
}
void orc_scene_free(void* h) { delete (Scene*)h; }
int orc_num_objects(void* h) { return ((Scene*)h)->nanum - 1; }
int orc_num_textures(void* h) { return (int)((Scene*)h)->tex.size(); }
int orc_texture_info(void* h, int i, int* w, int* hh) {
  Scene* s = (Scene*)h; if (i < 0 || i >= (int)s->tex.size()) return -1;
  *w = s->tex[(size_t)i].w; *hh = s->tex[(size_t)i].h; return 0;
}
int orc_texture_data(void* h, int i, uint8_t* rgba) {
  Scene* s = (Scene*)h; if (i < 0 || i >= (int)s->tex.size()) return -1;
  memcpy(rgba, s->tex[(size_t)i].rgba.data(), s->tex[(size_t)i].rgba.size()); return 0;
}
void orc_get_settings(void* h, OrcSettings* o) {
  const Settings& g = ((Scene*)h)->set;
  o->campos[0] = g.campos.x; o->campos[1] = g.campos.y; o->campos[2] = g.campos.z;
  o->look[0] = g.look.x; o->look[1] = g.look.y; o->look[2] = g.look.z;
  o->aperture = g.aperture; o->focus_dist = g.focus_dist; o->fov = g.fov; o->max_depth = g.max_depth;
  o->spp = g.spp; o->background = g.background; o->backtex = g.backtex; o->width = g.width; o->height = g.height;
}
// copies N+1 objects (slot N included)
void orc_get_objects(void* h, OrcObj* out) {
  Scene* s = (Scene*)h;
  for (size_t i = 0; i < s->objs.size(); i++) {
    const Obj& b = s->objs[i]; OrcObj& o = out[i];
    auto cp = [](float* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
    o.type = b.type; cp(o.pos, b.pos); cp(o.rot, b.rot); cp(o.norm, b.norm); cp(o.n1, b.n1); cp(o.n2, b.n2); cp(o.n3, b.n3);
    cp(o.t1, b.t1); cp(o.t2, b.t2); cp(o.t3, b.t3); o.smooth = b.smooth; o.tex = b.tex; o.mat = b.mat;
    cp(o.dim, b.dim); cp(o.col, b.col); o.texnum = b.texnum; o.rtexnum = b.rtexnum; cp(o.addional, b.addional);
  }
}
int orc_build_bvh(void* h) {
  Scene* s = (Scene*)h;
  Builder b(*s);
  if (!b.build()) { g_err = s->err; return -1; }
  return 0;
}
int orc_bvh_size(void* h) { return (int)((Scene*)h)->bvh.size(); }     // 2*(N+1)
int orc_bvh_used(void* h) { return ((Scene*)h)->next_node; }          // 2N-1
// arrays of orc_bvh_size() entries; bounds are 3 floats per node
void orc_get_bvh(void* h, int32_t* active, int32_t* child0, int32_t* child1, int32_t* count, int32_t* hitn,
                 int32_t* missn, int32_t* under, int32_t* end, float* mn, float* mx) {
  Scene* s = (Scene*)h;
  for (size_t i = 0; i < s->bvh.size(); i++) {
    const Node& n = s->bvh[i];
    active[i] = n.active; child0[i] = n.children[0]; child1[i] = n.children[1]; count[i] = n.count;
    hitn[i] = n.hit_node; missn[i] = n.miss_node; under[i] = n.under; end[i] = n.end;
    mn[3*i] = n.min.x; mn[3*i+1] = n.min.y; mn[3*i+2] = n.min.z;
    mx[3*i] = n.max.x; mx[3*i+1] = n.max.y; mx[3*i+2] = n.max.z;
  }
}

// One CudaStarter call (K:2562-2669) on the CPU.  out = int32[W*H*3], index (x*H + y)*3.
// Pixels are rendered for block columns bx with bx % col_mod == col_rem (col_mod = 1 -> all);
// everything else in `out` is zero (C9).  nthreads splits the block columns over std::threads.
static int render_impl(void* h, const float* settings13, int W, int H, float bgint, uint64_t frame_seed,
                       int32_t* out, OrcCounters* counters, int nthreads, int col_mod, int col_rem, uint32_t* visits) {
  Scene* s = (Scene*)h;
  if (s->bvh.empty()) { g_err = "BVH not built"; return -1; }
  RenderArgs a; memcpy(a.settings, settings13, sizeof(a.settings));
  a.W = W; a.H = H; a.bgint = bgint; a.frame_seed = frame_seed;
  int div = f2i(a.settings[11]);
  if (div < 1) { g_err = "divisor < 1"; return -1; }
  int bt = f2i(a.settings[12]);
  if (bt >= (int)s->tex.size()) { g_err = "backtex out of range"; return -1; }
  int gx = W / div / 8, gy = H / div / 8;          // K:2636 numBlocks
  unsigned stride = 8u * (unsigned)gx;             // blockDim.x * gridDim.x
  memset(out, 0, sizeof(int32_t) * 3 * (size_t)W * H);
  if (nthreads < 1) nthreads = 1;
  if (col_mod < 1) col_mod = 1;
  std::vector<Counters> cs((size_t)nthreads);
  std::atomic<int> next(0);
  auto work = [&](int tid) {
    Counters c;
    for (;;) {
      int bx = next.fetch_add(1);
      if (bx >= gx) break;
      if (bx % col_mod != col_rem) continue;
      for (int by = 0; by < gy; by++)
        for (int tx = 0; tx < 8; tx++)
          for (int ty = 0; ty < 8; ty++)
            pixel(*s, a, bx * 8 + tx, by * 8 + ty, stride, out, c, visits);
    }
    cs[(size_t)tid] = c;
  };
  if (nthreads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
    for (auto& t : th) t.join();
  }
  Counters tot; for (auto& c : cs) tot.add(c);
  if (counters) { counters->rays = tot.rays; counters->V = tot.V; counters->L = tot.L; counters->S = tot.S; counters->T = tot.T; counters->samples = tot.samples; }
  return 0;
}

int orc_render(void* h, const float* settings13, int W, int H, float bgint, uint64_t frame_seed,
               int32_t* out, OrcCounters* counters, int nthreads, int col_mod, int col_rem) {
  return render_impl(h, settings13, W, H, bgint, frame_seed, out, counters, nthreads, col_mod, col_rem, nullptr);
}
// same, and visits[x*H + y] = BVH node visits spent on that pixel (must be zero-filled by the caller)
int orc_render_visits(void* h, const float* settings13, int W, int H, float bgint, uint64_t frame_seed,
                      int32_t* out, OrcCounters* counters, int nthreads, int col_mod, int col_rem, uint32_t* visits) {
  return render_impl(h, settings13, W, H, bgint, frame_seed, out, counters, nthreads, col_mod, col_rem, visits);
}

// ---- known-answer entry points (per-function checks of the HIP device code) ----
void orc_kat_rng(uint64_t seed, int n, double* out) {
  Xorwow r; r.init(seed);
  for (int i = 0; i < n; i++) out[i] = r.uniform_double();
}
void orc_kat_rng_u32(uint64_t seed, int n, uint32_t* out) {
  Xorwow r; r.init(seed);
  for (int i = 0; i < n; i++) out[i] = r.next();
}
// rays: o[3n], d[3n]; boxes mn[3n], mx[3n] -> hit[n], dist[n] (dist = 0 when missed)
void orc_kat_aabb(int n, const float* o, const float* d, const float* mn, const float* mx, int32_t* hitf, float* dist) {
  for (int i = 0; i < n; i++) {
    float t = 0;
    bool h = aabb2(v3(o[3*i], o[3*i+1], o[3*i+2]), v3(d[3*i], d[3*i+1], d[3*i+2]),
                   v3(mn[3*i], mn[3*i+1], mn[3*i+2]), v3(mx[3*i], mx[3*i+1], mx[3*i+2]), t);
    hitf[i] = h; dist[i] = h ? t : 0;
  }
}
void orc_kat_tri(int n, const float* o, const float* d, const float* v0, const float* v1, const float* v2, float* t) {
  for (int i = 0; i < n; i++)
    t[i] = hit_tri(v3(o[3*i], o[3*i+1], o[3*i+2]), v3(d[3*i], d[3*i+1], d[3*i+2]), v3(v0[3*i], v0[3*i+1], v0[3*i+2]),
                   v3(v1[3*i], v1[3*i+1], v1[3*i+2]), v3(v2[3*i], v2[3*i+1], v2[3*i+2]));
}
void orc_kat_sphere(int n, const float* o, const float* d, const float* c, const float* r, float* t) {
  for (int i = 0; i < n; i++)
    t[i] = hit_sphere(v3(c[3*i], c[3*i+1], c[3*i+2]), r[i], v3(o[3*i], o[3*i+1], o[3*i+2]), v3(d[3*i], d[3*i+1], d[3*i+2]));
}
// closest-hit queries against the scene's BVH: t[n] (-1 = miss), idx[n]
void orc_kat_hit(void* h, int n, const float* o, const float* d, float* t, int32_t* idx) {
  Scene* s = (Scene*)h; Counters c;
  for (int i = 0; i < n; i++) {
    HitRec r = hit(*s, v3(o[3*i], o[3*i+1], o[3*i+2]), v3(d[3*i], d[3*i+1], d[3*i+2]), c);
    t[i] = r.t; idx[i] = r.idx;
  }
}
// scatter helpers: in v[3n], nrm[3n], eta[n] -> reflect[3n], refract[3n], schlick[n] (cosine = v.x)
// getnormal K:703-773 as called from raycolor K:806-808: hitpoint = origin + t * dir, texco starts at 0 (C6)
void orc_kat_normal(void* h, int n, const int32_t* obj, const float* o, const float* d, const float* t, float* nrm, float* texco) {
  const Scene& s = *(Scene*)h;
  for (int i = 0; i < n; i++) {
    V3 ro = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    V3 hp = ro + splat(t[i]) * rd;
    V3 tc = v3(0, 0, 0);
    V3 N = getnormal(s, obj[i], ro, hp, rd, tc);
    nrm[3 * i] = N.x; nrm[3 * i + 1] = N.y; nrm[3 * i + 2] = N.z;
    texco[3 * i] = tc.x; texco[3 * i + 1] = tc.y; texco[3 * i + 2] = tc.z;
  }
}

void orc_kat_optics(int n, const float* v, const float* nrm, const float* eta, float* refl, float* refr, float* schlick) {
  for (int i = 0; i < n; i++) {
    V3 a = v3(v[3*i], v[3*i+1], v[3*i+2]), b = v3(nrm[3*i], nrm[3*i+1], nrm[3*i+2]);
    V3 r1 = reflect(a, b), r2 = refract(a, b, eta[i]);
    refl[3*i] = r1.x; refl[3*i+1] = r1.y; refl[3*i+2] = r1.z;
    refr[3*i] = r2.x; refr[3*i+1] = r2.y; refr[3*i+2] = r2.z;
    schlick[i] = reflectance(a.x, eta[i]);
  }
}

}  // extern "C"
