// dogeray -- headless host application over the C ABI (include/dogeray_amd.h).
//
// Mirrors the reference's main() (kernel.cu K:2021-2557) without SDL2 (absent in this image):
//   argv[1] or scene.rts                                   K:2041-2051
//   getnum/read/readtextures/build_bvh                     K:2055-2094  -> dr_scene_load / dr_scene_build_bvh
//   present loop: 1/8, 1/4, 1/2, 1/1 previews then full-res frames accumulated,
//   display value clamp(sum / (iter - pnum), 0, 255)       K:2154-2322  -> dr_render_accumulate / dr_accum_present
//   status line "Time = ..[us]  .. FPS  .. samples"        K:2327
//   SPACE -> <scene>.bmp                                   K:2501-2516  -> --out FILE (.bmp or .ppm) at exit
// Everything that touches the GPU goes through dr_* calls; this file includes no HIP header.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/stat.h>

#include "../../include/dogeray_amd.h"

namespace {

void die(const char* what) {
  fprintf(stderr, "dogeray: %s: %s\n", what, dr_last_error());
  exit(1);
}

void pack13(const dr_settings& s, int divisor, int spp, int depth, float out[13]) {   // K:2581
  const float v[13] = {s.campos[0], s.campos[1], s.campos[2], s.look[0], s.look[1], s.look[2], s.aperture, s.focus_dist,
                       (float)s.fov, (float)depth, (float)spp, (float)divisor, (float)s.backtex};
  memcpy(out, v, sizeof(v));
}

bool ends_with(const std::string& s, const char* suf) {
  size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

bool write_ppm(const std::string& path, const std::vector<uint8_t>& rgb, int W, int H) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  fprintf(f, "P6\n%d %d\n255\n", W, H);
  bool ok = fwrite(rgb.data(), 1, rgb.size(), f) == rgb.size();
  fclose(f);
  return ok;
}

// The file the reference's export key writes (SDL_SaveBMP of an ARGB8888 surface, K:2505-2513), byte for byte in
// layout: BITMAPV4HEADER (108 bytes), 32 bits per pixel, BI_BITFIELDS with masks R 00ff0000 G 0000ff00 B 000000ff
// A ff000000, colour space 'Win ', rows bottom-up, alpha 255 -- checked against images/eorovan.blend.rts.bmp.
bool write_bmp(const std::string& path, const std::vector<uint8_t>& rgb, int W, int H) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const uint32_t row = (uint32_t)W * 4, size = row * (uint32_t)H;
  uint8_t hdr[122] = {'B', 'M'};
  auto put32 = [&](int off, uint32_t v) { memcpy(hdr + off, &v, 4); };
  auto put16 = [&](int off, uint16_t v) { memcpy(hdr + off, &v, 2); };
  put32(2, 122 + size); put32(10, 122); put32(14, 108); put32(18, (uint32_t)W); put32(22, (uint32_t)H);
  put16(26, 1); put16(28, 32); put32(30, 3); put32(34, size);
  put32(54, 0x00ff0000u); put32(58, 0x0000ff00u); put32(62, 0x000000ffu); put32(66, 0xff000000u);
  put32(70, 0x57696e20u);                                    // LCS_WINDOWS_COLOR_SPACE, "Win "
  bool ok = fwrite(hdr, 1, sizeof(hdr), f) == sizeof(hdr);
  std::vector<uint8_t> line(row, 0);
  for (int y = H - 1; y >= 0 && ok; y--) {
    const uint8_t* src = &rgb[(size_t)y * W * 3];
    for (int x = 0; x < W; x++) {
      line[(size_t)x * 4] = src[(size_t)x * 3 + 2]; line[(size_t)x * 4 + 1] = src[(size_t)x * 3 + 1];
      line[(size_t)x * 4 + 2] = src[(size_t)x * 3]; line[(size_t)x * 4 + 3] = 255;
    }
    ok = fwrite(line.data(), 1, row, f) == row;
  }
  fclose(f);
  return ok;
}

void usage() {
  fprintf(stderr,
          "usage: dogeray [scene.rts] [--textures DIR] [--frames N] [--out FILE.bmp|.ppm] [--width W] [--height H]\n"
          "               [--spp S] [--depth D] [--seed N] [--device I] [--gpus N] [--group G] [--gather-every K] [--cache] [--quiet]\n"
          "  scene        .rts file (default scene.rts, as the reference)\n"
          "  --textures   directory scanned for *ppm* textures (default: current directory, as the reference)\n"
          "  --frames     full-resolution frames to accumulate after the 4 preview stages (default 64)\n"
          "  --group      frames rendered between two presents once accumulating (default 8)\n"
          "  --gpus       N > 1: GPUs 0..N-1 of this node render interleaved 8-pixel block columns of every frame (one context and one\n"
          "               host thread each); the stripes are gathered to GPU 0 over RCCL every --gather-every frames (default: once per\n"
          "               present), the gather of one batch running beside the rendering of the next\n"
          "  --cache      keep a binary image of the parsed scene + BVH next to the scene (scene.rtsb) and start from\n"
          "               it while it is newer than the .rts ('r' fields and textures are frozen in it: delete it to redraw)\n");
}

}  // namespace

int main(int argc, char** argv) {
  std::string scene_path = "scene.rts", out_path;
  const char* texdir = nullptr;
  int frames = 64, device = 0, group = 8, width = 0, height = 0, spp = 0, depth = 0, gpus = 0, gather_every = 0;
  uint64_t seed = 1;
  bool quiet = false, have_scene = false, use_cache = false;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
    if (a == "--textures") texdir = next();
    else if (a == "--frames") frames = atoi(next());
    else if (a == "--out") out_path = next();
    else if (a == "--width") width = atoi(next());
    else if (a == "--height") height = atoi(next());
    else if (a == "--spp") spp = atoi(next());
    else if (a == "--depth") depth = atoi(next());
    else if (a == "--seed") seed = strtoull(next(), nullptr, 10);
    else if (a == "--device") device = atoi(next());
    else if (a == "--group") group = atoi(next());
    else if (a == "--gpus") gpus = atoi(next());
    else if (a == "--gather-every") gather_every = atoi(next());
    else if (a == "--quiet") quiet = true;
    else if (a == "--cache") use_cache = true;
    else if (a == "-h" || a == "--help") { usage(); return 0; }
    else if (!have_scene && a[0] != '-') { scene_path = a; have_scene = true; }
    else { usage(); return 2; }
  }
  if (group < 1) group = 1;

  if (!quiet) printf("DOGERAY render path on MI355X (dogeray_amd, C ABI v%d)\n", dr_abi_version());
  printf("%s%s\n", have_scene ? "Opening:" : "Opening Default Scene: ", scene_path.c_str());   // K:2045-2050
  dr_scene* scene = nullptr;
  const std::string cache_path = scene_path + "b";           // x.rts -> x.rtsb
  bool from_cache = false;
  if (use_cache) {
    struct stat st_rts, st_bin;
    if (stat(scene_path.c_str(), &st_rts) == 0 && stat(cache_path.c_str(), &st_bin) == 0 && st_bin.st_mtime >= st_rts.st_mtime) {
      if (dr_scene_load_binary(cache_path.c_str(), &scene) == DR_OK) from_cache = true;
      else fprintf(stderr, "ignoring %s: %s\n", cache_path.c_str(), dr_last_error());
    }
  }
  if (!from_cache && dr_scene_load(scene_path.c_str(), texdir, &scene) != DR_OK) die("cannot load scene");
  if (from_cache && !quiet) printf("(scene taken from %s)\n", cache_path.c_str());
  printf("%d tris\n%d textures total\n", dr_scene_num_objects(scene) + 1, dr_scene_num_textures(scene));   // K:2056,1995 (objnum = count + 1)
  dr_settings s;
  dr_scene_get_settings(scene, &s);
  if (width > 0) s.width = width;
  if (height > 0) s.height = height;
  if (spp > 0) s.spp = spp;
  if (depth > 0) s.max_depth = depth;
  printf("Building BVH..\n");
  auto t0 = std::chrono::steady_clock::now();
  if (dr_scene_bvh_size(scene) == 0 && dr_scene_build_bvh(scene, 0) != DR_OK) die("cannot build the BVH");
  if (use_cache && !from_cache && dr_scene_save_binary(scene, cache_path.c_str()) != DR_OK) fprintf(stderr, "cannot write %s: %s\n", cache_path.c_str(), dr_last_error());
  double bvh_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  printf("Done!\n%d nodes total (%.0f ms)\n", dr_scene_bvh_size(scene), bvh_ms);               // K:2093-2094

  // one GPU: a context; several: a group (rank 0 = GPU 0 assembles and presents)
  dr_context* ctx = nullptr;
  dr_group* grp = nullptr;
  const int W = s.width, H = s.height;
  if (gpus >= 1) {                       // --gpus given (even 1): the group path
    if (dr_group_create(gpus, nullptr, &grp) != DR_OK) die("cannot create the GPU group");
    if (dr_group_upload_scene(grp, scene) != DR_OK) die("cannot upload the scene");
    if (dr_group_accum_reset(grp, W, H) != DR_OK) die("cannot allocate the accumulators");
    ctx = dr_group_context(grp, 0);
    if (!quiet) printf("%d GPUs, stripes gathered to GPU 0 by %s\n", gpus, dr_group_uses_rccl(grp) ? "RCCL send/recv" : "peer copies");
  } else {
    if (dr_context_create(device, &ctx) != DR_OK) die("cannot create the device context");
    if (dr_context_upload_scene(ctx, scene) != DR_OK) die("cannot upload the scene");
    if (dr_accum_reset(ctx, W, H) != DR_OK) die("cannot allocate the accumulator");
  }
  auto reset = [&]() { return grp ? dr_group_accum_reset(grp, W, H) : dr_accum_reset(ctx, W, H); };
  auto render = [&](const float* st13, uint64_t sd, int n) {
    return grp ? dr_group_render_accumulate(grp, st13, W, H, s.background, sd, 1000003, n, gather_every)
               : dr_render_accumulate(ctx, st13, W, H, s.background, sd, 1000003, n);
  };

  // ---- the present loop (K:2154-2224)
  int iter = 0;
  uint64_t frame_no = 0;
  const uint64_t seed_stride = 1000003;
  std::vector<uint8_t> rgb((size_t)W * H * 3);
  int divide_by = 1;
  const int total_iters = 4 + frames;
  while (iter < total_iters) {
    auto begin = std::chrono::steady_clock::now();
    float st[13];
    int pnum, n = 1;
    if (iter < 4) {
      static const int ladder[4] = {8, 4, 2, 1};
      // iter 0 runs with the file's spp/depth, iters 1-3 with spp 1 / depth 2 (K:2171-2204)
      pack13(s, ladder[iter], iter == 0 ? s.spp : 1, iter == 0 ? s.max_depth : 2, st);
      if (reset() != DR_OK) die("reset");                            // CudaStarter overwrites outr on these calls
      pnum = iter;
    } else if (!grp && group == 1) {
      // one frame per present, as the reference does it -- but pipelined: frame k + 1 is queued before frame k's image is waited for, so
      // its launch starts while frame k's slowest pixels drain; every image is still exactly clamp(sum of the frames so far / count)
      pack13(s, 1, s.spp, s.max_depth, st);
      // (frames submitted one after the other are rendered in groups -- one launch for up to pipe_group frames, each into its own buffer -- and added and
      // shown one by one: two groups' worth of frames are kept in flight)
      int group_opt = 1;
      (void)dr_context_get_option(ctx, "pipe_group", &group_opt);
      const size_t window = (size_t)(2 * (group_opt > 1 ? group_opt : 1));
      std::vector<uint64_t> tickets; std::vector<int> divs;
      auto t_prev = std::chrono::steady_clock::now();
      while (iter < total_iters || !tickets.empty()) {
        if (iter < total_iters && tickets.size() < window) {
          iter++;
          uint64_t t = 0;
          if (dr_pipeline_submit(ctx, st, W, H, s.background, seed + frame_no * seed_stride, iter - 3, &t) != DR_OK) die("submit");      // K:2287
          tickets.push_back(t); divs.push_back(iter - 3);
          frame_no++;
          if (iter < total_iters && tickets.size() < window) continue;
        }
        if (dr_pipeline_wait(ctx, tickets[0], rgb.data()) != DR_OK) die("present");
        divide_by = divs[0];
        tickets.erase(tickets.begin()); divs.erase(divs.begin());
        auto now = std::chrono::steady_clock::now();
        long long us = std::chrono::duration_cast<std::chrono::microseconds>(now - t_prev).count();
        t_prev = now;
        if (!quiet) {
          printf("\rTime = %lld[us]  %.2f FPS       %d samples             ", us, us > 0 ? 1e6 / (double)us : 0.0, divide_by * s.spp);   // K:2327
          fflush(stdout);
        }
      }
      break;
    } else {
      pack13(s, 1, s.spp, s.max_depth, st);
      n = total_iters - iter < group ? total_iters - iter : group;   // several frames per present
      pnum = 3;
    }
    if (render(st, seed + frame_no * seed_stride, n) != DR_OK) die("render");
    frame_no += (uint64_t)n;
    iter += n;
    divide_by = iter - pnum;                                          // K:2287
    if (dr_accum_present(ctx, divide_by, rgb.data()) != DR_OK) die("present");
    long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - begin).count();
    if (!quiet) {
      printf("\rTime = %lld[us]  %.2f FPS       %d samples             ", us, us > 0 ? 1e6 * n / (double)us : 0.0, divide_by * s.spp);   // K:2327
      fflush(stdout);
    }
  }
  if (!quiet) printf("\n");

  dr_stats stats;
  if (dr_stats_get(ctx, &stats) == DR_OK && stats.frames > 0)
    printf("rendered %llu frames in %llu launches, %.3f ms of kernel time per frame\n", (unsigned long long)stats.frames,
           (unsigned long long)stats.launches, stats.kernel_ms / (double)stats.frames);
  if (!out_path.empty()) {
    bool ok = ends_with(out_path, ".ppm") ? write_ppm(out_path, rgb, W, H) : write_bmp(out_path, rgb, W, H);
    if (!ok) { fprintf(stderr, "dogeray: cannot write %s\n", out_path.c_str()); return 1; }
    printf("exported image:%s\n", out_path.c_str());                  // K:2515
  }
  if (grp) dr_group_destroy(grp); else dr_context_destroy(ctx);
  dr_scene_free(scene);
  return 0;
}
