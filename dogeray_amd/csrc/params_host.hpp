// settings[13] -> the per-launch constants of RenderParams that do not depend on the device: the camera block of Kernel
// (kernel.cu K:1016-1052, identical for every pixel, so evaluated once on the host with the reference's float / double promotions),
// the sample scale, the block grid and the stripe.  Host only; shared by context.cpp and the host build of the kernel arithmetic
// (tools/host_kernel.cpp).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "device_layout.h"

namespace dr {

struct V3h { float x, y, z; };
inline V3h hsub(V3h a, V3h b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3h hmul(V3h a, V3h b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3h hdiv(V3h a, V3h b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3h hsplat(float a) { return {a, a, a}; }
inline float hdot(V3h a, V3h b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3h hcross(V3h a, V3h b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3h hnorm(V3h v) { float inv = 1.0f / sqrtf(hdot(v, v)); return {v.x * inv, v.y * inv, v.z * inv}; }
inline void st3(float* d, V3h v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline int hf2i(float f) {       // float -> int as CUDA's cvt.rzi.s32.f32: saturating, NaN -> 0
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
}

// Fills the camera, sampling and grid fields of P (everything else is left as it is).  Returns null, or what is wrong with the arguments.
inline const char* fill_view_params(const float* st, int W, int H, float background, uint64_t seed, int stripe_mod, int stripe_rem, RenderParams& P) {
  if (W <= 0 || H <= 0 || W > 65000 || H > 65000 || (size_t)W * (size_t)H > (size_t)1 << 28) return "bad frame size";      // (x and y share a word in the phase stash)
  const int div = hf2i(st[11]);
  if (div < 1) return "divisor must be >= 1";
  float aspect = float(W / st[11]) / float(H / st[11]);           // K:1016 (int / float)
  float fov = (float)((double)st[8] * M_PI / 180);                // K:1020
  float vh = (float)(2.0 * (double)tanf(fov / 2));                // K:1023
  float vw = aspect * vh;
  V3h from = {st[0], st[1], st[2]}, at = {st[3], st[4], st[5]};
  float focus = st[7];
  V3h vup = {0, 1, 0};
  V3h wu = hnorm(hsub(from, at));
  V3h uu = hnorm(hcross(vup, wu));
  V3h vu = hcross(wu, uu);
  V3h hor = hmul(hmul(hsplat(focus), hsplat(vw)), uu);            // K:1047
  V3h ver = hmul(hmul(hsplat(focus), hsplat(vh)), vu);
  V3h llc = hsub(hsub(hsub(from, hdiv(hor, hsplat(2))), hdiv(ver, hsplat(2))), hmul(hsplat(focus), wu));
  st3(P.from, from); st3(P.llc, llc); st3(P.hor, hor); st3(P.ver, ver); st3(P.uu, uu); st3(P.vu, vu);
  P.lens_radius = st[6] / 2;                                      // K:1052
  P.bgint = background;
  P.spp_f = st[10];
  P.scale = (float)(1.0 / (double)st[10]);                        // K:1081
  P.den_w = (double)float(W / st[11]);                            // K:1067
  P.den_h = (double)float(H / st[11]);
  P.seed = seed;
  P.W = W; P.H = H;
  P.gx = W / div / 8; P.gy = H / div / 8;                         // K:2636
  P.stripe_mod = stripe_mod; P.stripe_rem = stripe_rem;
  P.ncols = P.gx > stripe_rem ? (P.gx - stripe_rem + stripe_mod - 1) / stripe_mod : 0;
  P.seed_stride = 8u * (unsigned)P.gx;                            // blockDim.x * gridDim.x, K:1065
  P.max_depth = hf2i(st[9]);
  P.backtex = hf2i(st[12]);
  P.batch = 1;
  P.batch_seed_stride = 0;
  return nullptr;
}

}  // namespace dr
