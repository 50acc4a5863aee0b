// Microbenchmark: how fast MI355X serves DIVERGENT, DEPENDENT fetches of small records (a tree walk: every lane reads a
// record at an address computed from the previous one) -- the rate that bounds the node loop.
// hipcc --offload-arch=gfx950 -O3 -o gather tools/gather_bench.hip && ./gather <array MB>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NLOAD, int STRIDE, int ACTIVE_MOD, int OCC = 5>
__global__ __launch_bounds__(256, OCC) void gather(const void* base, unsigned bytes, int iters, unsigned* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  unsigned x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  unsigned acc = 0;
  const unsigned nrec = bytes / 128;
  const bool active = (threadIdx.x % ACTIVE_MOD) == 0;
  for (int i = 0; i < iters; i++) {
    // next address depends on the loaded data (a dependent chain per lane, like a tree walk)
    unsigned off = (x % nrec) * 128u + (NLOAD > 4 ? 0u : ((x >> 27) & 3u) * 32u);     // a 32-byte record at 32-byte alignment (a whole line for the wide records)
    u32x4 v[NLOAD];
    if (active) {
#pragma unroll
      for (int k = 0; k < NLOAD; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off + k * STRIDE), 0, 0);
#pragma unroll
      for (int k = 0; k < NLOAD; k++) acc += v[k].x ^ v[k].w;
    }
    x = x * 1664525u + 1013904223u + (acc & 1u);
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int NLOAD, int STRIDE, int ACTIVE_MOD, int OCC = 5>
void run(const char* name, void* d, unsigned bytes, unsigned* out) {
  const int iters = 2000, blocks = 256 * OCC;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  gather<NLOAD, STRIDE, ACTIVE_MOD, OCC><<<blocks, 256>>>(d, bytes, 200, out);
  hipEventRecord(e0);
  gather<NLOAD, STRIDE, ACTIVE_MOD, OCC><<<blocks, 256>>>(d, bytes, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double steps = (double)blocks * 4 * iters;                       // wave-steps
  double lanes = 64.0 / ACTIVE_MOD;
  printf("%-46s %7.3f ms  %6.1f ns per wave-step and CU-slot  %6.2f G lane-steps/s  %6.2f G lane-requests/s (%.2f per clock and CU at 2.4 GHz)\n", name, ms,
         ms * 1e6 / (steps / 256.0), steps * lanes / ms / 1e6, steps * lanes * NLOAD / ms / 1e6, steps * lanes * NLOAD / ms / 1e6 / 256 / 2.4);
}
int main(int argc, char** argv) {
  unsigned mb = argc > 1 ? atoi(argv[1]) : 96;
  unsigned bytes = mb << 20;
  void* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
  unsigned* out; hipMalloc(&out, 4);
  printf("array %u MB, 5 waves/SIMD, dependent chain per lane\n", mb);
  run<1, 16, 1>("1 x 16 B per lane", d, bytes, out);
  run<2, 16, 1>("2 x 16 B per lane (one 32-byte record)", d, bytes, out);
  run<4, 16, 1>("4 x 16 B per lane (64 contiguous bytes)", d, bytes, out);
  run<2, 64, 1>("2 x 16 B per lane, 64 bytes apart", d, bytes, out);
  run<2, 16, 2>("2 x 16 B, every 2nd lane", d, bytes, out);
  run<2, 16, 4>("2 x 16 B, every 4th lane", d, bytes, out);
  run<2, 16, 16>("2 x 16 B, every 16th lane", d, bytes, out);
  run<4, 16, 4>("4 x 16 B, every 4th lane", d, bytes, out);
  run<2, 16, 1, 2>("2 x 16 B per lane, 2 waves/SIMD", d, bytes, out);
  run<2, 16, 1, 4>("2 x 16 B per lane, 4 waves/SIMD", d, bytes, out);
  run<2, 16, 1, 8>("2 x 16 B per lane, 8 waves/SIMD", d, bytes, out);
  run<5, 16, 1, 5>("5 x 16 B per lane (80 B), 5 waves/SIMD", d, bytes, out);
  run<8, 16, 1, 5>("8 x 16 B per lane (128 B), 5 waves/SIMD", d, bytes, out);
  return 0;
}
