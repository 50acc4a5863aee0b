// A short launch of the persistent kernel as a chain of kernels (persistent_kernel.hpp, template parameter HAND): the lean six-wave build renders from the
// tile queues until they are empty and leaves its live paths in a list; the work-sharing build picks them up, spread evenly over all its waves, and
// finishes them with helpers from the first step on.  Registers are allocated per kernel, so the bulk runs at the lean build's rate and only the tail pays
// for work sharing.  (The reference renders one frame per CudaStarter call, kernel.cu K:2154-2224, K:2634-2640: every launch of its present loop is short.)
#include <hip/hip_runtime.h>

#include "device_core.hpp"
#include "kernels.hpp"
#include "persistent_kernel.hpp"
#include "../../include/dogeray_amd.h"

namespace dr {

int launch_handoff_chain(hipStream_t stream, const RenderParams& P_in, const PersistentCfg& cfg, unsigned* tile_counter, const int* order,
                         const int* region_start, unsigned* pixel_cost, const HandoffPlan& plan) {
  RenderParams P = P_in;
  const int* rstart = order ? region_start : nullptr;        // identity order: the split travels in P.region_start
  const long long work = (long long)P.ncols * P.gy * P.batch;
  int blocks = cfg.num_cus * (cfg.wgs_per_cu > 0 && cfg.wgs_per_cu < 6 ? cfg.wgs_per_cu : 6);      // stage 0: the lean build, six waves per SIMD (five beside the other half of a duo launch)
  if ((long long)blocks * 4 > work) blocks = (int)((work + 3) / 4);
  const int waves0 = blocks * 4;
  const bool log = P.wave_log != nullptr && 2 * waves0 <= WAVE_LOG_STAGE_WAVES;
  unsigned long long* const log_base = log ? P.wave_log : nullptr;
  int stage = 0;
  auto stage_log = [&]() { return log_base ? log_base + (size_t)stage * WAVE_LOG_STAGE_WAVES * 16 : nullptr; };
  Handoff hf;
  hf.ctl = plan.ctl; hf.cap = plan.cap; hf.flags = plan.flags; hf.epoch = plan.epoch;
  // ---- stage 0: tiles -> list 0
  hf.in = nullptr; hf.out = plan.list[0]; hf.in_count = 0; hf.out_count = 1;
  hf.wait = plan.wait;
  hf.margin = (waves0 + P.regions - 1) / P.regions;          // queue positions: the last tile of every wave, spread over the queues
  hf.resident_waves = cfg.num_cus * 6 * 4;
  P.wave_log = stage_log();
  hipLaunchKernelGGL((render_handoff_kernel<6, 32, 20, 2, false, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, P, tile_counter, order, rstart, pixel_cost, hf);
  stage++;
  // every later stage starts twice as many waves as stage 0 had -- a wave of stage 0 leaves at most 64 paths and 63 pixels of its last tile behind, so 64
  // entries per wave always suffice --; those beyond the resident ones end at once unless the list is that long
  const dim3 grid2((unsigned)(2 * blocks));
  int cur = 0;
  if (plan.mid != 0) {
    // ---- a middle stage: no tiles, no sharing; runs mid_wait iterations (the short paths end) and dumps what is left
    RenderParams Pm = P;
    Pm.coop_steps = 0;
    Pm.wave_log = stage_log();
    hf.in = plan.list[cur]; hf.in_count = 1 + cur; hf.out = plan.list[cur ^ 1]; hf.out_count = 2 + cur;
    hf.wait = plan.mid_wait; hf.margin = 0;
    if (plan.mid == 1) {
      hf.resident_waves = cfg.num_cus * 6 * 4;
      hipLaunchKernelGGL((render_handoff_kernel<6, 32, 20, 2, false, 2>), grid2, dim3(256), 0, stream, Pm, tile_counter, order, rstart, pixel_cost, hf);
    } else {
      hf.resident_waves = cfg.num_cus * 5 * 4;
      hipLaunchKernelGGL((render_handoff_kernel<5, 32, 20, 2, true, 2>), grid2, dim3(256), 0, stream, Pm, tile_counter, order, rstart, pixel_cost, hf);
    }
    cur ^= 1;
    stage++;
  }
  // ---- the last stage: the work-sharing build on the paths left, to the end
  P.wave_log = stage_log();
  hf.in = plan.list[cur]; hf.in_count = stage; hf.out = nullptr; hf.out_count = 0;
  hf.wait = 0x7fffffff; hf.margin = 0;
  hf.resident_waves = cfg.num_cus * 5 * 4;
  hipLaunchKernelGGL((render_handoff_kernel<5, 32, 20, 2, true, 2>), grid2, dim3(256), 0, stream, P, tile_counter, order, rstart, pixel_cost, hf);
  stage++;
  return log ? stage * WAVE_LOG_STAGE_WAVES : 0;
}

}  // namespace dr
