// dr_group: one process, one device context and one host thread per GPU, framebuffer stripes gathered to rank 0.
//
// The reference is single-device (kernel.cu K:2614-2615).  Pixels are independent and seeded by (x, y, frame) (K:1065),
// so rank r of R renders the 8-pixel block columns bx % R == r (dr_context_set_stripe) of every frame into its own
// accumulator; every `gather_every` frames each rank packs its stripe (one contiguous run of 8*H*3 int32 per column,
// K:1006) and sends it to rank 0, which writes it into its accumulator's columns r, r+R, ....  Everything is queued:
//
//   render stream (per rank)   [render batch k] [pack -> slot k&1] ........ [render batch k+1] [pack -> slot (k+1)&1]
//   comm stream   (per rank)                     wait packed(k) : ncclSend / ncclRecv x (R-1) + unpack : record sent(k)
//
// so the gather of batch k runs beside the rendering of batch k+1 (ranks render disjoint columns, the unpack on rank 0
// touches only other ranks' columns); a slot is packed again only after its send has finished (render stream waits on
// sent(k-2)).  Transport: RCCL point-to-point (ncclSend / ncclRecv, grouped on rank 0: the R-1 transfers land on R-1
// distinct xGMI links), resolved from librccl.so at run time; DOGERAY_GROUP_TRANSPORT=copy (or several ranks on one
// device, as the single-GPU tests do) uses hipMemcpyPeerAsync + events instead.
// Threads: one per rank, created with the group and fed jobs through a condition variable (none is created inside a render call).
// Failure: a rank that fails records the first error, releases the rendezvous and raises `failed`; every other rank sees the flag before its next
// RCCL call or while it waits for its streams -- that wait is a hipStreamQuery poll with a deadline (DOGERAY_GROUP_TIMEOUT_S, default 30 s), never a
// bare hipStreamSynchronize -- and returns.  Only then, when no rank thread is inside RCCL any more, the coordinator aborts the communicators
// (transfers that will never be matched are cancelled) and marks the group broken.  A failed upload breaks nothing.
// This file is a client of the C ABI (include/dogeray_amd.h) plus HIP events/streams and RCCL: no kernels.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "scene_host.hpp"

using namespace dr;

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load() {
    if (handle) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (handle) break;
    }
    if (!handle) return false;
    auto sym = [&](const char* n) { return dlsym(handle, n); };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
    CommCount = (decltype(CommCount))sym("ncclCommCount");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    return CommInitAll && CommDestroy && CommAbort && CommCount && GroupStart && GroupEnd && Send && Recv && GetErrorString;
  }
};

// all ranks meet here once per batch (copy transport only: an event must have been recorded before another thread waits on it)
struct Rendezvous {
  std::mutex m; std::condition_variable cv; int waiting = 0; unsigned long generation = 0; int n = 1; bool aborted = false;
  void arrive() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned long g = generation;
    if (++waiting == n) { waiting = 0; generation++; cv.notify_all(); }
    else cv.wait(lk, [&] { return generation != g || aborted; });
  }
  void abort() { std::lock_guard<std::mutex> lk(m); aborted = true; cv.notify_all(); }      // a rank failed: nobody waits for it
};

}  // namespace

struct dr_group {
  int n = 0;
  std::vector<int> device;
  std::vector<dr_context*> ctx;
  std::vector<hipStream_t> comm;                   // one communication stream per rank, on that rank's device
  std::vector<hipEvent_t> packed[2], sent[2];      // per slot, per rank
  bool use_rccl = false;
  Rccl rccl;
  std::vector<ncclComm_t> comms;
  int32_t* stage[2] = {nullptr, nullptr};          // rank 0's device: [n][stage_stride] int32 per slot
  size_t stage_stride = 0;                         // int32 per rank
  int W = 0, H = 0;
  Rendezvous meet;
  std::string error;                               // first failure of a worker
  std::mutex error_lock;
  std::atomic<bool> failed{false};                 // a rank failed in the call that is running: the others stop at their next check
  bool broken = false;                             // a render call failed under RCCL: the communicators were aborted, the group can only be destroyed
  double timeout_s = 30.0;                         // deadline of a rank's wait for its streams (DOGERAY_GROUP_TIMEOUT_S)
  int fail_rank = -1, fail_batch = -1;             // failure injection for the tests (DOGERAY_GROUP_FAIL_RANK / _BATCH): that rank fails before queuing that batch
  // rank threads: created with the group, one job at a time for all of them
  std::vector<std::thread> threads;
  std::mutex job_lock; std::condition_variable job_cv, done_cv;
  std::function<bool(int)> job; unsigned long job_id = 0; int job_left = 0; bool quit = false;
  std::vector<char> job_ok;
  // the first error of a call; wakes everybody who waits for the failed rank.  Touches no communicator: the rank threads may be inside RCCL.
  void fail(const std::string& msg) {
    { std::lock_guard<std::mutex> g(error_lock); if (error.empty()) error = msg; }
    failed.store(true);
    meet.abort();
  }
};

namespace {

#define G_HIP(expr)                                                                                       \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) { g->fail(std::string(#expr) + ": " + hipGetErrorString(e_)); return false; }   \
  } while (0)
#define G_NCCL(expr)                                                                                      \
  do {                                                                                                    \
    if (g->failed.load()) return false;             /* another rank failed: no further transfer is queued */ \
    ncclResult_t r_ = (expr);                                                                             \
    if (r_ != ncclSuccess) { g->fail(std::string(#expr) + ": " + g->rccl.GetErrorString(r_)); return false; } \
  } while (0)
#define G_DR(expr)                                                                                        \
  do {                                                                                                    \
    if ((expr) != DR_OK) { g->fail(std::string(#expr) + ": " + dr_last_error()); return false; }          \
  } while (0)

size_t stripe_elems(int W, int H, int world, int rank) {
  const int gx = W / 8;
  const int ncols = gx > rank ? (gx - rank + world - 1) / world : 0;
  return (size_t)ncols * 8 * (size_t)H * 3;
}

// waits until everything queued on `st` has run: a poll, so that the wait can end on a deadline or because another rank failed.
// 0 = idle, 1 = gave up (deadline / failed flag), 2 = the stream reports an error
int wait_stream(dr_group* g, hipStream_t st, std::chrono::steady_clock::time_point t_end, bool watch_failed, hipError_t* err) {
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) { if (err) *err = q; return 2; }
    if (watch_failed && g->failed.load()) return 1;
    if (std::chrono::steady_clock::now() > t_end) return 1;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

// the same for one event (recorded before the call)
int wait_event(dr_group* g, hipEvent_t ev, std::chrono::steady_clock::time_point t_end, hipError_t* err) {
  for (;;) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) { if (err) *err = q; return 2; }
    if (g->failed.load()) return 1;
    if (std::chrono::steady_clock::now() > t_end) return 1;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

// one rank's share of dr_group_render_accumulate
bool rank_work(dr_group* g, int r, const float* st, int W, int H, float bg, uint64_t seed, uint64_t stride, int nframes, int every) {
  G_HIP(hipSetDevice(g->device[(size_t)r]));
  dr_context* c = g->ctx[(size_t)r];
  hipStream_t render = nullptr;
  G_DR(dr_context_stream(c, (void**)&render));
  hipStream_t comm = g->comm[(size_t)r];
  const bool copy = !g->use_rccl;
  int batch = 0;
  for (int k = 0; k < nframes; k += every, batch++) {
    const int n = nframes - k < every ? nframes - k : every;
    const int slot = batch & 1;
    if (g->failed.load()) return false;
    if (r == g->fail_rank && batch == g->fail_batch) { g->fail("rank " + std::to_string(r) + ": injected failure before batch " + std::to_string(batch)); return false; }
    if (batch >= 2) {
      // the slot's previous contents must have left before it is packed again -- and the library call below waits on the host for the batch before last
      // (at most two are in flight): that wait must not be the place where a transfer that is never matched hangs the call, so it is made here, bounded
      const auto t_slot = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(g->timeout_s));
      hipError_t err = hipSuccess;
      const int w = wait_event(g, g->sent[slot][(size_t)r], t_slot, &err);
      if (w == 2) { g->fail(std::string("rank ") + std::to_string(r) + ": hipEventQuery: " + hipGetErrorString(err)); return false; }
      if (w == 1) {
        if (!g->failed.load()) g->fail("rank " + std::to_string(r) + ": the gather of batch " + std::to_string(batch - 2) + " has not ended after " + std::to_string((int)g->timeout_s) + " s (DOGERAY_GROUP_TIMEOUT_S)");
        return false;
      }
      G_HIP(hipStreamWaitEvent(render, g->sent[slot][(size_t)r], 0));
    }
    G_DR(dr_render_accumulate_async(c, st, W, H, bg, seed + (uint64_t)k * stride, stride, n));
    if (g->n == 1) continue;
    void* pk = nullptr; uint64_t bytes = 0;
    if (r != 0) G_DR(dr_accum_pack_stripe(c, slot, &pk, &bytes));            // rank 0's own columns are already where they belong
    G_HIP(hipEventRecord(g->packed[slot][(size_t)r], render));
    G_HIP(hipStreamWaitEvent(comm, g->packed[slot][(size_t)r], 0));
    int32_t* stage = g->stage[slot];
    if (copy) {
      // stage[slot] was last read by rank 0's unpack of batch - 2, whose end is sent[slot][0] (recorded before that batch's second
      // rendezvous, so it exists by now): the copy into the same staging stripe may not overtake it
      if (r != 0 && batch >= 2) G_HIP(hipStreamWaitEvent(comm, g->sent[slot][0], 0));
      if (r != 0 && bytes) G_HIP(hipMemcpyPeerAsync(stage + (size_t)r * g->stage_stride, g->device[0], pk, g->device[(size_t)r], bytes, comm));
      G_HIP(hipEventRecord(g->sent[slot][(size_t)r], comm));
      g->meet.arrive();                                        // every rank has recorded sent(batch)
      if (g->failed.load()) return false;                      // (or was released because one of them failed)
      if (r == 0) {
        for (int q = 1; q < g->n; q++) G_HIP(hipStreamWaitEvent(comm, g->sent[slot][(size_t)q], 0));
        G_DR(dr_accum_unpack_stripes(c, stage, g->stage_stride * sizeof(int32_t), g->n, 1, comm));
        G_HIP(hipEventRecord(g->sent[slot][0], comm));
      }
      g->meet.arrive();                                        // nobody re-records an event rank 0 has not yet waited on
      if (g->failed.load()) return false;
    } else {
      if (r != 0) {
        if (bytes) G_NCCL(g->rccl.Send(pk, bytes / sizeof(int32_t), ncclInt32, 0, g->comms[(size_t)r], comm));
      } else {
        G_NCCL(g->rccl.GroupStart());
        for (int q = 1; q < g->n; q++) {
          const size_t cnt = stripe_elems(W, H, g->n, q);
          if (cnt) G_NCCL(g->rccl.Recv(stage + (size_t)q * g->stage_stride, cnt, ncclInt32, q, g->comms[0], comm));
        }
        G_NCCL(g->rccl.GroupEnd());
        G_DR(dr_accum_unpack_stripes(c, stage, g->stage_stride * sizeof(int32_t), g->n, 1, comm));
      }
      G_HIP(hipEventRecord(g->sent[slot][(size_t)r], comm));
    }
  }
  // wait for both streams: a poll with a deadline, so that a transfer that is never matched (or a rank that failed) cannot hang the call
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(g->timeout_s));
  for (hipStream_t st : {comm, render}) {
    hipError_t err = hipSuccess;
    const int w = wait_stream(g, st, t_end, true, &err);
    if (w == 2) { g->fail(std::string("rank ") + std::to_string(r) + ": hipStreamQuery: " + hipGetErrorString(err)); return false; }
    if (w == 1) {
      if (!g->failed.load())
        g->fail("rank " + std::to_string(r) + ": no end of the " + (st == comm ? "gather" : "rendering") + " of batch " + std::to_string(batch - 1) + " after " +
                std::to_string((int)g->timeout_s) + " s (DOGERAY_GROUP_TIMEOUT_S)");
      return false;
    }
  }
  G_DR(dr_context_synchronize(c));      // (both streams are idle: this collects the batches' times)
  return true;
}

// every rank thread runs f(rank); returns when all have (true: all succeeded)
bool for_all_ranks(dr_group* g, const std::function<bool(int)>& f) {
  std::unique_lock<std::mutex> lk(g->job_lock);
  g->job = f; g->job_left = g->n; g->job_ok.assign((size_t)g->n, 0); g->job_id++;
  g->job_cv.notify_all();
  g->done_cv.wait(lk, [&] { return g->job_left == 0; });
  g->job = nullptr;
  for (char o : g->job_ok) if (!o) return false;
  return true;
}

void rank_thread(dr_group* g, int r) {
  (void)hipSetDevice(g->device[(size_t)r]);
  unsigned long seen = 0;
  for (;;) {
    std::function<bool(int)> f;
    {
      std::unique_lock<std::mutex> lk(g->job_lock);
      g->job_cv.wait(lk, [&] { return g->quit || g->job_id != seen; });
      if (g->quit) return;
      seen = g->job_id; f = g->job;
    }
    const bool ok = f(r);
    {
      std::lock_guard<std::mutex> lk(g->job_lock);
      g->job_ok[(size_t)r] = ok ? 1 : 0;
      if (--g->job_left == 0) g->done_cv.notify_all();
    }
  }
}

// after a failed render call, when every rank thread has returned: cancel what RCCL still has queued (transfers that will never be matched)
void abort_communicators(dr_group* g) {
  if (!g->use_rccl) return;
  for (ncclComm_t& c : g->comms) if (c) { (void)g->rccl.CommAbort(c); c = nullptr; }
  g->broken = true;
}

}  // namespace

extern "C" {

int dr_group_create(int n, const int* device_ordinals, dr_group** out) {
  if (!out || n < 1) { set_error("group: need n >= 1"); return DR_ERR_INVALID; }
  *out = nullptr;
  dr_group* g = new dr_group();
  g->n = n;
  g->meet.n = n;
  bool distinct = true;
  // DOGERAY_GROUP_DEVICES="0,0,0": ordinals for a caller that passes none (rehearsing `dogeray --gpus 3` on one GPU)
  std::vector<int> env_dev;
  if (const char* e = getenv("DOGERAY_GROUP_DEVICES")) {
    for (const char* p = e; *p;) { env_dev.push_back(atoi(p)); while (*p && *p != ',') p++; if (*p == ',') p++; }
  }
  for (int r = 0; r < n; r++) {
    g->device.push_back(device_ordinals ? device_ordinals[r] : (r < (int)env_dev.size() ? env_dev[(size_t)r] : r));
    for (int q = 0; q < r; q++) if (g->device[(size_t)q] == g->device[(size_t)r]) distinct = false;
  }
  if (const char* e = getenv("DOGERAY_GROUP_TIMEOUT_S")) { const double v = atof(e); if (v > 0) g->timeout_s = v; }
  if (const char* e = getenv("DOGERAY_GROUP_FAIL_RANK")) g->fail_rank = atoi(e);
  if (const char* e = getenv("DOGERAY_GROUP_FAIL_BATCH")) g->fail_batch = atoi(e);
  const char* tr = getenv("DOGERAY_GROUP_TRANSPORT");
  // DOGERAY_GROUP_TRANSPORT=copy: peer copies even between distinct devices; =rccl: RCCL even for one rank (rehearsal: loads the
  // library, creates the communicator and sends a buffer to itself, which is all a one-GPU box allows)
  const bool force_rccl = tr && std::string(tr) == "rccl" && distinct;
  g->use_rccl = (n > 1 && distinct && !(tr && std::string(tr) == "copy")) || force_rccl;
  auto bail = [&](int rc, const std::string& msg) { set_error(msg); dr_group_destroy(g); return rc; };
  g->ctx.assign((size_t)n, nullptr);
  g->comm.assign((size_t)n, nullptr);
  for (int k = 0; k < 2; k++) { g->packed[k].assign((size_t)n, nullptr); g->sent[k].assign((size_t)n, nullptr); }
  for (int r = 0; r < n; r++) {
    if (dr_context_create(g->device[(size_t)r], &g->ctx[(size_t)r]) != DR_OK) return bail(DR_ERR_DEVICE, std::string("group: ") + dr_last_error());
    if (dr_context_set_stripe(g->ctx[(size_t)r], n, r) != DR_OK) return bail(DR_ERR_INVALID, std::string("group: ") + dr_last_error());
    if (hipSetDevice(g->device[(size_t)r]) != hipSuccess || hipStreamCreateWithFlags(&g->comm[(size_t)r], hipStreamNonBlocking) != hipSuccess)
      return bail(DR_ERR_DEVICE, "group: cannot create the communication stream");
    for (int k = 0; k < 2; k++)
      if (hipEventCreateWithFlags(&g->packed[k][(size_t)r], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&g->sent[k][(size_t)r], hipEventDisableTiming) != hipSuccess)
        return bail(DR_ERR_DEVICE, "group: cannot create events");
  }
  if (g->use_rccl) {
    if (!g->rccl.load()) return bail(DR_ERR_DEVICE, "group: librccl.so not found (set DOGERAY_GROUP_TRANSPORT=copy for peer copies)");
    g->comms.assign((size_t)n, nullptr);
    ncclResult_t rc = g->rccl.CommInitAll(g->comms.data(), n, g->device.data());
    if (rc != ncclSuccess) { g->comms.clear(); return bail(DR_ERR_DEVICE, std::string("group: ncclCommInitAll: ") + g->rccl.GetErrorString(rc)); }
    if (force_rccl) {
      // self-check of the resolved entry points: rank 0 sends 4096 int32 to itself (grouped send + recv on its communication stream)
      std::vector<int32_t> host(4096), back(4096, 0);
      for (size_t i = 0; i < host.size(); i++) host[i] = (int32_t)(i * 2654435761u);
      int32_t *a = nullptr, *b = nullptr;
      bool ok = hipSetDevice(g->device[0]) == hipSuccess && hipMalloc((void**)&a, host.size() * 4) == hipSuccess && hipMalloc((void**)&b, host.size() * 4) == hipSuccess &&
                hipMemcpy(a, host.data(), host.size() * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemset(b, 0, host.size() * 4) == hipSuccess;
      ok = ok && g->rccl.GroupStart() == ncclSuccess;
      ok = ok && g->rccl.Send(a, host.size(), ncclInt32, 0, g->comms[0], g->comm[0]) == ncclSuccess;
      ok = ok && g->rccl.Recv(b, host.size(), ncclInt32, 0, g->comms[0], g->comm[0]) == ncclSuccess;
      ok = ok && g->rccl.GroupEnd() == ncclSuccess;
      ok = ok && hipStreamSynchronize(g->comm[0]) == hipSuccess && hipMemcpy(back.data(), b, host.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
      ok = ok && back == host;
      if (a) (void)hipFree(a);
      if (b) (void)hipFree(b);
      if (!ok) return bail(DR_ERR_DEVICE, "group: RCCL self-check (send to self) failed");
    }
  } else if (n > 1) {
    for (int r = 1; r < n; r++) {                 // peer copies into rank 0's staging buffer
      if (g->device[(size_t)r] == g->device[0]) continue;
      (void)hipSetDevice(g->device[(size_t)r]);
      (void)hipDeviceEnablePeerAccess(g->device[0], 0);
    }
  }
  for (int r = 0; r < n; r++) g->threads.emplace_back(rank_thread, g, r);
  *out = g;
  return DR_OK;
}

void dr_group_destroy(dr_group* g) {
  if (!g) return;
  { std::lock_guard<std::mutex> lk(g->job_lock); g->quit = true; }
  g->job_cv.notify_all();
  for (std::thread& t : g->threads) t.join();
  for (ncclComm_t c : g->comms) if (c) (void)g->rccl.CommDestroy(c);
  for (int r = 0; r < g->n; r++) {
    if (r < (int)g->device.size()) (void)hipSetDevice(g->device[(size_t)r]);
    if (r < (int)g->comm.size() && g->comm[(size_t)r]) { (void)hipStreamSynchronize(g->comm[(size_t)r]); (void)hipStreamDestroy(g->comm[(size_t)r]); }
    for (int k = 0; k < 2; k++) {
      if (r < (int)g->packed[k].size() && g->packed[k][(size_t)r]) (void)hipEventDestroy(g->packed[k][(size_t)r]);
      if (r < (int)g->sent[k].size() && g->sent[k][(size_t)r]) (void)hipEventDestroy(g->sent[k][(size_t)r]);
    }
  }
  if (!g->device.empty()) (void)hipSetDevice(g->device[0]);
  for (int k = 0; k < 2; k++) if (g->stage[k]) (void)hipFree(g->stage[k]);
  for (dr_context* c : g->ctx) if (c) dr_context_destroy(c);
  delete g;
}

int dr_group_size(const dr_group* g) { return g ? g->n : DR_ERR_INVALID; }
dr_context* dr_group_context(dr_group* g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[(size_t)rank] : nullptr; }
int dr_group_uses_rccl(const dr_group* g) { return g && g->use_rccl ? 1 : 0; }
int dr_group_rccl_ranks(const dr_group* g) {
  if (!g || !g->use_rccl || g->comms.empty() || !g->comms[0]) return 0;
  int n = 0;
  if (g->rccl.CommCount(g->comms[0], &n) != ncclSuccess) return 0;
  return n;
}

int dr_group_upload_scene(dr_group* g, const dr_scene* s) {
  if (!g || !s) { set_error("null argument"); return DR_ERR_INVALID; }
  if (g->broken) { set_error("group: unusable after a failed call (its RCCL communicators were aborted); destroy it"); return DR_ERR_DEVICE; }
  g->error.clear(); g->failed.store(false);
  const bool ok = for_all_ranks(g, [&](int r) {
    if (dr_context_upload_scene(g->ctx[(size_t)r], s) != DR_OK) { g->fail(std::string("upload on rank ") + std::to_string(r) + ": " + dr_last_error()); return false; }
    return true;
  });
  if (!ok) { set_error(g->error); return DR_ERR_DEVICE; }      // (nothing was in flight between the ranks: the group stays usable)
  return DR_OK;
}

int dr_group_accum_reset(dr_group* g, int W, int H) {
  if (!g || W <= 0 || H <= 0) { set_error("bad argument"); return DR_ERR_INVALID; }
  for (int r = 0; r < g->n; r++) {
    if (dr_accum_reset(g->ctx[(size_t)r], W, H) != DR_OK) return DR_ERR_DEVICE;
    // both pack buffers now, not inside the first gathers of a timed region (dr_accum_pack_stripe allocates on first use)
    if (g->n > 1 && r != 0 && (dr_accum_reserve_pack(g->ctx[(size_t)r], 0) != DR_OK || dr_accum_reserve_pack(g->ctx[(size_t)r], 1) != DR_OK)) return DR_ERR_NOMEM;
  }
  const size_t stride = (stripe_elems(W, H, g->n, 0) + 3) & ~(size_t)3;          // rank 0 owns the most columns; 16-byte multiple
  if (g->n > 1 && (stride != g->stage_stride || !g->stage[0])) {
    if (hipSetDevice(g->device[0]) != hipSuccess) { set_error("group: hipSetDevice"); return DR_ERR_DEVICE; }
    for (int k = 0; k < 2; k++) {
      if (g->stage[k]) { (void)hipFree(g->stage[k]); g->stage[k] = nullptr; }
      if (hipMalloc((void**)&g->stage[k], (size_t)g->n * stride * sizeof(int32_t)) != hipSuccess) { set_error("group: cannot allocate the staging buffer"); return DR_ERR_NOMEM; }
    }
    g->stage_stride = stride;
  }
  g->W = W; g->H = H;
  return DR_OK;
}

int dr_group_render_accumulate(dr_group* g, const float settings13[13], int W, int H, float background, uint64_t frame_seed,
                               uint64_t seed_stride, int nframes, int gather_every) {
  if (!g || !settings13 || nframes < 0) { set_error("bad argument"); return DR_ERR_INVALID; }
  if (g->W != W || g->H != H) { set_error("call dr_group_accum_reset(W, H) first"); return DR_ERR_INVALID; }
  if (g->broken) { set_error("group: unusable after a failed call (its RCCL communicators were aborted); destroy it"); return DR_ERR_DEVICE; }
  if (nframes == 0) return DR_OK;
  const int every = gather_every > 0 ? gather_every : nframes;
  g->error.clear(); g->failed.store(false);
  g->meet.aborted = false; g->meet.waiting = 0;
  const bool ok = for_all_ranks(g, [&](int r) { return rank_work(g, r, settings13, W, H, background, frame_seed, seed_stride, nframes, every); });
  if (!ok) {
    abort_communicators(g);      // every rank thread is back: nobody is inside RCCL
    // what the ranks had queued may still be running: wait for it (bounded: a kernel ends, an aborted transfer is gone) before the caller touches anything
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(g->timeout_s));
    for (int r = 0; r < g->n; r++) {
      (void)hipSetDevice(g->device[(size_t)r]);
      hipStream_t render = nullptr;
      (void)dr_context_stream(g->ctx[(size_t)r], (void**)&render);
      if (wait_stream(g, g->comm[(size_t)r], t_end, false, nullptr) == 0 && wait_stream(g, render, t_end, false, nullptr) == 0) (void)dr_context_synchronize(g->ctx[(size_t)r]);
      else g->broken = true;      // (still busy after the deadline: nothing more can be done with this group but destroying it)
    }
    set_error("group: " + g->error);
    return DR_ERR_DEVICE;
  }
  return DR_OK;
}

}  // extern "C"
