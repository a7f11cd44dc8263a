// Error reporting and ABI version of libsr_hip.so.
#include <string.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "sr_internal.h"

namespace {
thread_local char g_err[512] = "";
}

__device__ __attribute__((aligned(64))) float g_sr_zero_line[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

namespace sr {
int ensure_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& d : done)
    if (d.first == kernel && d.second == dev) return SR_OK;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    set_error("hipFuncSetAttribute(%d B of LDS): %s", bytes, hipGetErrorString(e));
    return SR_ELAUNCH;
  }
  done.emplace_back(kernel, dev);
  return SR_OK;
}

const void* zero_line() {
  static void* cache[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!cache[dev]) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_sr_zero_line)) == hipSuccess) cache[dev] = ptr;
  }
  return cache[dev];
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace sr

extern "C" int sr_version(void) { return SR_ABI_VERSION; }
extern "C" const char* sr_last_error(void) { return g_err; }

// ---- opt-in launch profiler (process-wide, mutex-guarded; off unless sr_profile_start was called).
// Process-wide because autograd runs the backward launches on its own thread. ----
#include <map>
#include <mutex>
#include <vector>

namespace {
struct ProfState {
  bool on = false;
  size_t cap = 0;
  std::vector<sr_launch_record> recs;
  std::vector<hipEvent_t> ev;  // 2 per record
};
ProfState g_prof;
std::mutex g_prof_mu;
}  // namespace

namespace sr {
bool prof_on() {
  if (!g_prof.on) return false;  // unsynchronised fast path: the flag only flips inside start/stop
  std::lock_guard<std::mutex> lk(g_prof_mu);
  return g_prof.on && g_prof.recs.size() < g_prof.cap;
}
void prof_begin(hipStream_t s, const sr_launch_record& r) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipEventRecord(a, s);
  g_prof.recs.push_back(r);
  g_prof.ev.push_back(a);
  g_prof.ev.push_back(b);
}
void prof_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof.ev.back(), s);
}
}  // namespace sr

extern "C" int sr_profile_start(int max_records) {
  SR_CHECK_ARG(max_records > 0, "sr_profile_start: max_records must be positive");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  SR_CHECK_ARG(!g_prof.on, "sr_profile_start: already recording");
  g_prof.on = true;
  g_prof.cap = (size_t)max_records;
  g_prof.recs.clear();
  g_prof.ev.clear();
  return SR_OK;
}

extern "C" int sr_profile_stop(sr_launch_record* out, int capacity, int* count) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  SR_CHECK_ARG(g_prof.on, "sr_profile_stop: not recording");
  g_prof.on = false;
  int rc = SR_OK;
  const int n = (int)g_prof.recs.size();
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess ||
        hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) {
      sr::set_error("sr_profile_stop: event %d failed", i);
      rc = SR_ELAUNCH;
    }
    g_prof.recs[i].ms = ms;
    if (out && i < capacity) out[i] = g_prof.recs[i];
    (void)hipEventDestroy(g_prof.ev[2 * i]);
    (void)hipEventDestroy(g_prof.ev[2 * i + 1]);
  }
  if (count) *count = n;
  g_prof.recs.clear();
  g_prof.ev.clear();
  return rc;
}

// ---- tuning knob: number of concurrent image groups of the whole-network forward ----
namespace {
int g_forward_groups = 0;  // 0 = per-path default
}
namespace sr {
int forward_groups() { return g_forward_groups; }
bool SideStreams::ensure() {
  if (device >= 0) return true;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  for (int i = 0; i < 3; ++i) {
    if (hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&join[i], hipEventDisableTiming) != hipSuccess) return false;
  }
  if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return false;
  device = dev;
  return true;
}
int& launch_concurrency() {
  thread_local int c = 1;
  return c;
}
SideStreams& side_streams() {  // one set per (thread, device): streams and events belong to the device they were made on
  thread_local std::map<int, SideStreams> per_device;
  int dev = 0;
  (void)hipGetDevice(&dev);
  return per_device[dev];
}
}  // namespace sr
// ---- hand-off watchdog: a dense-block launch whose tiles waited for each other in vain raises the abort word of its sync block
// (bounded spins: it never hangs, but what it wrote is invalid).  The whole-network drivers copy those words into pinned host memory
// behind their launches (asynchronously: nothing waits) and look at the copies of EARLIER calls when they are entered, so a
// time-out — a GPU shared with another process, CUs withheld from this one — surfaces as an error on the next call instead of as
// silently wrong images.  sr_chain_watchdog() reports and clears it at any time (after a synchronisation: for the work before it).
namespace {
// One watch per DEVICE for the whole process (mutex-guarded): autograd runs the backward drivers on its own thread, and a word those
// queue must be seen by sr_chain_watchdog() on the caller's thread.
struct Watch {
  static constexpr int kSlots = 256;
  int32_t* host = nullptr;  // kSlots pinned words
  hipEvent_t landed[kSlots] = {};  // recorded behind the copy into a slot: a slot is only reused once its copy has landed
  bool queued[kSlots] = {};
  int next = 0;
  bool sticky = false;  // a raised word that was about to be overwritten before anybody looked
};
std::mutex g_watch_mu;
std::map<int, Watch> g_watch;
Watch& watch_of_device() {  // g_watch_mu held
  int dev = 0;
  (void)hipGetDevice(&dev);
  Watch& w = g_watch[dev];
  if (!w.host && hipHostMalloc((void**)&w.host, Watch::kSlots * sizeof(int32_t), hipHostMallocDefault) == hipSuccess)
    for (int i = 0; i < Watch::kSlots; ++i) w.host[i] = 0;
  return w;
}
}  // namespace
__device__ __attribute__((aligned(64))) int32_t g_sr_abort_latch[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
namespace {
__global__ void abort_latch_kernel(const int32_t* __restrict__ word, int32_t* __restrict__ latch) {
  if (threadIdx.x == 0 && *word != 0) *latch = 1;
}
__global__ void abort_latch_clear_kernel(int32_t* __restrict__ latch) {
  if (threadIdx.x == 0) *latch = 0;
}
}  // namespace
namespace sr {
const int32_t* abort_latch() {  // device address of the calling thread's current device's latch word
  static void* cache[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!cache[dev]) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_sr_abort_latch)) == hipSuccess) cache[dev] = ptr;
  }
  return (const int32_t*)cache[dev];
}
void chain_watch(const int32_t* abort_word, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_watch_mu);
  Watch& w = watch_of_device();
  if (!w.host || !abort_word) return;
  const int slot = w.next;
  w.next = (w.next + 1) % Watch::kSlots;
  // The copy queued into this slot kSlots calls ago may still be in flight when the host runs far ahead of the device (no step
  // synchronises any more): reading and re-arming the slot now could lose a word that lands afterwards.  Wait for THAT copy (rare,
  // and only when the host is more than kSlots watched launches ahead), then keep what it brought.
  if (w.queued[slot] && w.landed[slot] && hipEventQuery(w.landed[slot]) != hipSuccess) (void)hipEventSynchronize(w.landed[slot]);
  if (((volatile int32_t*)w.host)[slot] != 0) {  // kSlots copies ago and never checked: keep it
    w.sticky = true;
    w.host[slot] = 0;
  }
  (void)hipMemcpyAsync(w.host + slot, abort_word, sizeof(int32_t), hipMemcpyDeviceToHost, stream);
  if (!w.landed[slot] && hipEventCreateWithFlags(&w.landed[slot], hipEventDisableTiming) != hipSuccess) w.landed[slot] = nullptr;
  w.queued[slot] = w.landed[slot] && hipEventRecord(w.landed[slot], stream) == hipSuccess;
  // and the device-side latch: consumers on the stream (the optimiser) see the fact without the host
  hipLaunchKernelGGL(abort_latch_kernel, dim3(1), dim3(64), 0, stream, abort_word, (int32_t*)sr::abort_latch());
}
int chain_check(const char* who) {
  std::lock_guard<std::mutex> lk(g_watch_mu);
  Watch& w = watch_of_device();
  if (!w.host) return SR_OK;
  bool hit = w.sticky;
  w.sticky = false;
  for (int i = 0; i < Watch::kSlots; ++i)
    if (((volatile int32_t*)w.host)[i] != 0) {
      hit = true;
      w.host[i] = 0;
    }
  if (!hit) return SR_OK;
  // The report ends the episode: optimiser steps queued BEFORE this point (the host had not noticed yet) still see the device latch
  // and refuse; whatever the caller issues after it was told runs normally again.  The null stream orders the clear behind
  // everything queued so far on the blocking streams (torch's current stream is one of them).
  if (int32_t* latch = (int32_t*)abort_latch()) hipLaunchKernelGGL(abort_latch_clear_kernel, dim3(1), dim3(64), 0, (hipStream_t) nullptr, latch);
  set_error("%s: an earlier dense-block launch timed out waiting for a neighbour tile; its results were invalid.  The fused kernel "
            "needs about six rows of 16x32 tiles resident at once; if the GPU cannot give this process that many CUs, call "
            "sr_set_conv_chain(2): the chain launch has no such need", who);
  return SR_ELAUNCH;
}
}  // namespace sr
extern "C" int sr_chain_watchdog(void) { return sr::chain_check("sr_chain_watchdog"); }
extern "C" const int32_t* sr_abort_latch(void) { return sr::abort_latch(); }
// Development hook (tests): raise the device latch from a device word WITHOUT the host-side copy, i.e. the window in which the
// device knows of a time-out and the host does not yet.
extern "C" void sr_dev_abort_latch_from(const int32_t* word, void* stream) {
  hipLaunchKernelGGL(abort_latch_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, word, (int32_t*)sr::abort_latch());
}
extern "C" int sr_abort_latch_clear(void* stream) {
  int32_t* latch = (int32_t*)sr::abort_latch();
  SR_CHECK_ARG(latch, "sr_abort_latch_clear: no latch on this device");
  hipLaunchKernelGGL(abort_latch_clear_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, latch);
  SR_CHECK_LAUNCH("abort_latch_clear");
  return SR_OK;
}
// Development hook (tests): queue a copy of one device word the way the network drivers queue their abort words.
extern "C" void sr_dev_chain_watch(const int32_t* word, void* stream) { sr::chain_watch(word, (hipStream_t)stream); }

// ---- the weight-gradient lane of the backward drivers (sr_internal.h) ----
namespace {
int g_backward_overlap = -1;
struct LaneEvents {
  hipEvent_t handoff = nullptr, ring[4] = {nullptr, nullptr, nullptr, nullptr};
  bool ok = false;
};
}  // namespace
namespace sr {
int backward_overlap() { return g_backward_overlap; }
bool WgradLane::begin(hipStream_t caller, bool enable) {
  main = side = caller;
  on = false;
  last = -1;
  if (!enable || prof_on()) return false;  // (the launch profiler times launches on one stream)
  SideStreams& S = side_streams();
  if (!S.ensure()) return false;
  thread_local std::map<int, LaneEvents> per_device;  // events belong to the device they were made on
  LaneEvents& E = per_device[S.device];
  if (!E.ok) {
    if (hipEventCreateWithFlags(&E.handoff, hipEventDisableTiming) != hipSuccess) return false;
    for (int i = 0; i < 4; ++i)
      if (hipEventCreateWithFlags(&E.ring[i], hipEventDisableTiming) != hipSuccess) return false;
    E.ok = true;
  }
  side = S.s[2];
  handoff = E.handoff;
  for (int i = 0; i < 4; ++i) ring[i] = E.ring[i];
  on = true;
  return true;
}
}  // namespace sr
namespace {
struct PendingLane {
  hipEvent_t ev = nullptr;
  bool armed = false;
};
std::mutex g_pending_mu;
std::map<int, PendingLane> g_pending;  // per device: autograd issues the backward on its own thread, the optimiser joins on another
}  // namespace
namespace sr {
void lane_detach(WgradLane& lane) {
  if (!lane.on) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_pending_mu);
  PendingLane& p = g_pending[dev];
  if (!p.ev && hipEventCreateWithFlags(&p.ev, hipEventDisableTiming) != hipSuccess) p.ev = nullptr;
  if (p.ev && hipEventRecord(p.ev, lane.side) == hipSuccess) {
    p.armed = true;
    lane.on = false;  // no join at the end of the call
  } else {
    lane.end();
  }
}
}  // namespace sr
extern "C" int sr_backward_lane_join(void* stream) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_pending_mu);
  PendingLane& p = g_pending[dev];
  if (p.armed && p.ev) {
    if (hipStreamWaitEvent((hipStream_t)stream, p.ev, 0) != hipSuccess) {
      sr::set_error("sr_backward_lane_join: hipStreamWaitEvent failed");
      return SR_ELAUNCH;
    }
    p.armed = false;
  }
  return SR_OK;
}
namespace {
int g_bn_small = 1;
}
namespace sr {
bool bn_small_enabled() { return g_bn_small != 0; }
}
// Development switch (A/B, tests; not part of the ABI): 0 = BatchNorm of small tensors on the general multi-launch path.
extern "C" void sr_dev_set_bn_small(int on) { g_bn_small = on; }
// Development switch (not part of the ABI): -1 = automatic, 0 = weight gradients on the caller's stream, 1 = always on the lane.
extern "C" void sr_dev_set_backward_overlap(int mode) { g_backward_overlap = mode; }

extern "C" int sr_set_forward_groups(int groups) {
  SR_CHECK_ARG(groups >= 0 && groups <= 4, "sr_set_forward_groups: 0 (automatic) or 1..4");
  g_forward_groups = groups;
  return SR_OK;
}
