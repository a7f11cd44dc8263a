// Internal helpers shared by the HIP translation units of libsr_hip.so (gfx950 only).
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/sr_hip.h"

namespace sr {

void set_error(const char* fmt, ...);
bool prof_on();
void prof_begin(hipStream_t s, const sr_launch_record& r);
void prof_end(hipStream_t s);
size_t rdb_dgrad_step_floats(int nf, int gc, int s);
int rdb_pack_dgrad_step(const float* const w[5], int nf, int gc, int s, float scale5, float* out, hipStream_t stream);
size_t rdb_dgrad_step_elems16(int nf, int gc, int s);
int rdb_pack_dgrad_step_bf16(const float* const w[5], int nf, int gc, int s, float scale5, void* out, hipStream_t stream);
struct WgradReduce {  // wgrad_f32.hip: slab reduction shared with wgrad_bf16.hip
  const float* slab;
  const float* bslab;  // null: no bias gradient
  float* part;
  float* bpart;
  long long splits;
  long long split_stride;  // floats between consecutive splits of the slab (0: P*ntap*1024, dense)
  int bsplit_stride;       // floats between consecutive splits of the bias slab (0: CT*32)
  int groups, gi;          // multi-group launches: groups = rows x gi tile groups behind one another in the slab (0: 1)
  int P, IT, CT, ntap, ks, kdim, t_mul, dy_off, dx_off, cin_tile0, cout_tile0;
  int cout, cin, first_seg, seg, seg_pad;
  float scale;
  int accumulate;
  float* dw;
  float* db;
};
int wgrad_reduce(const WgradReduce& r, hipStream_t stream);
// up to 8 independent single-group rows with equal split counts in two launches; the rows share r[0].part / r[0].bpart
int wgrad_reduce_rows(const WgradReduce* r, int nrows, hipStream_t stream);
int rdb_wgrad_bf16(const void* cat, const void* D, long long ns, int n, int h, int w, int nf, int gc, float* const* dparams,
                   float scale5, int accumulate, void* slab, size_t slab_bytes, hipStream_t stream);
// wgrad_f32.hip: the five weight gradients of a dense block as one launch + one table-driven reduction (fp32 twin of rdb_wgrad_bf16)
bool rdb_wgrad_f32_enabled();
size_t rdb_wgrad_slab_bytes_f32(int n, int h, int w, int nf, int gc);
int rdb_wgrad_f32(const float* cat, const float* D, long long ns, int n, int h, int w, int nf, int gc, float* const* dparams,
                  float scale5, int accumulate, void* slab, size_t slab_bytes, hipStream_t stream);
// pack_net.hip: one-launch packing of every weight image of a network
struct PackEntry {
  const float* w[5];  // kind 0/1: w[0] = OIHW weight; kind 2: conv1..conv5 of the dense block
  const float* bias;  // kind 0
  void* out;          // image
  float* bout;        // kind 0: packed bias
  int kind;           // 0 forward image + bias, 1 data-gradient image, 2 transposed-dense-block step
  int cout, cin, first_seg, seg, cin_pad;
  int nf, gc, s;
  float scale5;
  long long block0, elems;
};
size_t pack_table_bytes(size_t entries);
int pack_table_run(std::vector<PackEntry>& entries, void* blob, size_t image_bytes, bool bf16, hipStream_t stream);
// Device address of a 64-byte line of zeros (padding source of the LDS-DMA loaders).  Kernels take it as a parameter:
// naming the __device__ symbol inside a loop makes hipcc re-load its address (s_getpc + s_load + wait) at every use.
const void* zero_line();
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): a per-kernel `static bool` would leave the
// second device of a process without the attribute.
int ensure_dynamic_lds(const void* kernel, int bytes);
// capi.hip: hand-off watchdog of the dense-block launches (asynchronous copy of a sync block's abort word / check of earlier copies)
void chain_watch(const int32_t* abort_word, hipStream_t stream);
int chain_check(const char* who);
const int32_t* abort_latch();  // device word raised behind a dense-block launch whose abort word went up (sr_abort_latch)
int chain_mode();      // conv_bf16.hip: sr_set_conv_chain's value
// conv_bf16.hip: sr_conv3x3_chain_bf16 with the caller's word that only the last conv's output is read afterwards
int conv3x3_chain_bf16(const sr_conv3x3_desc* d, int nconv, int32_t* sync, int call_index, hipStream_t stream, bool mids_scratch);
int forward_groups();  // image groups of the forward (sr_set_forward_groups; 0 = the path's own default)
// Side streams + fork / join events of the calling thread for the grouped forward (created once per device).
struct SideStreams {
  hipStream_t s[3] = {nullptr, nullptr, nullptr};
  hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr};
  int device = -1;
  bool ensure();
};
SideStreams& side_streams();
// Number of image groups whose launches run concurrently with the calling thread's (1 outside a grouped forward): the
// tile dispatch of the convs counts the workgroups of all groups when it decides whether a launch fills the chip.
int& launch_concurrency();
// Runs body(g, n0, cnt, stream_g) for `groups` contiguous image ranges of n: group 0 on `stream`, the others on side streams
// that fork from and join back into `stream` (no host synchronisation; the caller's stream order is preserved).
template <class Body>
int run_image_groups(int n, int groups, hipStream_t stream, Body body) {
  SideStreams& S = side_streams();
  if (groups <= 1 || !S.ensure()) return body(0, 0, n, stream);
  if (hipEventRecord(S.fork, stream) != hipSuccess) return SR_ELAUNCH;
  launch_concurrency() = groups;
  int rc = SR_OK, n0 = 0;
  for (int g = 0; g < groups && rc == SR_OK; ++g) {
    const int cnt = n / groups + (g < n % groups ? 1 : 0);
    hipStream_t s = g == 0 ? stream : S.s[g - 1];
    if (g > 0 && hipStreamWaitEvent(s, S.fork, 0) != hipSuccess) rc = SR_ELAUNCH;
    if (rc == SR_OK) rc = body(g, n0, cnt, s);
    if (g > 0 && (hipEventRecord(S.join[g - 1], s) != hipSuccess || hipStreamWaitEvent(stream, S.join[g - 1], 0) != hipSuccess))
      rc = rc ? rc : SR_ELAUNCH;
    n0 += cnt;
  }
  launch_concurrency() = 1;
  return rc;
}

// A second lane (side stream) for work of a driver call that nothing on the caller's stream waits for before the call's end: the
// weight gradients of a backward pass, which depend on each block's data gradients but feed only the optimiser.  On small launches
// (the reference recipe's 32x32 patches: 64-256 workgroups of a few microseconds each) the data-gradient chain and the weight
// gradients then overlap instead of queueing behind one another.  Stream order replaces every host synchronisation:
//   hand()   work issued so far on the caller's stream is a dependency of what the lane gets next; returns the lane's stream
//   done(k)  marks the lane's progress as `ticket k` (a ring of 4 events)
//   need(k)  the caller's stream waits for ticket k — before it overwrites a buffer the lane's work of ticket k reads
//   end()    the caller's stream waits for everything the lane was given
// Disabled (on = false) every call degenerates to the caller's stream and no events.
struct WgradLane {
  hipStream_t main = nullptr, side = nullptr;
  hipEvent_t handoff = nullptr, ring[4] = {nullptr, nullptr, nullptr, nullptr};
  bool on = false;
  long long last = -1;
  bool begin(hipStream_t caller, bool enable);
  hipStream_t hand() {
    if (!on) return main;
    (void)hipEventRecord(handoff, main);
    (void)hipStreamWaitEvent(side, handoff, 0);
    return side;
  }
  void done(long long k) {
    if (!on) return;
    (void)hipEventRecord(ring[k & 3], side);
    last = k;
  }
  void need(long long k) {  // (a slot the ring has lapped holds a LATER ticket of the same stream: waiting for that one is safe)
    if (on && k >= 0 && k <= last) (void)hipStreamWaitEvent(main, ring[k & 3], 0);
  }
  void end() {
    if (on && last >= 0) (void)hipStreamWaitEvent(main, ring[last & 3], 0);
    on = false;
  }
  ~WgradLane() { end(); }  // error returns included: the caller's stream never runs ahead of work the lane still holds
};
// Deferred join (sr_set_backward_wgrad_deferred): leaves what the lane still holds pending in a per-device record instead of making
// the caller's stream wait for it; sr_backward_lane_join(stream) — any thread — makes `stream` wait for the pending work.
void lane_detach(WgradLane& lane);
bool bn_small_enabled();  // capi.hip: sr_dev_set_bn_small (development switch; default on)
int backward_overlap();  // sr_dev_set_backward_overlap: -1 automatic (small launches only), 0 never, 1 always

#define SR_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::sr::set_error(__VA_ARGS__);        \
      return SR_EINVAL;                    \
    }                                      \
  } while (0)

#define SR_CHECK_LAUNCH(what)                                                   \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      ::sr::set_error("%s: %s", what, hipGetErrorString(e__));                  \
      return SR_ELAUNCH;                                                        \
    }                                                                           \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace sr
