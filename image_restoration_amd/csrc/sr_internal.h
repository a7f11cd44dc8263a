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
int rdb_wgrad_bf16(const void* cat, const void* D, long long ns, int n, int h, int w, int nf, int gc, float* const* dparams,
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
int forward_groups();  // image groups of the forward (sr_set_forward_groups; default 1)

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
