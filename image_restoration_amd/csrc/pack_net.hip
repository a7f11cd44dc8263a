// Whole-network weight packing in ONE launch.
//
// A training step re-rounds / re-lays-out every weight after the optimiser update: 351 forward images + biases and 351
// data-gradient images.  Per conv (sr_conv3x3_pack_*: memset + scatter + bias) that is ~1750 launches = 6.9 ms per step —
// 3 % of an fp32 step but 16 % of a bf16 one.  Here a table of PackEntry (one per image, built on the host from the
// network plan, copied into the tail of the packed blob) drives one scatter kernel: a block finds its entry by binary
// search over the entries' first-block prefix, a thread handles one source element exactly as the per-conv kernels do
// (layout.hip / layout_bf16.hip: same index formulas, fp32 CB8 or bf16 CB16).  One memset zeroes all padding first.
#include <vector>

#include "sr_internal.h"

namespace {

__host__ __device__ inline int rup(int v, int a) { return (v + a - 1) / a * a; }
__host__ __device__ inline int group_cout_p(int cout) { return (((cout + 31) / 32 * 32) % 64 == 0) ? 64 : 32; }

template <typename T>
__device__ inline T cvt(float v);
template <>
__device__ inline float cvt<float>(float v) { return v; }
template <>
__device__ inline __bf16 cvt<__bf16>(float v) { return (__bf16)v; }

template <typename T, int CB>
__global__ __launch_bounds__(256) void pack_table_kernel(const sr::PackEntry* __restrict__ table, int n) {
  __shared__ int s_entry;
  if (threadIdx.x == 0) {
    const long long b = blockIdx.x;
    int lo = 0, hi = n - 1;  // last entry with block0 <= b
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (table[mid].block0 <= b) lo = mid; else hi = mid - 1;
    }
    s_entry = lo;
  }
  __syncthreads();
  const sr::PackEntry& e = table[s_entry];
  long long i = ((long long)blockIdx.x - e.block0) * 256 + threadIdx.x;
  if (i >= e.elems) return;
  T* out = (T*)e.out;
  if (e.kind == 2) {  // one step of a transposed dense block (layout.hip: pack_dense_dgrad_kernel)
    const int nfp = rup(e.nf, CB), gcp = rup(e.gc, CB);
    const int slice = e.s == 0 ? e.nf : e.gc;
    const int slice0 = e.s == 0 ? 0 : e.nf + (e.s - 1) * e.gc;
    const int cinp = nfp + (4 - e.s) * gcp;
    for (int k = 5; k > e.s; --k) {
      const int cout_k = k == 5 ? e.nf : e.gc, cin_k = e.nf + (k - 1) * e.gc;
      const long long cnt = (long long)cout_k * slice * 9;
      if (i < cnt) {
        const int tap = (int)(i % 9);
        const int cil = (int)((i / 9) % slice);
        const int co = (int)(i / (9LL * slice));
        float v = e.w[k - 1][((long long)co * cin_k + slice0 + cil) * 9 + tap];
        if (k == 5) v *= e.scale5;
        const int pos = k == 5 ? co : nfp + (4 - k) * gcp + co;
        const int gcw = group_cout_p(slice), cbs = cinp / CB;
        const int g = cil / gcw, col = cil % gcw;
        out[((((long long)g * cbs + pos / CB) * 9 + (8 - tap)) * gcw + col) * CB + pos % CB] = cvt<T>(v);
        return;
      }
      i -= cnt;
    }
    return;
  }
  const long long nw = (long long)e.cout * e.cin * 9;
  if (i >= nw) {  // kind 0: packed bias (fp32, zero-padded)
    const int j = (int)(i - nw);
    e.bout[j] = (e.bias && j < e.cout) ? e.bias[j] : 0.f;
    return;
  }
  const int tap = (int)(i % 9), ci = (int)((i / 9) % e.cin), co = (int)(i / (9LL * e.cin));
  int pos = ci;
  if (ci >= e.first_seg) {
    const int r = ci - e.first_seg;
    pos = rup(e.first_seg, CB) + (r / e.seg) * rup(e.seg, CB) + r % e.seg;
  }
  const float val = e.w[0][i];
  if (e.kind == 0) {
    const int gc = group_cout_p(e.cout), cbs = e.cin_pad / CB;
    const int g = co / gc, col = co % gc;
    out[((((long long)g * cbs + pos / CB) * 9 + tap) * gc + col) * CB + pos % CB] = cvt<T>(val);
  } else {
    const int gc = group_cout_p(e.cin_pad), cbs = (e.cout + CB - 1) / CB;
    const int g = pos / gc, col = pos % gc;
    out[((((long long)g * cbs + co / CB) * 9 + (8 - tap)) * gc + col) * CB + co % CB] = cvt<T>(val);
  }
}

}  // namespace

namespace sr {

size_t pack_table_bytes(size_t entries) { return align_up((entries + 1) * sizeof(PackEntry), 256); }

// entries: host array (kind, pointers, shapes filled; block0/elems computed here).  The images live in
// [blob, blob + image_bytes) and are zeroed first; the table is staged at blob + image_bytes.
int pack_table_run(std::vector<PackEntry>& entries, void* blob, size_t image_bytes, bool bf16, hipStream_t stream) {
  long long blocks = 0;
  for (PackEntry& e : entries) {
    if (e.kind == 2) {
      const int slice = e.s == 0 ? e.nf : e.gc;
      long long total = 0;
      for (int k = 5; k > e.s; --k) total += (long long)(k == 5 ? e.nf : e.gc) * slice * 9;
      e.elems = total;
    } else {
      e.elems = (long long)e.cout * e.cin * 9 + (e.kind == 0 ? (long long)sr_conv3x3_packed_bias_floats(e.cout) : 0);
    }
    e.block0 = blocks;
    blocks += (e.elems + 255) / 256;
  }
  if (entries.empty() || blocks == 0) return SR_OK;
  if (blocks >= (1ll << 31)) {
    set_error("pack_table_run: too many elements");
    return SR_EINVAL;
  }
  if (hipMemsetAsync(blob, 0, image_bytes, stream) != hipSuccess) {
    set_error("pack_table_run: memset failed");
    return SR_ELAUNCH;
  }
  PackEntry* table = (PackEntry*)((char*)blob + image_bytes);
  // pageable source: the runtime stages it before returning, so `entries` may die right after the call
  if (hipMemcpyAsync(table, entries.data(), entries.size() * sizeof(PackEntry), hipMemcpyHostToDevice, stream) != hipSuccess) {
    set_error("pack_table_run: table copy failed");
    return SR_ELAUNCH;
  }
  if (bf16)
    hipLaunchKernelGGL((pack_table_kernel<__bf16, 16>), dim3((unsigned)blocks), dim3(256), 0, stream, table, (int)entries.size());
  else
    hipLaunchKernelGGL((pack_table_kernel<float, 8>), dim3((unsigned)blocks), dim3(256), 0, stream, table, (int)entries.size());
  SR_CHECK_LAUNCH("pack_table");
  return SR_OK;
}

}  // namespace sr
