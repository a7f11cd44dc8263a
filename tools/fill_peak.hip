// What the LDS fill paths of a CU sustain (development probe; build: hipcc --offload-arch=gfx950 -O3 tools/fill_peak.hip -o tools/bin/fill_peak).
//   dma  : buffer_load_dwordx4 ... lds (LDS-DMA, 16 B per lane, 1 KB per instruction) - what every conv / wgrad kernel stages with
//   vgpr : global_load_dwordx4 -> VGPR -> ds_write_b128
// Each workgroup (512 threads) re-reads its own 32 KB window (L2-resident after the first pass: the L2 -> LDS path) or walks a private
// stream of `span` KB (HBM / MALL).  Prints bytes per clock per CU and TB/s for the chip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void fill_kernel(const char* src, long long wg_stride, int span_kb, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 64 KB: 2 x 32 KB stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (long long)blockIdx.x * wg_stride;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)span_kb * 1024u, 0x00020000);
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned window = (unsigned)((it * 32) % span_kb) * 1024u;  // 32 KB per iteration
    char* stage = smem + (it & 1) * 32768;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k)  // 8 waves x 4 x 1 KB = 32 KB
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_p)(stage + (wave * 4 + k) * 1024), 16, lane * 16, window + (wave * 4 + k) * 1024, 0, 0);
      if (DEPTH == 1 || (it & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, window + (wave * 4 + k) * 1024, 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) *(u32x4*)(stage + (wave * 4 + k) * 1024 + lane * 16) = v[k];
    }
    if ((it & 7) == 7) {
      __syncthreads();
      acc += *(unsigned*)(smem + tid * 4);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int iters = 2000;
  char* buf = nullptr;
  const size_t total = (size_t)1024 * 1024 * 1024 * 4;
  if (hipMalloc(&buf, total) != hipSuccess) return 1;
  hipMemset(buf, 1, total);
  unsigned* sink;
  hipMalloc(&sink, 64);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  auto run = [&](const char* name, auto kern, int wgs, int span_kb) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const long long stride = (long long)span_kb * 1024;
    if ((size_t)stride * wgs > total) return;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a, 0);
      hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 65536, 0, buf, stride, span_kb, iters, sink);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)wgs * iters * 32768.0;
    const double clk = prop.clockRate * 1e3;  // Hz (nominal)
    printf("%-28s wgs %4d span %6d KB: %7.3f ms  %6.2f TB/s  %6.1f GB/s per CU  %5.1f B/clk/CU at %.2f GHz nominal\n", name, wgs, span_kb, ms,
           bytes / ms / 1e9, bytes / ms / 1e6 / cus, bytes / (ms * 1e-3) / cus / clk, clk / 1e9);
  };
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
    for (int span : {32, 64, 128, 256, 1024, 4096}) {  // per workgroup: TCP-resident, L2-resident (<= 128 KB x 32 workgroups per XCD), Infinity Cache, HBM
      run("dma  depth1 (wait each)", fill_kernel<0, 1>, cus * wg_per_cu, span);
      run("dma  depth2", fill_kernel<0, 2>, cus * wg_per_cu, span);
      run("vgpr (load + ds_write)", fill_kernel<1, 1>, cus * wg_per_cu, span);
    }
  }
  return 0;
}
