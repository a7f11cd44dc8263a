// Timing-only microbenchmarks for the bf16 conv design (not part of the library):
//   K0 MFMA only; K1 + ds_read_b128 operand traffic of the conv loop; K2 + LDS-DMA refill per chunk.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bf16_ablate.hip -o tools/bf16_ablate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int COT, int PT>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, float* out, int chunks, int lds_stage, size_t src_span) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  f32x16 acc[COT][PT];
  for (int a = 0; a < COT; ++a)
    for (int b = 0; b < PT; ++b)
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  for (int i = tid; i < lds_stage * 2 / 4; i += 256) ((float*)smem)[i] = 0.f;
  __syncthreads();
  bf16x8 ra, rb;
  for (int e = 0; e < 8; ++e) { ra[e] = (__bf16)(float)(lane + e); rb[e] = (__bf16)(float)(wave + e); }
  const int xlane = ((wave * PT) * 34 + j) * 32 + h * 16, wlane = j * 32 + h * 16;
  const int XB = (((4 * PT + 2) * 34 * 32 + 1023) / 1024) * 1024;
  const int units = lds_stage / 1024;
  const char* g = src + ((size_t)blockIdx.x * 65536) % src_span;
  for (int c = 0; c < chunks; ++c) {
    const char* xs = smem + (c & 1) * lds_stage + xlane;
    const char* ws = smem + (c & 1) * lds_stage + XB + wlane;
    if constexpr (MODE >= 2) {
      char* dst = smem + ((c + 1) & 1) * lds_stage;
      const char* s = g + ((size_t)c * lds_stage) % 32768;
      for (int u = wave; u < units; u += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + u * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + u * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      bf16x8 bx[PT + 2], a[3][COT];
      if constexpr (MODE >= 1) {
#pragma unroll
        for (int r = 0; r < PT + 2; ++r) bx[r] = *(const bf16x8*)(xs + (r * 34 + dx) * 32);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int cc = 0; cc < COT; ++cc) a[dy][cc] = *(const bf16x8*)(ws + ((dy * 3 + dx) * COT + cc) * 1024);
      } else {
#pragma unroll
        for (int r = 0; r < PT + 2; ++r) bx[r] = rb;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int cc = 0; cc < COT; ++cc) a[dy][cc] = ra;
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int cc = 0; cc < COT; ++cc)
#pragma unroll
          for (int r = 0; r < PT; ++r) acc[cc][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[dy][cc], bx[r + dy], acc[cc][r], 0, 0, 0);
    }
    if constexpr (MODE >= 2) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < COT; ++a)
    for (int b = 0; b < PT; ++b)
      for (int e = 0; e < 16; ++e) s += acc[a][b][e];
  if (s == 12345.678f) out[tid] = s;
}

template <int MODE, int COT, int PT>
void run(const char* name, int wgs, int chunks, const char* src, float* out, size_t span) {
  const int XB = (((4 * PT + 2) * 34 * 32 + 1023) / 1024) * 1024;
  const int stage = XB + 9 * COT * 1024;
  auto kern = k<MODE, COT, PT>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * stage);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 2 * stage, 0, src, out, chunks, stage, span);
  hipEventRecord(a, 0);
  const int iters = 10;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 2 * stage, 0, src, out, chunks, stage, span);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  ms /= iters;
  const double flops = (double)wgs * 4 * chunks * 9.0 * COT * PT * 32 * 32 * 16 * 2;
  printf("%-34s COT%d PT%d wgs %5d chunks %4d lds %3d KB: %8.1f us %7.1f TF\n", name, COT, PT, wgs, chunks, 2 * stage / 1024, ms * 1e3,
         flops / ms / 1e9);
}

int main() {
  char* src;
  float* out;
  const size_t span = 256u << 20;
  hipMalloc(&src, span + (1 << 20));
  hipMemset(src, 0, span + (1 << 20));
  hipMalloc(&out, 4096);
  for (int wgs : {256, 512, 1024}) {
    run<0, 2, 4>("mfma only", wgs, 200, src, out, span);
    run<1, 2, 4>("mfma + ds_read", wgs, 200, src, out, span);
    run<2, 2, 4>("mfma + ds_read + glds + barrier", wgs, 200, src, out, span);
    run<0, 1, 4>("mfma only", wgs, 200, src, out, span);
    run<1, 1, 4>("mfma + ds_read", wgs, 200, src, out, span);
    run<2, 1, 4>("mfma + ds_read + glds + barrier", wgs, 200, src, out, span);
    run<1, 2, 2>("mfma + ds_read", wgs, 200, src, out, span);
    run<2, 2, 2>("mfma + ds_read + glds + barrier", wgs, 200, src, out, span);
    run<1, 1, 2>("mfma + ds_read", wgs, 200, src, out, span);
    run<2, 1, 2>("mfma + ds_read + glds + barrier", wgs, 200, src, out, span);
  }
  return 0;
}
