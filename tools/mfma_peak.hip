// Development aid: what the matrix pipe delivers in a bare v_mfma_f32_32x32x16_bf16 loop (operands in registers, random data,
// 4 independent accumulators per wave) at 1, 2 and 4 waves per SIMD — the ceiling the conv kernels' MFMA-busy fractions refer to,
// and the clock the chip holds under it.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(1024) void k(const unsigned* seed, float* out, int iters) {
  unsigned s = seed[threadIdx.x & 63] * 2654435761u + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      s = s * 1664525u + 1013904223u;
      a[i][j] = (__bf16)(((int)(s >> 8) & 0xffff) / 65536.f - 0.5f);
      s = s * 1664525u + 1013904223u;
      b[i][j] = (__bf16)(((int)(s >> 8) & 0xffff) / 65536.f - 0.5f);
    }
  f32x16 acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
  }
  float r = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) r += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ __launch_bounds__(1024) void kf(const unsigned* seed, float* out, int iters) {
  unsigned s = seed[threadIdx.x & 63] * 2654435761u + threadIdx.x;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    s = s * 1664525u + 1013904223u;
    a[i] = ((int)(s >> 8) & 0xffff) / 65536.f - 0.5f;
    s = s * 1664525u + 1013904223u;
    b[i] = ((int)(s >> 8) & 0xffff) / 65536.f - 0.5f;
  }
  f32x16 acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
  }
  float r = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) r += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  unsigned h[64];
  for (int i = 0; i < 64; ++i) h[i] = rand();
  unsigned* d;
  float* o;
  hipMalloc(&d, sizeof(h));
  hipMalloc(&o, 256 * 1024 * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  for (int threads : {256, 512, 1024}) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<<<cus, threads>>>(d, o, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) k<<<cus, threads>>>(d, o, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * cus * (threads / 64) * (double)iters * 16 * 32768.0;
    printf("%d CUs, %d waves per SIMD: %.1f ms, %.0f TFLOP/s = %.3f of 2500; cycles per MFMA at 2.4 GHz would be %.1f\n", cus, threads / 256, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 2500, ms * 1e-3 * 2.4e9 / (5.0 * iters * 16 * (threads / 256)));
  }
  for (int threads : {256, 512}) {
    const int iters = 10000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kf<<<cus, threads>>>(d, o, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) kf<<<cus, threads>>>(d, o, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * cus * (threads / 64) * (double)iters * 16 * 4096.0;
    printf("fp32 32x32x2: %d waves per SIMD: %.1f ms, %.1f TFLOP/s = %.3f of 157.3\n", threads / 256, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
  }
  return 0;
}
