// BatchNorm2d (+ LeakyReLU) of a SMALL activation in ONE launch (included by train_ops.hip: fp32 / CB8, and disc_bf16.hip: bf16 / CB16).
// The general path is two-stage and deterministic: statistics = reduce + finalise launches (twice in the forward: mean, then the
// centred second moment), then the apply pass — 5 launches forward, 3 backward, each 5-20 us of mostly launch latency when the tensor
// is the 16 x 16 ... 4 x 4 level of VGGStyleDiscriminator128 (discriminator_arch.py:23-49) on a batch of 32: the reference recipe runs
// 117 such reductions per step (2.4 ms of a 30 ms bf16 step, profiles/r03_recipe_*).  Below n*h*w <= kBnSmallPixels a workgroup of
// 1024 threads owns FOUR channels for the whole pass: it walks the pixels three times (forward: sum, centred squares, apply; the tensor
// is L2-resident) or twice (backward), reduces in a fixed order (wave shuffle tree, then the 16 wave results in order: bit-reproducible
// like the general path, not bit-identical to it), and also writes the batch statistics / running buffers / dgamma, dbeta.
// Same arithmetic per element as bn_lrelu_{fwd,bwd}_kernel / bn_lrelu16_kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace bnsmall {
constexpr long long kBnSmallPixels = 32768;

template <typename T>
struct Params {
  const T* x;
  const T* dy;   // backward
  const T* y;    // backward: the forward output (LeakyReLU mask)
  T* out;        // forward: y; backward: dx
  long long x_ns, dy_ns, y_ns, out_ns;  // image strides in elements
  int n, c, hw;
  const float* gamma;
  const float* beta;
  float* mean;     // forward: written; backward: read
  float* invstd;
  float* running_mean;  // forward, may be null
  float* running_var;
  float* dgamma;  // backward: written
  float* dbeta;
  float momentum, eps, slope;
};

// four consecutive channels of a pixel: one 16-byte (fp32) / 8-byte (bf16) access
struct F4 {
  float v[4];
};
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ F4 ld4(const float* p) {
  const f32x4_t t = *(const f32x4_t*)p;
  return F4{{t[0], t[1], t[2], t[3]}};
}
__device__ __forceinline__ F4 ld4(const __bf16* p) {
  const bf16x4_t t = *(const bf16x4_t*)p;
  return F4{{(float)t[0], (float)t[1], (float)t[2], (float)t[3]}};
}
__device__ __forceinline__ void st4(float* p, const F4& a) { *(f32x4_t*)p = f32x4_t{a.v[0], a.v[1], a.v[2], a.v[3]}; }
__device__ __forceinline__ void st4(__bf16* p, const F4& a) {
  *(bf16x4_t*)p = bf16x4_t{(__bf16)a.v[0], (__bf16)a.v[1], (__bf16)a.v[2], (__bf16)a.v[3]};
}

// sum of v over the 1024 threads of the workgroup in a fixed order; the result reaches every thread
__device__ __forceinline__ float wg_sum(float v, float* sh /* [16] */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += sh[k];
  return s;
}

// CBW channels per block of the layout (8: fp32 CB8, 16: bf16 CB16); grid.x = cblocks * CBW / 4
template <typename T, int CBW, bool BWD>
__global__ __launch_bounds__(1024) void bn_small_kernel(const Params<T> p) {
  __shared__ float sh[16];
  constexpr int G = CBW / 4;
  const int cb = blockIdx.x / G, g = blockIdx.x - cb * G;
  const int ch0 = cb * CBW + g * 4;
  const long long total = (long long)p.n * p.hw;
  const float inv_count = 1.f / (float)total;
  auto at = [&](const T* base, long long ns, long long i) {
    const int n = (int)(i / p.hw);
    return base + n * ns + ((long long)cb * p.hw + (i - (long long)n * p.hw)) * CBW + g * 4;
  };
  float mu[4], is[4], ga[4], be[4];
  bool live[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    live[e] = ch0 + e < p.c;
    ga[e] = live[e] ? p.gamma[ch0 + e] : 0.f;
    be[e] = (!BWD && live[e]) ? p.beta[ch0 + e] : 0.f;
    mu[e] = (BWD && live[e]) ? p.mean[ch0 + e] : 0.f;
    is[e] = (BWD && live[e]) ? p.invstd[ch0 + e] : 0.f;
  }
  if constexpr (!BWD) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long i = threadIdx.x; i < total; i += 1024) {
      const F4 xv = ld4(at(p.x, p.x_ns, i));
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += xv.v[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) mu[e] = wg_sum(a[e], sh) * inv_count;
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = 0.f;
    for (long long i = threadIdx.x; i < total; i += 1024) {
      const F4 xv = ld4(at(p.x, p.x_ns, i));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = xv.v[e] - mu[e];
        a[e] += d * d;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ss = wg_sum(a[e], sh);
      const float var = ss * inv_count;
      is[e] = rsqrtf(var + p.eps);
      if (threadIdx.x == 0 && live[e]) {
        p.mean[ch0 + e] = mu[e];
        p.invstd[ch0 + e] = is[e];
        if (p.running_mean) {  // nn.BatchNorm2d: running = (1 - m) running + m batch, with the UNBIASED variance
          const float unbiased = total > 1 ? ss / (float)(total - 1) : var;
          p.running_mean[ch0 + e] = (1.f - p.momentum) * p.running_mean[ch0 + e] + p.momentum * mu[e];
          p.running_var[ch0 + e] = (1.f - p.momentum) * p.running_var[ch0 + e] + p.momentum * unbiased;
        }
      }
    }
    for (long long i = threadIdx.x; i < total; i += 1024) {
      const F4 xv = ld4(at(p.x, p.x_ns, i));
      F4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.f;
        if (live[e]) {
          v = (xv.v[e] - mu[e]) * is[e] * ga[e] + be[e];
          v = v > 0.f ? v : v * p.slope;
        }
        o.v[e] = v;
      }
      st4((T*)at(p.out, p.out_ns, i), o);
    }
  } else {
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long i = threadIdx.x; i < total; i += 1024) {
      const F4 xv = ld4(at(p.x, p.x_ns, i)), gv = ld4(at(p.dy, p.dy_ns, i)), yv = ld4(at(p.y, p.y_ns, i));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dz = yv.v[e] > 0.f ? gv.v[e] : gv.v[e] * p.slope;
        a[e] += dz;
        b[e] += dz * (xv.v[e] - mu[e]) * is[e];
      }
    }
    float db[4], dg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      db[e] = wg_sum(a[e], sh);
      dg[e] = wg_sum(b[e], sh);
      if (threadIdx.x == 0 && live[e]) {
        p.dbeta[ch0 + e] = db[e];
        p.dgamma[ch0 + e] = dg[e];
      }
    }
    for (long long i = threadIdx.x; i < total; i += 1024) {
      const F4 xv = ld4(at(p.x, p.x_ns, i)), gv = ld4(at(p.dy, p.dy_ns, i)), yv = ld4(at(p.y, p.y_ns, i));
      F4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.f;
        if (live[e]) {
          const float dz = yv.v[e] > 0.f ? gv.v[e] : gv.v[e] * p.slope;
          const float xhat = (xv.v[e] - mu[e]) * is[e];
          v = ga[e] * is[e] * (dz - (db[e] + xhat * dg[e]) * inv_count);
        }
        o.v[e] = v;
      }
      st4((T*)at(p.out, p.out_ns, i), o);
    }
  }
}

template <typename T, int CBW, bool BWD>
inline void launch(const Params<T>& p, hipStream_t stream) {
  const int cblocks = (p.c + CBW - 1) / CBW;
  hipLaunchKernelGGL((bn_small_kernel<T, CBW, BWD>), dim3(cblocks * (CBW / 4)), dim3(1024), 0, stream, p);
}
}  // namespace bnsmall
