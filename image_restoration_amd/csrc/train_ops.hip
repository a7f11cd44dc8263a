// HBM-bound training ops of the ESRGAN step (everything that is not a convolution):
//   BatchNorm2d(+LeakyReLU) forward/backward of VGGStyleDiscriminator128 (discriminator_arch.py:23-49),
//   nn.Linear forward/backward (:45-46), L1Loss (losses.py:80-106), the relativistic vanilla GAN loss
//   (BCEWithLogits, losses.py:379-380 under esrgan_model.py:40-41,67,71), Adam (base_model.py:78-83) and
//   EMA (base_model.py:50-57).
// All reductions are two-stage and deterministic (fixed grid, fixed summation order); no atomics.
#include "sr_internal.h"
#include "bn_small.h"

namespace {

constexpr int RED_SPLITS = 64;  // blocks per reduced quantity

__device__ __forceinline__ float block_sum(float v, float* sh) {
  // 256 threads: wave shuffle then 4 partials through LDS
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------------ BatchNorm on CB8
// per-channel sums over N*H*W of f(x): mode 0: x ; mode 1: (x-mean)^2 ; mode 2: dz (and dz*xhat) for backward.
// grid (cblocks, RED_SPLITS), block 256.  Thread t covers float4 `half = t&1` of pixels t>>1, t>>1 + 128, ...
struct BnRedParams {
  const float* x;
  const float* dy;
  const float* y;
  const float* mean;
  const float* invstd;
  float* part;  // [cblocks][RED_SPLITS][8][2]
  long long x_ns, dy_ns, y_ns;
  int n, hw, mode;
  float slope;
};

__global__ __launch_bounds__(256) void bn_reduce_kernel(const BnRedParams p) {
  __shared__ float sh[4];
  const int cb = blockIdx.x, sp = blockIdx.y;
  const int half = threadIdx.x & 1;
  const long long total = (long long)p.n * p.hw;
  float a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
  float4 mu = make_float4(0, 0, 0, 0), is = make_float4(1, 1, 1, 1);
  if (p.mode >= 1) mu = *(const float4*)(p.mean + cb * 8 + half * 4);
  if (p.mode == 2) is = *(const float4*)(p.invstd + cb * 8 + half * 4);
  for (long long i = (long long)sp * 128 + (threadIdx.x >> 1); i < total; i += 128 * RED_SPLITS) {
    const int n = (int)(i / p.hw);
    const long long off = ((long long)cb * p.hw + (i - (long long)n * p.hw)) * 8 + half * 4;
    const float4 xv = *(const float4*)(p.x + n * p.x_ns + off);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    const float ms[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w};
    if (p.mode == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += xs[e];
    } else if (p.mode == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += (xs[e] - ms[e]) * (xs[e] - ms[e]);
    } else {
      const float4 gv = *(const float4*)(p.dy + n * p.dy_ns + off);
      const float4 yv = *(const float4*)(p.y + n * p.y_ns + off);
      const float gs[4] = {gv.x, gv.y, gv.z, gv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dz = ys[e] > 0.f ? gs[e] : gs[e] * p.slope;  // LeakyReLU backward from the saved output
        a[e] += dz;
        b[e] += dz * (xs[e] - ms[e]) * iv[e];
      }
    }
  }
  // reduce the two parities separately: even threads hold channels 0-3, odd threads 4-7
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float s0 = block_sum(half == 0 ? a[e] : 0.f, sh), s1 = block_sum(half == 1 ? a[e] : 0.f, sh);
    const float t0 = block_sum(half == 0 ? b[e] : 0.f, sh), t1 = block_sum(half == 1 ? b[e] : 0.f, sh);
    if (threadIdx.x == 0) {
      float* o = p.part + (((long long)cb * RED_SPLITS + sp) * 8) * 2;
      o[e * 2] = s0;
      o[e * 2 + 1] = t0;
      o[(4 + e) * 2] = s1;
      o[(4 + e) * 2 + 1] = t1;
    }
  }
}

// one thread per channel: sums the partials.  which 0: mean ; 1: var -> invstd (+ running stats) ; 2: dbeta/dgamma
__global__ void bn_finalize_kernel(const float* part, int c, int which, long long count, float eps, float momentum,
                                   float* mean, float* invstd, float* running_mean, float* running_var, float* dgamma,
                                   float* dbeta, const float* gamma_scale_unused) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const int cb = ch >> 3, e = ch & 7;
  float s = 0.f, t = 0.f;
  for (int k = 0; k < RED_SPLITS; ++k) {
    const float* o = part + (((long long)cb * RED_SPLITS + k) * 8 + e) * 2;
    s += o[0];
    t += o[1];
  }
  if (which == 0) {
    mean[ch] = s / (float)count;
  } else if (which == 1) {
    const float var = s / (float)count;  // biased, used for normalisation
    invstd[ch] = rsqrtf(var + eps);
    if (running_mean) {
      // nn.BatchNorm2d: running = (1-m)*running + m*batch, with the UNBIASED variance
      const float unbiased = count > 1 ? s / (float)(count - 1) : var;
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean[ch];
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
    }
  } else {
    dbeta[ch] = s;
    dgamma[ch] = t;
  }
}

// y = lrelu((x - mean)*invstd*gamma + beta); pad channels (>= c) are written as zero.
__global__ void bn_lrelu_fwd_kernel(const float* __restrict__ x, long long x_ns, float* __restrict__ y, long long y_ns,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, float slope, int c,
                                    int cblocks, int hw, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int pix = (int)(r % hw);
  r /= hw;
  const int cb = (int)(r % cblocks), n = (int)(r / cblocks);
  const long long off = ((long long)cb * hw + pix) * 8 + half * 4;
  const float4 xv = *(const float4*)(x + n * x_ns + off);
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = cb * 8 + half * 4 + e;
    float v = 0.f;
    if (ch < c) {
      v = (xs[e] - mean[ch]) * invstd[ch] * gamma[ch] + beta[ch];
      v = v > 0.f ? v : v * slope;
    }
    o[e] = v;
  }
  *(float4*)(y + n * y_ns + off) = make_float4(o[0], o[1], o[2], o[3]);
}

// dx = gamma*invstd*(dz - [train] (dbeta + xhat*dgamma)/M),  dz = dy * lrelu'(y)
__global__ void bn_lrelu_bwd_kernel(const float* __restrict__ x, long long x_ns, const float* __restrict__ dy,
                                    long long dy_ns, const float* __restrict__ y, long long y_ns, float* __restrict__ dx,
                                    long long dx_ns, const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                    const float* __restrict__ dbeta, float slope, int train, float inv_count, int c,
                                    int cblocks, int hw, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int pix = (int)(r % hw);
  r /= hw;
  const int cb = (int)(r % cblocks), n = (int)(r / cblocks);
  const long long off = ((long long)cb * hw + pix) * 8 + half * 4;
  const float4 xv = *(const float4*)(x + n * x_ns + off), gv = *(const float4*)(dy + n * dy_ns + off),
               yv = *(const float4*)(y + n * y_ns + off);
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = cb * 8 + half * 4 + e;
    float v = 0.f;
    if (ch < c) {
      const float dz = ys[e] > 0.f ? gs[e] : gs[e] * slope;
      const float xhat = (xs[e] - mean[ch]) * invstd[ch];
      v = train ? gamma[ch] * invstd[ch] * (dz - (dbeta[ch] + xhat * dgamma[ch]) * inv_count)
                : gamma[ch] * invstd[ch] * dz;
    }
    o[e] = v;
  }
  *(float4*)(dx + n * dx_ns + off) = make_float4(o[0], o[1], o[2], o[3]);
}

// invstd from running_var for eval mode; mean = running_mean
__global__ void bn_eval_prep_kernel(const float* rm, const float* rv, float eps, int c, float* mean, float* invstd) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch < c) {
    mean[ch] = rm[ch];
    invstd[ch] = rsqrtf(rv[ch] + eps);
  }
}

// ------------------------------------------------------------------ Linear (small: in <= 16K, out <= 256)
// y[n][o] = b[o] + sum_i x[n][i] w[o][i]      grid (out, n), block 256
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int in,
                                                         int out, float slope) {
  __shared__ float sh[4];
  const int o = blockIdx.x, n = blockIdx.y;
  float s = 0.f;
  for (int i = threadIdx.x; i < in; i += 256) s += x[(long long)n * in + i] * w[(long long)o * in + i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    float v = s + (b ? b[o] : 0.f);
    y[(long long)n * out + o] = v > 0.f ? v : v * slope;
  }
}
// dz[n][o] = dy * lrelu'(y) (in place into dz);  dx[n][i] = sum_o dz[n][o] w[o][i]      grid (ceil(in/256), n)
__global__ void linear_bwd_x_kernel(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ dx,
                                    int in, int out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (i >= in) return;
  float s = 0.f;
  for (int o = 0; o < out; ++o) s += dz[(long long)n * out + o] * w[(long long)o * in + i];
  dx[(long long)n * in + i] = s;
}
// dw[o][i] = sum_n dz[n][o] x[n][i] ; db[o] = sum_n dz[n][o]      grid (ceil(in/256), out)
__global__ void linear_bwd_w_kernel(const float* __restrict__ dz, const float* __restrict__ x, float* __restrict__ dw,
                                    float* __restrict__ db, int in, int out, int nb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, o = blockIdx.y;
  if (i < in) {
    float s = 0.f;
    for (int n = 0; n < nb; ++n) s += dz[(long long)n * out + o] * x[(long long)n * in + i];
    dw[(long long)o * in + i] = s;
  }
  if (db && blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int n = 0; n < nb; ++n) s += dz[(long long)n * out + o];
    db[o] = s;
  }
}
__global__ void lrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz,
                                 float slope, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) dz[i] = y[i] > 0.f ? dy[i] : dy[i] * slope;
}

// ------------------------------------------------------------------ flat reductions / losses
// mode 0: sum x ; 1: sum |x - t| (L1) ; 2: sum softplus(sign*(x - shift)) (BCE-with-logits vs target 1: sign=-1, 0: +1)
// mode 3: sum sigmoid-based derivative d/dx of mode 2 (for the gradient through the mean of the other logits)
// mode 4: sum x*t (dot product) ; 5: sum (x - t)^2 (MSE) ; 6: sum sqrt((x - t)^2 + sign) (Charbonnier, eps in `sign`)
// mode 7: sum (x - sign)^2 (least-squares GAN, label in `sign`) ; 8: sum relu(1 + sign*x) (hinge) ;
// mode 9: sum softplus(x) - sign*x (BCE-with-logits against the soft label `sign`)
__device__ __forceinline__ float softplus(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }
__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }

struct FlatRedParams {
  const float* x;
  const float* t;
  const float* shift;  // device scalar or null
  float* part;         // [RED_SPLITS*4]
  long long n;
  int mode;
  float sign;
};

__global__ __launch_bounds__(256) void flat_reduce_kernel(const FlatRedParams p) {
  __shared__ float sh[4];
  const float shift = p.shift ? *p.shift : 0.f;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.n; i += 256LL * gridDim.x) {
    const float v = p.x[i];
    if (p.mode == 0) s += v;
    else if (p.mode == 1) s += fabsf(v - p.t[i]);
    else if (p.mode == 2) s += softplus(p.sign * (v - shift));
    else if (p.mode == 4) s += v * p.t[i];
    else if (p.mode == 5) s += (v - p.t[i]) * (v - p.t[i]);
    else if (p.mode == 6) s += sqrtf((v - p.t[i]) * (v - p.t[i]) + p.sign);
    else if (p.mode == 7) s += (v - p.sign) * (v - p.sign);
    else if (p.mode == 8) s += fmaxf(1.f + p.sign * v, 0.f);
    else if (p.mode == 9) s += softplus(v) - p.sign * v;
    else s += p.sign * sigmoidf(p.sign * (v - shift));
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) p.part[blockIdx.x] = s;
}
// out[0] = scale * sum(part[0..nparts))
__global__ void flat_finalize_kernel(const float* part, int nparts, float scale, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += part[k];
    out[0] = s * scale;
  }
}
// L1 backward: dx = g * scale * sign(x - t)
__global__ void l1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ g,
                              float scale, float* __restrict__ dx, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float d = x[i] - t[i];
    dx[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * scale * g[0];
  }
}
// MSE / Charbonnier backward: dx = g * scale * 2 (x - t)   or   g * scale * (x - t) / sqrt((x - t)^2 + eps)
__global__ void pixel_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ g,
                                 float scale, int kind, float eps, float* __restrict__ dx, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float d = x[i] - t[i];
    dx[i] = (kind == 1 ? 2.f * d : d / sqrtf(d * d + eps)) * scale * g[0];
  }
}
// point-wise GAN criteria, backward: kind 1 least squares 2 (x - c); 2 linear c; 3 softplus(c x): c sigmoid(c x);
// 4 hinge relu(1 + c x): c [1 + c x > 0]; 5 BCE against the soft label c: sigmoid(x) - c
__global__ void gan_point_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, int kind, float c, float scale,
                                     float* __restrict__ dx, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  float d;
  if (kind == 1) d = 2.f * (v - c);
  else if (kind == 2) d = c;
  else if (kind == 3) d = c * sigmoidf(c * v);
  else if (kind == 4) d = (1.f + c * v > 0.f) ? c : 0.f;
  else d = sigmoidf(v) - c;
  dx[i] = d * scale * g[0];
}
// BCE backward wrt x: dx = g * scale * sign*sigmoid(sign*(x-shift))
__global__ void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ g,
                               float sign, float scale, float* __restrict__ dx, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = sign * sigmoidf(sign * (x[i] - (shift ? *shift : 0.f))) * scale * g[0];
}
// fill: dx[i] = scale * g[0] * s[0]   (gradient of a mean that was subtracted from every logit)
__global__ void fill_scaled_kernel(const float* __restrict__ g, const float* __restrict__ s, float scale,
                                   float* __restrict__ dx, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = scale * g[0] * s[0];
}

// ------------------------------------------------------------------ optimiser
// torch.optim.Adam (amsgrad=False, maximize=False): g += wd*p; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd,
                            float bc1, float bc2_sqrt, float grad_scale, const int32_t* __restrict__ abort_word,
                            int32_t* __restrict__ skipped) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (abort_word && *abort_word != 0) {  // a launch of this step timed out: its gradients are invalid, the state stays as it is
    if (i == 0 && skipped) *skipped += 1;
    return;
  }
  if (i >= n) return;
  float gi = g[i] * grad_scale;
  const float pi = p[i];
  if (wd != 0.f) gi += wd * pi;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = pi - (lr / bc1) * (mi / denom);
}
__global__ void axpby_kernel(float* __restrict__ dst, const float* __restrict__ src, float a, float b, long long n,
                             const int32_t* __restrict__ abort_word) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (abort_word && *abort_word != 0) return;
  if (i < n) dst[i] = a * dst[i] + b * src[i];
}

// ------------------------------------------------------------------ bilinear x2 (align_corners=False) on CB8
// F.interpolate(scale_factor=2, mode='bilinear', align_corners=False): source coordinate (o+0.5)/2-0.5 clamped at 0,
// second tap clamped at size-1.  Both directions are written as gathers (deterministic).
__device__ __forceinline__ void bil_taps(int o, int size, int& i0, int& i1, float& w0, float& w1) {
  float s = (o + 0.5f) * 0.5f - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < size - 1 ? 1 : 0);
  w1 = s - (float)i0;
  w0 = 1.f - w1;
}
__global__ void bilinear2x_fwd_kernel(const float* __restrict__ src, long long src_ns, float* __restrict__ dst,
                                      long long dst_ns, int cblocks, int h, int w, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int W2 = 2 * w, H2 = 2 * h;
  const int ox = (int)(r % W2);
  r /= W2;
  const int oy = (int)(r % H2);
  r /= H2;
  const int cb = (int)(r % cblocks), n = (int)(r / cblocks);
  int y0, y1, x0, x1;
  float wy0, wy1, wx0, wx1;
  bil_taps(oy, h, y0, y1, wy0, wy1);
  bil_taps(ox, w, x0, x1, wx0, wx1);
  const float* b = src + n * src_ns + (long long)cb * h * w * 8 + half * 4;
  const float4 a00 = *(const float4*)(b + ((long long)y0 * w + x0) * 8), a01 = *(const float4*)(b + ((long long)y0 * w + x1) * 8),
               a10 = *(const float4*)(b + ((long long)y1 * w + x0) * 8), a11 = *(const float4*)(b + ((long long)y1 * w + x1) * 8);
  float4 o;
  o.x = wy0 * (wx0 * a00.x + wx1 * a01.x) + wy1 * (wx0 * a10.x + wx1 * a11.x);
  o.y = wy0 * (wx0 * a00.y + wx1 * a01.y) + wy1 * (wx0 * a10.y + wx1 * a11.y);
  o.z = wy0 * (wx0 * a00.z + wx1 * a01.z) + wy1 * (wx0 * a10.z + wx1 * a11.z);
  o.w = wy0 * (wx0 * a00.w + wx1 * a01.w) + wy1 * (wx0 * a10.w + wx1 * a11.w);
  *(float4*)(dst + n * dst_ns + (((long long)cb * H2 + oy) * W2 + ox) * 8 + half * 4) = o;
}
// gsrc[y][x] = sum over the (at most 4x4) outputs whose taps touch (y, x)
__global__ void bilinear2x_bwd_kernel(const float* __restrict__ g, long long g_ns, float* __restrict__ gsrc,
                                      long long gsrc_ns, int cblocks, int h, int w, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int x = (int)(r % w);
  r /= w;
  const int y = (int)(r % h);
  r /= h;
  const int cb = (int)(r % cblocks), n = (int)(r / cblocks);
  const int W2 = 2 * w, H2 = 2 * h;
  const float* b = g + n * g_ns + (long long)cb * H2 * W2 * 8 + half * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int oy = 2 * y - 1; oy <= 2 * y + 2; ++oy) {
    if (oy < 0 || oy >= H2) continue;
    int y0, y1;
    float wy0, wy1;
    bil_taps(oy, h, y0, y1, wy0, wy1);
    const float wy = (y0 == y ? wy0 : 0.f) + (y1 == y ? wy1 : 0.f);
    if (wy == 0.f) continue;
    for (int ox = 2 * x - 1; ox <= 2 * x + 2; ++ox) {
      if (ox < 0 || ox >= W2) continue;
      int x0, x1;
      float wx0, wx1;
      bil_taps(ox, w, x0, x1, wx0, wx1);
      const float wgt = wy * ((x0 == x ? wx0 : 0.f) + (x1 == x ? wx1 : 0.f));
      if (wgt == 0.f) continue;
      const float4 v = *(const float4*)(b + ((long long)oy * W2 + ox) * 8);
      acc.x += wgt * v.x;
      acc.y += wgt * v.y;
      acc.z += wgt * v.z;
      acc.w += wgt * v.w;
    }
  }
  *(float4*)(gsrc + n * gsrc_ns + (((long long)cb * h + y) * w + x) * 8 + half * 4) = acc;
}

// ------------------------------------------------------------------ spectral norm (torch.nn.utils.spectral_norm)
// W viewed as [rows][cols] row-major.  t = W^T u : one thread per column.
// t = W^T u in row chunks: block (x, y) sums rows [y*chunk, (y+1)*chunk) for 256 columns into t[y][cols]; the single-block
// normalisation that follows adds the chunks in fixed order (a column walk over all rows on cols/256 workgroups took 70 us)
__global__ void sn_wt_u_kernel(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ t, int rows,
                               int cols, int chunk) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * chunk, r1 = min(r0 + chunk, rows);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += W[(long long)r * cols + c] * u[r];
  t[(long long)blockIdx.y * cols + c] = s;
}
// s = W v : one block per row
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* __restrict__ W, const float* __restrict__ v,
                                                     float* __restrict__ s, int cols) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  float a = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) a += W[(long long)r * cols + c] * v[c];
  a = block_sum(a, sh);
  if (threadIdx.x == 0) s[r] = a;
}
// single block: out = x / max(||x||, eps); if sigma: sigma[0] = out . x
__global__ __launch_bounds__(256) void sn_normalize_kernel(float* __restrict__ x, float* __restrict__ out, int n,
                                                           float eps, float* sigma, int parts) {
  __shared__ float sh[4];
  if (parts > 1) {  // x holds `parts` partial vectors behind one another: fold them into the first
    for (int i = threadIdx.x; i < n; i += 256) {
      float s = 0.f;
      for (int y = 0; y < parts; ++y) s += x[(long long)y * n + i];
      x[i] = s;
    }
    __syncthreads();
  }
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += x[i] * x[i];
  a = block_sum(a, sh);
  const float nrm = fmaxf(sqrtf(a), eps);
  for (int i = threadIdx.x; i < n; i += 256) out[i] = x[i] / nrm;
  if (sigma && threadIdx.x == 0) sigma[0] = a / nrm;
}
// single block: sigma = u . s   (eval mode: stored u, fresh s = W v)
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                     float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += a[i] * b[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = s;
}
__global__ void sn_scale_kernel(const float* __restrict__ W, const float* __restrict__ sigma, float* __restrict__ out,
                                long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = W[i] / sigma[0];
}
// dW_orig = (G - dot * u[r] v[c]) / sigma
__global__ void sn_bwd_kernel(const float* __restrict__ G, const float* __restrict__ u, const float* __restrict__ v,
                              const float* __restrict__ sigma, const float* __restrict__ dot, float* __restrict__ out,
                              int cols, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (G[i] - dot[0] * u[i / cols] * v[i % cols]) / sigma[0];
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) {
    const float4 x = ((const float4*)a)[i], y = ((const float4*)b)[i];
    ((float4*)out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

// sum over the cropped region of (round(clamp(a,0,1)*255) - round(clamp(b,0,1)*255))^2 for image n = blockIdx.y
__global__ __launch_bounds__(256) void psnr_sse_kernel(const float* __restrict__ a, const float* __restrict__ b, int c, int h,
                                                       int w, int crop, float* __restrict__ part) {
  __shared__ float sh[4];
  const int n = blockIdx.y;
  const int hh = h - 2 * crop, ww = w - 2 * crop;
  const long long total = (long long)c * hh * ww;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += 256LL * gridDim.x) {
    const int x = (int)(i % ww), y = (int)((i / ww) % hh), ch = (int)(i / ((long long)ww * hh));
    const long long idx = (((long long)n * c + ch) * h + (y + crop)) * w + (x + crop);
    const float qa = rintf(fminf(fmaxf(a[idx], 0.f), 1.f) * 255.f), qb = rintf(fminf(fmaxf(b[idx], 0.f), 1.f) * 255.f);
    s += (qa - qb) * (qa - qb);
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[(long long)n * gridDim.x + blockIdx.x] = s;
}
__global__ void psnr_finalize_kernel(const float* part, int nparts, float* out) {
  const int n = blockIdx.x;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += part[(long long)n * nparts + k];
    out[n] = s;
  }
}

// SSIM map sum (psnr_ssim.py:49-83): 11x11 Gaussian (sigma 1.5) statistics over the valid region of the border-cropped,
// uint8-quantised channel; one thread per output pixel, 121 taps, float accumulation of integers <= 255^2 weighted by a
// normalised window (relative error ~1e-6, the reference computes in float64).
struct SsimWin {
  float w[11];
};
__global__ __launch_bounds__(256) void ssim_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, int c, int h,
                                                       int w, int crop, SsimWin win, float* __restrict__ part) {
  __shared__ float sh[4];
  const int n = blockIdx.y;
  const int hh = h - 2 * crop - 10, ww = w - 2 * crop - 10;  // valid region of the cropped image
  const long long total = (long long)c * hh * ww;
  const float C1 = 6.5025f, C2 = 58.5225f;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += 256LL * gridDim.x) {
    const int x = (int)(i % ww), y = (int)((i / ww) % hh), ch = (int)(i / ((long long)ww * hh));
    const float* pa = a + (((long long)n * c + ch) * h + (y + crop)) * w + (x + crop);
    const float* pb = b + (((long long)n * c + ch) * h + (y + crop)) * w + (x + crop);
    float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
    for (int dy = 0; dy < 11; ++dy) {
      float r1 = 0.f, r2 = 0.f, r11 = 0.f, r22 = 0.f, r12 = 0.f;
#pragma unroll
      for (int dx = 0; dx < 11; ++dx) {
        const float qa = rintf(fminf(fmaxf(pa[dy * w + dx], 0.f), 1.f) * 255.f);
        const float qb = rintf(fminf(fmaxf(pb[dy * w + dx], 0.f), 1.f) * 255.f);
        const float g = win.w[dx];
        r1 += g * qa;
        r2 += g * qb;
        r11 += g * qa * qa;
        r22 += g * qb * qb;
        r12 += g * qa * qb;
      }
      const float g = win.w[dy];
      m1 += g * r1;
      m2 += g * r2;
      s11 += g * r11;
      s22 += g * r22;
      s12 += g * r12;
    }
    const float v1 = s11 - m1 * m1, v2 = s22 - m2 * m2, cv = s12 - m1 * m2;
    s += ((2.f * m1 * m2 + C1) * (2.f * cv + C2)) / ((m1 * m1 + m2 * m2 + C1) * (v1 + v2 + C2));
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[(long long)n * gridDim.x + blockIdx.x] = s;
}

// ------------------------------------------------------------------ VGG feature extractor helpers (PerceptualLoss)
// nn.MaxPool2d(kernel_size=2, stride=2) on CB8 (vgg_arch.py:131-137), floor mode; thread per output float4.
__global__ void maxpool2x2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int cblocks, int h, int w,
                                      long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int oh = h / 2, ow = w / 2;
  const int ox = (int)(r % ow);
  r /= ow;
  const int oy = (int)(r % oh);
  const long long ncb = r / oh;  // n * cblocks + cb
  const float* b = x + ((ncb * h + 2 * oy) * w + 2 * ox) * 8 + half * 4;
  const float4 a00 = *(const float4*)b, a01 = *(const float4*)(b + 8), a10 = *(const float4*)(b + (long long)w * 8),
               a11 = *(const float4*)(b + (long long)w * 8 + 8);
  float4 o;
  o.x = fmaxf(fmaxf(a00.x, a01.x), fmaxf(a10.x, a11.x));
  o.y = fmaxf(fmaxf(a00.y, a01.y), fmaxf(a10.y, a11.y));
  o.z = fmaxf(fmaxf(a00.z, a01.z), fmaxf(a10.z, a11.z));
  o.w = fmaxf(fmaxf(a00.w, a01.w), fmaxf(a10.w, a11.w));
  *(float4*)(y + ((ncb * oh + oy) * ow + ox) * 8 + half * 4) = o;
}
// dx: the gradient of a window goes to its first maximum in scan order (torch's choice); thread per INPUT float4
__global__ void maxpool2x2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int cblocks,
                                      int h, int w, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int xx = (int)(r % w);
  r /= w;
  const int yy = (int)(r % h);
  const long long ncb = r / h;
  const int oh = h / 2, ow = w / 2;
  const int oy = yy >> 1, ox = xx >> 1;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (oy < oh && ox < ow) {
    const float* b = x + ((ncb * h + 2 * oy) * w + 2 * ox) * 8 + half * 4;
    const float4 v[4] = {*(const float4*)b, *(const float4*)(b + 8), *(const float4*)(b + (long long)w * 8),
                         *(const float4*)(b + (long long)w * 8 + 8)};
    const float4 g = *(const float4*)(dy + ((ncb * oh + oy) * ow + ox) * 8 + half * 4);
    const int me = (yy & 1) * 2 + (xx & 1);
    const float* vf = (const float*)v;
    const float gs[4] = {g.x, g.y, g.z, g.w};
    float os[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int arg = 0;
      float m = vf[e];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (vf[k * 4 + e] > m) {
          m = vf[k * 4 + e];
          arg = k;
        }
      os[e] = arg == me ? gs[e] : 0.f;
    }
    o = make_float4(os[0], os[1], os[2], os[3]);
  }
  *(float4*)(dx + ((ncb * h + yy) * w + xx) * 8 + half * 4) = o;
}
// y[n][c][:] = x * a[c] + b[c] on NCHW (input normalisation, vgg_arch.py:156-159; its backward is the same with b = null)
__global__ void channel_affine_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ a,
                                      const float* __restrict__ b, int c, long long hw, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ch = (int)((i / hw) % c);
  y[i] = x[i] * a[ch] + (b ? b[ch] : 0.f);
}
__global__ void lrelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float slope, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] > 0.f ? x[i] : x[i] * slope;
}

inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// ===================================================================== C ABI
extern "C" size_t sr_reduce_workspace_bytes(int channels) {
  const size_t cb = (size_t)(channels + 7) / 8;
  return (cb * RED_SPLITS * 16 + 4 * RED_SPLITS + 64) * sizeof(float);
}

extern "C" int sr_bn_lrelu_fwd_f32(const float* x, int64_t x_ns, float* y, int64_t y_ns, int n, int c, int h, int w,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int train, float momentum, float eps, float slope, float* save_mean,
                                   float* save_invstd, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && y && gamma && beta && save_mean && save_invstd && ws && n > 0 && c > 0 && h > 0 && w > 0,
               "sr_bn_lrelu_fwd_f32: bad argument");
  SR_CHECK_ARG(train || (running_mean && running_var), "sr_bn_lrelu_fwd_f32: eval mode needs running statistics");
  SR_CHECK_ARG(ws_bytes >= sr_reduce_workspace_bytes(c), "sr_bn_lrelu_fwd_f32: workspace too small");
  const int cblocks = (c + 7) / 8, hw = h * w;
  const long long count = (long long)n * hw;
  if (train && count <= bnsmall::kBnSmallPixels && sr::bn_small_enabled()) {  // one launch for the whole pass (bn_small.h)
    bnsmall::Params<float> q = {};
    q.x = x;
    q.out = y;
    q.x_ns = x_ns;
    q.out_ns = y_ns;
    q.n = n;
    q.c = c;
    q.hw = hw;
    q.gamma = gamma;
    q.beta = beta;
    q.mean = save_mean;
    q.invstd = save_invstd;
    q.running_mean = running_mean;
    q.running_var = running_var;
    q.momentum = momentum;
    q.eps = eps;
    q.slope = slope;
    bnsmall::launch<float, 8, false>(q, stream);
    SR_CHECK_LAUNCH("bn_small fwd");
    return SR_OK;
  }
  float* part = (float*)ws;
  if (train) {
    BnRedParams p = {};
    p.x = x;
    p.x_ns = x_ns;
    p.n = n;
    p.hw = hw;
    p.part = part;
    p.mode = 0;
    hipLaunchKernelGGL(bn_reduce_kernel, dim3(cblocks, RED_SPLITS), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(nblk(c)), dim3(256), 0, stream, part, c, 0, count, eps, momentum, save_mean,
                       save_invstd, nullptr, nullptr, nullptr, nullptr, nullptr);
    p.mode = 1;
    p.mean = save_mean;
    hipLaunchKernelGGL(bn_reduce_kernel, dim3(cblocks, RED_SPLITS), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(nblk(c)), dim3(256), 0, stream, part, c, 1, count, eps, momentum, save_mean,
                       save_invstd, running_mean, running_var, nullptr, nullptr, nullptr);
  } else {
    hipLaunchKernelGGL(bn_eval_prep_kernel, dim3(nblk(c)), dim3(256), 0, stream, running_mean, running_var, eps, c,
                       save_mean, save_invstd);
  }
  const long long total = (long long)n * cblocks * hw * 2;
  hipLaunchKernelGGL(bn_lrelu_fwd_kernel, dim3(nblk(total)), dim3(256), 0, stream, x, (long long)x_ns, y, (long long)y_ns,
                     save_mean, save_invstd, gamma, beta, slope, c, cblocks, hw, total);
  SR_CHECK_LAUNCH("bn_lrelu_fwd");
  return SR_OK;
}

extern "C" int sr_bn_lrelu_bwd_f32(const float* x, int64_t x_ns, const float* dy, int64_t dy_ns, const float* y,
                                   int64_t y_ns, float* dx, int64_t dx_ns, int n, int c, int h, int w, const float* gamma,
                                   const float* save_mean, const float* save_invstd, int train, float slope,
                                   float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && dy && y && dx && gamma && save_mean && save_invstd && dgamma && dbeta && ws && n > 0 && c > 0,
               "sr_bn_lrelu_bwd_f32: bad argument");
  SR_CHECK_ARG(ws_bytes >= sr_reduce_workspace_bytes(c), "sr_bn_lrelu_bwd_f32: workspace too small");
  const int cblocks = (c + 7) / 8, hw = h * w;
  const long long count = (long long)n * hw;
  if (train && count <= bnsmall::kBnSmallPixels && sr::bn_small_enabled()) {
    bnsmall::Params<float> q = {};
    q.x = x;
    q.dy = dy;
    q.y = y;
    q.out = dx;
    q.x_ns = x_ns;
    q.dy_ns = dy_ns;
    q.y_ns = y_ns;
    q.out_ns = dx_ns;
    q.n = n;
    q.c = c;
    q.hw = hw;
    q.gamma = gamma;
    q.mean = (float*)save_mean;
    q.invstd = (float*)save_invstd;
    q.dgamma = dgamma;
    q.dbeta = dbeta;
    q.slope = slope;
    bnsmall::launch<float, 8, true>(q, stream);
    SR_CHECK_LAUNCH("bn_small bwd");
    return SR_OK;
  }
  float* part = (float*)ws;
  BnRedParams p = {};
  p.x = x;
  p.x_ns = x_ns;
  p.dy = dy;
  p.dy_ns = dy_ns;
  p.y = y;
  p.y_ns = y_ns;
  p.mean = save_mean;
  p.invstd = save_invstd;
  p.n = n;
  p.hw = hw;
  p.part = part;
  p.mode = 2;
  p.slope = slope;
  hipLaunchKernelGGL(bn_reduce_kernel, dim3(cblocks, RED_SPLITS), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(nblk(c)), dim3(256), 0, stream, part, c, 2, count, 0.f, 0.f, nullptr,
                     nullptr, nullptr, nullptr, dgamma, dbeta, nullptr);
  const long long total = (long long)n * cblocks * hw * 2;
  hipLaunchKernelGGL(bn_lrelu_bwd_kernel, dim3(nblk(total)), dim3(256), 0, stream, x, (long long)x_ns, dy, (long long)dy_ns,
                     y, (long long)y_ns, dx, (long long)dx_ns, save_mean, save_invstd, gamma, dgamma, dbeta, slope, train,
                     1.f / (float)count, c, cblocks, hw, total);
  SR_CHECK_LAUNCH("bn_lrelu_bwd");
  return SR_OK;
}

extern "C" int sr_lrelu_bwd_f32(const float* dy, const float* y, float* dz, float slope, int64_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(dy && y && dz && n > 0, "sr_lrelu_bwd_f32: bad argument");
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, dy, y, dz, slope, (long long)n);
  SR_CHECK_LAUNCH("lrelu_bwd");
  return SR_OK;
}

extern "C" int sr_linear_fwd_f32(const float* x, const float* w, const float* b, float* y, int n, int in, int out,
                                 float act_slope, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && w && y && n > 0 && in > 0 && out > 0, "sr_linear_fwd_f32: bad argument");
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(out, n), dim3(256), 0, stream, x, w, b, y, in, out, act_slope);
  SR_CHECK_LAUNCH("linear_fwd");
  return SR_OK;
}

extern "C" int sr_linear_bwd_f32(const float* x, const float* w, const float* y, const float* dy, int n, int in, int out,
                                 float act_slope, float* dz, float* dx, float* dw, float* db, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && w && y && dy && dz && n > 0 && in > 0 && out > 0, "sr_linear_bwd_f32: bad argument");
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(nblk((long long)n * out)), dim3(256), 0, stream, dy, y, dz, act_slope,
                     (long long)n * out);
  if (dx) hipLaunchKernelGGL(linear_bwd_x_kernel, dim3(nblk(in), n), dim3(256), 0, stream, dz, w, dx, in, out);
  if (dw) hipLaunchKernelGGL(linear_bwd_w_kernel, dim3(nblk(in), out), dim3(256), 0, stream, dz, x, dw, db, in, out, n);
  SR_CHECK_LAUNCH("linear_bwd");
  return SR_OK;
}

static int flat_reduce(const float* x, const float* t, const float* shift, long long n, int mode, float sign, float scale,
                       float* out, float* ws, hipStream_t stream) {
  FlatRedParams p = {};
  p.x = x;
  p.t = t;
  p.shift = shift;
  p.part = ws;
  p.n = n;
  p.mode = mode;
  p.sign = sign;
  int blocks = (int)((n + 4095) / 4096);
  if (blocks > 4 * RED_SPLITS) blocks = 4 * RED_SPLITS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(flat_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(flat_finalize_kernel, dim3(1), dim3(64), 0, stream, ws, blocks, scale, out);
  SR_CHECK_LAUNCH("flat_reduce");
  return SR_OK;
}

extern "C" int sr_mean_f32(const float* x, int64_t n, float* out, void* ws, size_t ws_bytes, void* stream) {
  SR_CHECK_ARG(x && out && ws && n > 0 && ws_bytes >= sr_reduce_workspace_bytes(8), "sr_mean_f32: bad argument");
  return flat_reduce(x, nullptr, nullptr, n, 0, 1.f, 1.f / (float)n, out, (float*)ws, (hipStream_t)stream);
}

extern "C" int sr_l1_loss_fwd_f32(const float* pred, const float* target, int64_t n, float weight, float* loss, void* ws,
                                  size_t ws_bytes, void* stream) {
  SR_CHECK_ARG(pred && target && loss && ws && n > 0 && ws_bytes >= sr_reduce_workspace_bytes(8),
               "sr_l1_loss_fwd_f32: bad argument");
  return flat_reduce(pred, target, nullptr, n, 1, 1.f, weight / (float)n, loss, (float*)ws, (hipStream_t)stream);
}

extern "C" int sr_l1_loss_bwd_f32(const float* pred, const float* target, int64_t n, float weight, const float* gout,
                                  float* dpred, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(pred && target && gout && dpred && n > 0, "sr_l1_loss_bwd_f32: bad argument");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, pred, target, gout, weight / (float)n, dpred,
                     (long long)n);
  SR_CHECK_LAUNCH("l1_bwd");
  return SR_OK;
}

// kind 0: L1, 1: MSE, 2: Charbonnier(eps) — weight * mean(loss(pred - target)) and its gradient
extern "C" int sr_pixel_loss_fwd_f32(const float* pred, const float* target, int64_t n, int kind, float eps, float weight,
                                     float* loss, void* ws, size_t ws_bytes, void* stream) {
  SR_CHECK_ARG(pred && target && loss && ws && n > 0 && ws_bytes >= sr_reduce_workspace_bytes(8) && kind >= 0 && kind <= 2,
               "sr_pixel_loss_fwd_f32: bad argument");
  return flat_reduce(pred, target, nullptr, n, kind == 0 ? 1 : kind == 1 ? 5 : 6, kind == 2 ? eps : 1.f, weight / (float)n, loss,
                     (float*)ws, (hipStream_t)stream);
}

extern "C" int sr_pixel_loss_bwd_f32(const float* pred, const float* target, int64_t n, int kind, float eps, float weight,
                                     const float* gout, float* dpred, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(pred && target && gout && dpred && n > 0 && kind >= 0 && kind <= 2, "sr_pixel_loss_bwd_f32: bad argument");
  if (kind == 0) return sr_l1_loss_bwd_f32(pred, target, n, weight, gout, dpred, stream_);
  hipLaunchKernelGGL(pixel_bwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, pred, target, gout, weight / (float)n, kind, eps, dpred,
                     (long long)n);
  SR_CHECK_LAUNCH("pixel_bwd");
  return SR_OK;
}

// Point-wise GAN criteria of GANLoss (losses.py:379-461) other than the fused relativistic BCE: weight * mean f(x) with
// kind 1: (x - c)^2 (lsgan, c = label); 2: c * x (wgan, c = -1 real / +1 fake; hinge generator c = -1);
// 3: softplus(c x) (wgan_softplus, c = -1 real / +1 fake); 4: relu(1 + c x) (hinge discriminator, c = -1 real / +1 fake);
// 5: softplus(x) - c x (vanilla against a soft label c)
extern "C" int sr_gan_point_loss_fwd_f32(const float* x, int64_t n, int kind, float c, float weight, float* loss, void* ws,
                                         size_t ws_bytes, void* stream) {
  SR_CHECK_ARG(x && loss && ws && n > 0 && kind >= 1 && kind <= 5 && ws_bytes >= sr_reduce_workspace_bytes(8),
               "sr_gan_point_loss_fwd_f32: bad argument");
  if (kind == 2) return flat_reduce(x, nullptr, nullptr, n, 0, 1.f, c * weight / (float)n, loss, (float*)ws, (hipStream_t)stream);
  const int mode = kind == 1 ? 7 : kind == 3 ? 2 : kind == 4 ? 8 : 9;
  return flat_reduce(x, nullptr, nullptr, n, mode, c, weight / (float)n, loss, (float*)ws, (hipStream_t)stream);
}

extern "C" int sr_gan_point_loss_bwd_f32(const float* x, int64_t n, int kind, float c, float weight, const float* gout, float* dx,
                                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && gout && dx && n > 0 && kind >= 1 && kind <= 5, "sr_gan_point_loss_bwd_f32: bad argument");
  hipLaunchKernelGGL(gan_point_bwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, x, gout, kind, c, weight / (float)n, dx,
                     (long long)n);
  SR_CHECK_LAUNCH("gan_point_bwd");
  return SR_OK;
}

extern "C" int sr_bce_logits_fwd_f32(const float* x, const float* shift, int64_t n, int target_is_real, float weight,
                                     float* loss, float* dsum, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && loss && ws && n > 0 && ws_bytes >= sr_reduce_workspace_bytes(8), "sr_bce_logits_fwd_f32: bad argument");
  const float sign = target_is_real ? -1.f : 1.f;
  int rc = flat_reduce(x, nullptr, shift, n, 2, sign, weight / (float)n, loss, (float*)ws, stream);
  if (rc) return rc;
  if (dsum) rc = flat_reduce(x, nullptr, shift, n, 3, sign, weight / (float)n, dsum, (float*)ws, stream);
  return rc;
}

extern "C" int sr_bce_logits_bwd_f32(const float* x, const float* shift, int64_t n, int target_is_real, float weight,
                                     const float* gout, float* dx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && gout && dx && n > 0, "sr_bce_logits_bwd_f32: bad argument");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, x, shift, gout, target_is_real ? -1.f : 1.f,
                     weight / (float)n, dx, (long long)n);
  SR_CHECK_LAUNCH("bce_bwd");
  return SR_OK;
}

extern "C" int sr_fill_scaled_f32(const float* gout, const float* s, float scale, float* dx, int64_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(gout && s && dx && n > 0, "sr_fill_scaled_f32: bad argument");
  hipLaunchKernelGGL(fill_scaled_kernel, dim3(nblk(n)), dim3(256), 0, stream, gout, s, scale, dx, (long long)n);
  SR_CHECK_LAUNCH("fill_scaled");
  return SR_OK;
}

extern "C" int sr_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                                float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                const int32_t* abort_word, int32_t* skipped, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "sr_adam_step_f32: bad argument");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(nblk(n)), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, (long long)n, lr,
                     beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale, abort_word, skipped);
  SR_CHECK_LAUNCH("adam");
  return SR_OK;
}

extern "C" int sr_axpby_f32(float* dst, const float* src, float a, float b, int64_t n, const int32_t* abort_word, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(dst && src && n > 0, "sr_axpby_f32: bad argument");
  hipLaunchKernelGGL(axpby_kernel, dim3(nblk(n)), dim3(256), 0, stream, dst, src, a, b, (long long)n, abort_word);
  SR_CHECK_LAUNCH("axpby");
  return SR_OK;
}


extern "C" int sr_bilinear2x_fwd_f32(const float* src, int64_t src_ns, float* dst, int64_t dst_ns, int n, int cblocks, int h,
                                     int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(src && dst && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_bilinear2x_fwd_f32: bad argument");
  const long long total = (long long)n * cblocks * h * w * 8;
  hipLaunchKernelGGL(bilinear2x_fwd_kernel, dim3(nblk(total)), dim3(256), 0, stream, src, (long long)src_ns, dst,
                     (long long)dst_ns, cblocks, h, w, total);
  SR_CHECK_LAUNCH("bilinear2x_fwd");
  return SR_OK;
}

extern "C" int sr_bilinear2x_bwd_f32(const float* g, int64_t g_ns, float* gsrc, int64_t gsrc_ns, int n, int cblocks, int h,
                                     int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(g && gsrc && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_bilinear2x_bwd_f32: bad argument");
  const long long total = (long long)n * cblocks * h * w * 2;
  hipLaunchKernelGGL(bilinear2x_bwd_kernel, dim3(nblk(total)), dim3(256), 0, stream, g, (long long)g_ns, gsrc,
                     (long long)gsrc_ns, cblocks, h, w, total);
  SR_CHECK_LAUNCH("bilinear2x_bwd");
  return SR_OK;
}

extern "C" int sr_spectral_norm_fwd_f32(const float* w_orig, float* u, float* v, int rows, int cols, int update, float eps,
                                        float* w_sn, float* sigma, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(w_orig && u && v && w_sn && sigma && ws && rows > 0 && cols > 0, "sr_spectral_norm_fwd_f32: bad argument");
  SR_CHECK_ARG(ws_bytes >= ((size_t)rows + (size_t)16 * cols) * sizeof(float), "sr_spectral_norm_fwd_f32: workspace too small");
  float* t = (float*)ws;           // [<= 16 row chunks][cols]
  float* s = t + (size_t)16 * cols;  // [rows]
  if (update) {  // one power iteration: v = normalize(W^T u); u = normalize(W v)  (in place, like the reference module)
    const int parts = rows >= 512 ? 16 : rows >= 64 ? 8 : 1, chunk = (rows + parts - 1) / parts;
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3(nblk(cols), parts), dim3(256), 0, stream, w_orig, u, t, rows, cols, chunk);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(256), 0, stream, t, v, cols, eps, (float*)nullptr, parts);
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(rows), dim3(256), 0, stream, w_orig, v, s, cols);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(256), 0, stream, s, u, rows, eps, sigma, 1);  // sigma = u.(Wv)
  } else {
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(rows), dim3(256), 0, stream, w_orig, v, s, cols);
    hipLaunchKernelGGL(sn_dot_kernel, dim3(1), dim3(256), 0, stream, u, s, rows, sigma);
  }
  const long long n = (long long)rows * cols;
  hipLaunchKernelGGL(sn_scale_kernel, dim3(nblk(n)), dim3(256), 0, stream, w_orig, sigma, w_sn, n);
  SR_CHECK_LAUNCH("spectral_norm_fwd");
  return SR_OK;
}

// ---- the forward for all spectral-norm layers of a network at once: one launch per stage, blockIdx.z = layer ----
namespace {
struct SnOne {
  const float* W;
  float* u;
  float* v;
  float* w_sn;
  float* sigma;
  float* t;  // [parts][cols]
  float* s;  // [rows]
  int rows, cols, parts, chunk;
};
struct SnBatch {
  SnOne l[SR_SN_BATCH_MAX];
};
// (the table is indexed by blockIdx.z: it lives in scratch for these launches, which are a few microseconds of latency-bound work)
__global__ void sn_wt_u_batch_kernel(const SnBatch b) {
  const SnOne& L = b.l[blockIdx.z];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= L.cols || (int)blockIdx.y >= L.parts) return;
  const int r0 = blockIdx.y * L.chunk, r1 = min(r0 + L.chunk, L.rows);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += L.W[(long long)r * L.cols + c] * L.u[r];
  L.t[(long long)blockIdx.y * L.cols + c] = s;
}
__global__ __launch_bounds__(256) void sn_w_v_batch_kernel(const SnBatch b) {
  __shared__ float sh[4];
  const SnOne& L = b.l[blockIdx.z];
  const int r = blockIdx.x;
  if (r >= L.rows) return;
  float a = 0.f;
  for (int c = threadIdx.x; c < L.cols; c += 256) a += L.W[(long long)r * L.cols + c] * L.v[c];
  a = block_sum(a, sh);
  if (threadIdx.x == 0) L.s[r] = a;
}
// which 0: v = normalize(fold(t)); 1: u = normalize(s), sigma = u . s; 2 (eval): sigma = u . s
__global__ __launch_bounds__(256) void sn_vec_batch_kernel(const SnBatch b, const int which, const float eps) {
  __shared__ float sh[4];
  const SnOne& L = b.l[blockIdx.z];
  if (which == 2) {
    float s = 0.f;
    for (int i = threadIdx.x; i < L.rows; i += 256) s += L.u[i] * L.s[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) L.sigma[0] = s;
    return;
  }
  float* x = which == 0 ? L.t : L.s;
  float* out = which == 0 ? L.v : L.u;
  const int n = which == 0 ? L.cols : L.rows, parts = which == 0 ? L.parts : 1;
  if (parts > 1) {
    for (int i = threadIdx.x; i < n; i += 256) {
      float s = 0.f;
      for (int y = 0; y < parts; ++y) s += x[(long long)y * n + i];
      x[i] = s;
    }
    __syncthreads();
  }
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += x[i] * x[i];
  a = block_sum(a, sh);
  const float nrm = fmaxf(sqrtf(a), eps);
  for (int i = threadIdx.x; i < n; i += 256) out[i] = x[i] / nrm;
  if (which == 1 && threadIdx.x == 0) L.sigma[0] = a / nrm;
}
__global__ void sn_scale_batch_kernel(const SnBatch b) {
  const SnOne& L = b.l[blockIdx.z];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long long)L.rows * L.cols) L.w_sn[i] = L.W[i] / L.sigma[0];
}
}  // namespace

extern "C" int sr_spectral_norm_fwd_batch_f32(const sr_sn_layer* layers, int n_layers, int update, float eps, void* ws, size_t ws_bytes,
                                              void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(layers && n_layers > 0 && n_layers <= SR_SN_BATCH_MAX && ws, "sr_spectral_norm_fwd_batch_f32: 1..%d layers", SR_SN_BATCH_MAX);
  SnBatch b = {};
  size_t need = 0;
  int max_cols = 0, max_rows = 0, max_parts = 0;
  long long max_n = 0;
  for (int i = 0; i < n_layers; ++i) {
    const sr_sn_layer& a = layers[i];
    SR_CHECK_ARG(a.w_orig && a.u && a.v && a.w_sn && a.sigma && a.rows > 0 && a.cols > 0, "sr_spectral_norm_fwd_batch_f32: bad layer %d", i);
    SnOne& L = b.l[i];
    L.W = a.w_orig;
    L.u = a.u;
    L.v = a.v;
    L.w_sn = a.w_sn;
    L.sigma = a.sigma;
    L.rows = a.rows;
    L.cols = a.cols;
    L.parts = a.rows >= 512 ? 16 : a.rows >= 64 ? 8 : 1;  // as sr_spectral_norm_fwd_f32
    L.chunk = (a.rows + L.parts - 1) / L.parts;
    L.t = (float*)((char*)ws + need);
    L.s = L.t + (size_t)16 * a.cols;
    need += ((size_t)a.rows + (size_t)16 * a.cols) * sizeof(float);
    need = (need + 255) / 256 * 256;
    max_cols = a.cols > max_cols ? a.cols : max_cols;
    max_rows = a.rows > max_rows ? a.rows : max_rows;
    max_parts = L.parts > max_parts ? L.parts : max_parts;
    max_n = (long long)a.rows * a.cols > max_n ? (long long)a.rows * a.cols : max_n;
  }
  SR_CHECK_ARG(ws_bytes >= need, "sr_spectral_norm_fwd_batch_f32: workspace %zu B < %zu B", ws_bytes, need);
  for (int i = n_layers; i < SR_SN_BATCH_MAX; ++i) b.l[i] = b.l[0];
  if (update) {
    hipLaunchKernelGGL(sn_wt_u_batch_kernel, dim3(nblk(max_cols), max_parts, n_layers), dim3(256), 0, stream, b);
    hipLaunchKernelGGL(sn_vec_batch_kernel, dim3(1, 1, n_layers), dim3(256), 0, stream, b, 0, eps);
    hipLaunchKernelGGL(sn_w_v_batch_kernel, dim3(max_rows, 1, n_layers), dim3(256), 0, stream, b);
    hipLaunchKernelGGL(sn_vec_batch_kernel, dim3(1, 1, n_layers), dim3(256), 0, stream, b, 1, eps);
  } else {
    hipLaunchKernelGGL(sn_w_v_batch_kernel, dim3(max_rows, 1, n_layers), dim3(256), 0, stream, b);
    hipLaunchKernelGGL(sn_vec_batch_kernel, dim3(1, 1, n_layers), dim3(256), 0, stream, b, 2, eps);
  }
  hipLaunchKernelGGL(sn_scale_batch_kernel, dim3(nblk(max_n), 1, n_layers), dim3(256), 0, stream, b);
  SR_CHECK_LAUNCH("spectral_norm_fwd_batch");
  return SR_OK;
}

extern "C" int sr_spectral_norm_bwd_f32(const float* g_wsn, const float* w_sn, const float* u, const float* v,
                                        const float* sigma, int rows, int cols, float* g_worig, void* ws, size_t ws_bytes,
                                        void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(g_wsn && w_sn && u && v && sigma && g_worig && ws && rows > 0 && cols > 0,
               "sr_spectral_norm_bwd_f32: bad argument");
  SR_CHECK_ARG(ws_bytes >= sr_reduce_workspace_bytes(8) + 64, "sr_spectral_norm_bwd_f32: workspace too small");
  const long long n = (long long)rows * cols;
  float* dot = (float*)ws;
  int rc = flat_reduce(g_wsn, w_sn, nullptr, n, 4, 1.f, 1.f, dot, dot + 16, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(sn_bwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, g_wsn, u, v, sigma, dot, g_worig, cols, n);
  SR_CHECK_LAUNCH("spectral_norm_bwd");
  return SR_OK;
}

extern "C" int sr_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(a && b && out && n > 0 && n % 4 == 0, "sr_add_f32: n must be a positive multiple of 4");
  hipLaunchKernelGGL(add_kernel, dim3(nblk(n / 4)), dim3(256), 0, stream, a, b, out, (long long)(n / 4));
  SR_CHECK_LAUNCH("add");
  return SR_OK;
}

extern "C" int sr_psnr_sse_f32(const float* a, const float* b, int n, int c, int h, int w, int crop_border, float* sse,
                               void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(a && b && sse && ws && n > 0 && c > 0 && crop_border >= 0 && h > 2 * crop_border && w > 2 * crop_border,
               "sr_psnr_sse_f32: bad argument");
  const int parts = 64;
  SR_CHECK_ARG(ws_bytes >= (size_t)n * parts * sizeof(float), "sr_psnr_sse_f32: workspace too small");
  hipLaunchKernelGGL(psnr_sse_kernel, dim3(parts, n), dim3(256), 0, stream, a, b, c, h, w, crop_border, (float*)ws);
  hipLaunchKernelGGL(psnr_finalize_kernel, dim3(n), dim3(64), 0, stream, (const float*)ws, parts, sse);
  SR_CHECK_LAUNCH("psnr_sse");
  return SR_OK;
}

extern "C" int sr_ssim_sum_f32(const float* a, const float* b, int n, int c, int h, int w, int crop_border, float* sum,
                               void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(a && b && sum && ws && n > 0 && c > 0 && crop_border >= 0 && h > 2 * crop_border + 10 && w > 2 * crop_border + 10,
               "sr_ssim_sum_f32: bad argument (the cropped image must be larger than the 11x11 window)");
  const int parts = 64;
  SR_CHECK_ARG(ws_bytes >= (size_t)n * parts * sizeof(float), "sr_ssim_sum_f32: workspace too small");
  SsimWin win;  // cv2.getGaussianKernel(11, 1.5): exp(-(i-5)^2 / (2 sigma^2)), normalised
  double g[11], tot = 0;
  for (int i = 0; i < 11; ++i) tot += g[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
  for (int i = 0; i < 11; ++i) win.w[i] = (float)(g[i] / tot);
  hipLaunchKernelGGL(ssim_sum_kernel, dim3(parts, n), dim3(256), 0, stream, a, b, c, h, w, crop_border, win, (float*)ws);
  hipLaunchKernelGGL(psnr_finalize_kernel, dim3(n), dim3(64), 0, stream, (const float*)ws, parts, sum);
  SR_CHECK_LAUNCH("ssim_sum");
  return SR_OK;
}

extern "C" int sr_maxpool2x2_fwd_f32(const float* x, float* y, int n, int cblocks, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && y && n > 0 && cblocks > 0 && h >= 2 && w >= 2, "sr_maxpool2x2_fwd_f32: bad argument");
  const long long total = (long long)n * cblocks * (h / 2) * (w / 2) * 2;
  hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3(nblk(total)), dim3(256), 0, stream, x, y, cblocks, h, w, total);
  SR_CHECK_LAUNCH("maxpool2x2_fwd");
  return SR_OK;
}

extern "C" int sr_maxpool2x2_bwd_f32(const float* x, const float* dy, float* dx, int n, int cblocks, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && dy && dx && n > 0 && cblocks > 0 && h >= 2 && w >= 2, "sr_maxpool2x2_bwd_f32: bad argument");
  const long long total = (long long)n * cblocks * h * w * 2;
  hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(nblk(total)), dim3(256), 0, stream, x, dy, dx, cblocks, h, w, total);
  SR_CHECK_LAUNCH("maxpool2x2_bwd");
  return SR_OK;
}

extern "C" int sr_channel_affine_f32(const float* x, float* y, const float* a, const float* b, int n, int c, int64_t hw,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && y && a && n > 0 && c > 0 && hw > 0, "sr_channel_affine_f32: bad argument");
  const long long total = (long long)n * c * hw;
  hipLaunchKernelGGL(channel_affine_kernel, dim3(nblk(total)), dim3(256), 0, stream, x, y, a, b, c, (long long)hw, total);
  SR_CHECK_LAUNCH("channel_affine");
  return SR_OK;
}

extern "C" int sr_lrelu_fwd_f32(const float* x, float* y, float slope, int64_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && y && n > 0, "sr_lrelu_fwd_f32: bad argument");
  hipLaunchKernelGGL(lrelu_fwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, x, y, slope, (long long)n);
  SR_CHECK_LAUNCH("lrelu_fwd");
  return SR_OK;
}

// ------------------------------------------------------------------------------------------------ Gram matrices (style term)
// PerceptualLoss._gram_mat (basicsr/losses/losses.py:342-356): G[n] = F[n] F[n]^T * scale, F = the NCHW feature map as [c, h*w].
// Small matrices (c <= 512) over long rows (h*w up to 128^2): a 32x32 tile of G per workgroup, the pixel axis in chunks of 32
// through the LDS (rows read along the contiguous pixel axis), 2x2 outputs per thread, fp32 FMA in pixel order — the result does not
// depend on the launch geometry.  Backward: dF = (dG + dG^T) F * scale, a 32 (channels) x 64 (pixels) tile per workgroup.
namespace {
__global__ __launch_bounds__(256) void gram_fwd_kernel(const float* __restrict__ f, float* __restrict__ g, int c, long long hw, float scale) {
  __shared__ float a[32][33], b[32][33];
  const int n = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const float* fn = f + (size_t)n * c * hw;
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  const int lr = tid >> 3, lc = (tid & 7) * 4;  // loader: row of the tile, first of 4 pixels
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  for (long long p0 = 0; p0 < hw; p0 += 32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long p = p0 + lc + e;
      a[lr][lc + e] = (i0 + lr < c && p < hw) ? fn[(size_t)(i0 + lr) * hw + p] : 0.f;
      b[lr][lc + e] = (j0 + lr < c && p < hw) ? fn[(size_t)(j0 + lr) * hw + p] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const float a0 = a[2 * ti][k], a1 = a[2 * ti + 1][k], b0 = b[2 * tj][k], b1 = b[2 * tj + 1][k];
      acc[0][0] += a0 * b0;
      acc[0][1] += a0 * b1;
      acc[1][0] += a1 * b0;
      acc[1][1] += a1 * b1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int i = i0 + 2 * ti + u, j = j0 + 2 * tj + v;
      if (i < c && j < c) g[((size_t)n * c + i) * c + j] = acc[u][v] * scale;
    }
}

__global__ __launch_bounds__(256) void gram_bwd_kernel(const float* __restrict__ f, const float* __restrict__ dg, float* __restrict__ df, int c,
                                                       long long hw, float scale) {
  __shared__ float s[32][33], x[32][65];
  const int n = blockIdx.z, i0 = blockIdx.y * 32;
  const long long p0 = (long long)blockIdx.x * 64;
  const float* fn = f + (size_t)n * c * hw;
  const float* dgn = dg + (size_t)n * c * c;
  const int tid = threadIdx.x, ti = tid >> 4, tp = tid & 15;  // outputs: channels 2 ti, 2 ti + 1; pixels tp + 16 q, q = 0..3
  float acc[2][4] = {};
  for (int j0 = 0; j0 < c; j0 += 32) {
    {
      const int r = tid >> 3, cc = (tid & 7) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = i0 + r, j = j0 + cc + e;
        s[r][cc + e] = (i < c && j < c) ? dgn[(size_t)i * c + j] + dgn[(size_t)j * c + i] : 0.f;
      }
      const int xr = tid >> 3, xc = (tid & 7) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const long long p = p0 + xc + e;
        x[xr][xc + e] = (j0 + xr < c && p < hw) ? fn[(size_t)(j0 + xr) * hw + p] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const float s0 = s[2 * ti][k], s1 = s[2 * ti + 1][k];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xv = x[k][tp + 16 * q];
        acc[0][q] += s0 * xv;
        acc[1][q] += s1 * xv;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + 2 * ti + u;
      const long long p = p0 + tp + 16 * q;
      if (i < c && p < hw) df[((size_t)n * c + i) * hw + p] = acc[u][q] * scale;
    }
}
}  // namespace

extern "C" int sr_gram_fwd_f32(const float* feat, int n, int c, int64_t hw, float scale, float* gram, void* stream) {
  SR_CHECK_ARG(feat && gram && n > 0 && c > 0 && hw > 0 && n < 65536, "sr_gram_fwd_f32: bad argument");
  hipLaunchKernelGGL(gram_fwd_kernel, dim3((c + 31) / 32, (c + 31) / 32, n), dim3(256), 0, (hipStream_t)stream, feat, gram, c, (long long)hw, scale);
  SR_CHECK_LAUNCH("gram_fwd");
  return SR_OK;
}

extern "C" int sr_gram_bwd_f32(const float* feat, const float* dgram, int n, int c, int64_t hw, float scale, float* dfeat, void* stream) {
  SR_CHECK_ARG(feat && dgram && dfeat && n > 0 && c > 0 && hw > 0 && n < 65536 && (hw + 63) / 64 < (1ll << 31), "sr_gram_bwd_f32: bad argument");
  hipLaunchKernelGGL(gram_bwd_kernel, dim3((unsigned)((hw + 63) / 64), (c + 31) / 32, n), dim3(256), 0, (hipStream_t)stream, feat, dgram, dfeat, c,
                     (long long)hw, scale);
  SR_CHECK_LAUNCH("gram_bwd");
  return SR_OK;
}
