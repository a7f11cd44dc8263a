// Discriminator-side helpers of the bf16 path (UNetDiscriminatorSN with compute_dtype='bf16'; the reference contains no
// reduced precision and no U-Net discriminator — SURVEY.md §0 D2/D5 — so everything here is a build extension whose
// parity is declared against this library's fp32 path, tests/test_unet_disc_bf16_gpu.py).
//
// The 3x3 convolutions, their data and weight gradients are the generator's bf16 kernels (conv_bf16.hip,
// wgrad_bf16.hip).  The 4x4 / stride-2 / pad-1 convolutions reuse them too: with X' = pixel_unshuffle(X, 2) in
// parity-major channel order c' = (2 ry + rx) C + c,
//     Y[oy][ox] = sum_{ky,kx} W[ky][kx] X[2 oy + ky - 1][2 ox + kx - 1]
// is a 3x3 / stride-1 / pad-1 convolution of X' with W'[co][(2 ry + rx) C + c][ty][tx] = W[co][c][ky][kx] where
// ky -> (ty, ry) = 0 -> (0,1), 1 -> (1,0), 2 -> (1,1), 3 -> (2,0) (same for kx) and all other entries zero: 16 of the
// 36 taps are live, a 2.25x MAC overhead that is free at bf16 rates (these layers are memory-bound), against four
// accumulating parity passes with bf16 rounding in between.
#include "sr_internal.h"
#include "bn_small.h"

namespace {
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

// CB16 [N][C/16][2h][2w][16] -> CB16 [N][4C/16][h][w][16], channel c' = (2 ry + rx) C + c (C % 16 == 0);
// one thread per destination half pixel-block.  inverse = 1 runs the adjoint (pixel_shuffle of that channel order).
__global__ void unshuffle2_kernel(const __bf16* __restrict__ src, long long src_ns, __bf16* __restrict__ dst, long long dst_ns,
                                  int cblocks, int h, int w, int inverse, long long total) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int x = (int)(r % w);
  r /= w;
  const int y = (int)(r % h);
  r /= h;
  const int cb4 = (int)(r % (4 * cblocks)), n = (int)(r / (4 * cblocks));  // block of the 4C-channel tensor
  const int par = cb4 / cblocks, cb = cb4 - par * cblocks;
  const int ry = par >> 1, rx = par & 1;
  const long long big = n * (inverse ? dst_ns : src_ns) + (((long long)cb * 2 * h + 2 * y + ry) * (2 * w) + 2 * x + rx) * 16 + half * 8;
  const long long small = n * (inverse ? src_ns : dst_ns) + (((long long)cb4 * h + y) * w + x) * 16 + half * 8;
  if (inverse)
    *(bf16x8_t*)(dst + big) = *(const bf16x8_t*)(src + small);
  else
    *(bf16x8_t*)(dst + small) = *(const bf16x8_t*)(src + big);
}

// Forward pixel unshuffle with whole-line reads: a thread owns one pixel PAIR (2x, 2x+1) of one big row (8 channels) and
// stores it to the two column-parity planes, so a wave reads contiguous memory instead of every other 32-byte pixel.
__global__ void unshuffle2_pairs_kernel(const __bf16* __restrict__ src, long long src_ns, __bf16* __restrict__ dst, long long dst_ns,
                                        int cblocks, int h, int w, long long total) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int x = (int)(r % w);
  r /= w;
  const int Y = (int)(r % (2 * h));
  r /= 2 * h;
  const int cb = (int)(r % cblocks), n = (int)(r / cblocks);
  const int y = Y >> 1, ry = Y & 1;
  const __bf16* b = src + n * src_ns + (((long long)cb * 2 * h + Y) * (2 * w) + 2 * x) * 16 + half * 8;
  const bf16x8_t p0 = *(const bf16x8_t*)b, p1 = *(const bf16x8_t*)(b + 16);
  __bf16* o = dst + n * dst_ns + half * 8;
  *(bf16x8_t*)(o + (((long long)((2 * ry + 0) * cblocks + cb) * h + y) * w + x) * 16) = p0;
  *(bf16x8_t*)(o + (((long long)((2 * ry + 1) * cblocks + cb) * h + y) * w + x) * 16) = p1;
}

// out = bf16(a + b), 8 elements per thread (skip connection in one pass)
__global__ void add16_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b, __bf16* __restrict__ out, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const bf16x8_t x = ((const bf16x8_t*)a)[i], y = ((const bf16x8_t*)b)[i];
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)x[e] + (float)y[e]);
  ((bf16x8_t*)out)[i] = o;
}

// Gradient of an encoder activation x that feeds a skip connection and, pixel-unshuffled, the next 4x4/s2 convolution:
//   dz = lrelu'(x) * (g_skip + unshuffle^-1(g_u))   in one pass (g_skip may be null; mask may be null = no LeakyReLU);
// the sum is rounded to bf16 before the mask, like the separate add and LeakyReLU-backward passes it replaces.
// MASK_U2: the activation only exists pixel-unshuffled (sr_conv3x3_desc.out_unshuffle2): the mask is read at g_u's index.
template <bool MASK_U2>
__global__ void fork_bwd16_kernel(const __bf16* __restrict__ g_skip, const __bf16* __restrict__ g_u, const __bf16* __restrict__ mask,
                                  __bf16* __restrict__ dz, float slope, int cblocks, int h, int w, long long total) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int X = (int)(r % (2 * w));
  r /= 2 * w;
  const int Y = (int)(r % (2 * h));
  r /= 2 * h;
  const int cb = (int)(r % cblocks);
  const long long n = r / cblocks;
  const long long big = ((n * cblocks + cb) * 2 * h + Y) * (2LL * w) * 16 + (long long)X * 16 + half * 8;
  const int par = (Y & 1) * 2 + (X & 1);
  const long long small = (((n * 4 * cblocks + (long long)par * cblocks + cb) * h + (Y >> 1)) * w + (X >> 1)) * 16 + half * 8;
  bf16x8_t s = *(const bf16x8_t*)(g_u + small);
  if (g_skip) {
    const bf16x8_t k = *(const bf16x8_t*)(g_skip + big);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = (__bf16)((float)s[e] + (float)k[e]);
  }
  if (mask) {
    const bf16x8_t m = *(const bf16x8_t*)(mask + (MASK_U2 ? small : big));
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (!((float)m[e] > 0.f)) s[e] = (__bf16)((float)s[e] * slope);
  }
  *(bf16x8_t*)(dz + big) = s;
}

// out = bf16(a + b) where b only exists pixel-unshuffled ([n][4 cblocks][h][w][16] standing for [n][cblocks][2h][2w][16]); thread =
// 8 channels of one pixel of the plain tensors
__global__ void add16_u2_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b_u2, __bf16* __restrict__ out, int cblocks, int h,
                                int w, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int X = (int)(r % (2 * w));
  r /= 2 * w;
  const int Y = (int)(r % (2 * h));
  r /= 2 * h;
  const int cb = (int)(r % cblocks);
  const long long n = r / cblocks;
  const int par = (Y & 1) * 2 + (X & 1);
  const long long small = (((n * 4 * cblocks + (long long)par * cblocks + cb) * h + (Y >> 1)) * w + (X >> 1)) * 16 + half * 8;
  const bf16x8_t x = *(const bf16x8_t*)(a + i * 8), y = *(const bf16x8_t*)(b_u2 + small);
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)x[e] + (float)y[e]);
  *(bf16x8_t*)(out + i * 8) = o;
}

// dz = gy * (xsum - x0 > 0 ? 1 : slope) with x0 pixel-unshuffled: LeakyReLU backward of a conv stored as act + x0 (res1_keep_sign)
__global__ void lrelu_bwd_diff16_u2_kernel(const __bf16* __restrict__ gy, const __bf16* __restrict__ xsum, const __bf16* __restrict__ x0_u2,
                                           __bf16* __restrict__ dz, float slope, int cblocks, int h, int w, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int X = (int)(r % (2 * w));
  r /= 2 * w;
  const int Y = (int)(r % (2 * h));
  r /= 2 * h;
  const int cb = (int)(r % cblocks);
  const long long n = r / cblocks;
  const int par = (Y & 1) * 2 + (X & 1);
  const long long small = (((n * 4 * cblocks + (long long)par * cblocks + cb) * h + (Y >> 1)) * w + (X >> 1)) * 16 + half * 8;
  const bf16x8_t g = *(const bf16x8_t*)(gy + i * 8), s = *(const bf16x8_t*)(xsum + i * 8), x0 = *(const bf16x8_t*)(x0_u2 + small);
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = ((float)s[e] - (float)x0[e] > 0.f) ? g[e] : (__bf16)((float)g[e] * slope);
  *(bf16x8_t*)(dz + i * 8) = o;
}

// y = x > 0 ? x : slope * x, 8 elements per thread (a stand-alone ReLU after a feature that is wanted before it)
__global__ void lrelu_fwd16_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, float slope, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const bf16x8_t a = ((const bf16x8_t*)x)[i];
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (float)a[e] > 0.f ? a[e] : (__bf16)((float)a[e] * slope);
  ((bf16x8_t*)y)[i] = o;
}

// nn.MaxPool2d(2, 2) on CB16 (VGG feature extractor): [N][cb][h][w][16] -> [N][cb][h/2][w/2][16]; thread = 8 channels
__global__ void maxpool16_fwd_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, int h, int w, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int oh = h / 2, ow = w / 2;
  const int ox = (int)(r % ow);
  r /= ow;
  const int oy = (int)(r % oh);
  const long long ncb = r / oh;
  const __bf16* b = x + ((ncb * h + 2 * oy) * w + 2 * ox) * 16 + half * 8;
  const bf16x8_t a00 = *(const bf16x8_t*)b, a01 = *(const bf16x8_t*)(b + 16), a10 = *(const bf16x8_t*)(b + (long long)w * 16),
                 a11 = *(const bf16x8_t*)(b + (long long)w * 16 + 16);
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)fmaxf(fmaxf((float)a00[e], (float)a01[e]), fmaxf((float)a10[e], (float)a11[e]));
  *(bf16x8_t*)(y + ((ncb * oh + oy) * ow + ox) * 16 + half * 8) = o;
}
// the gradient of a window goes to its first maximum in scan order (torch's choice); thread per INPUT pixel half
__global__ void maxpool16_bwd_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy, __bf16* __restrict__ dx, int h,
                                     int w, long long total) {
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
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)0.f;
  if (oy < oh && ox < ow) {
    const __bf16* b = x + ((ncb * h + 2 * oy) * w + 2 * ox) * 16 + half * 8;
    const bf16x8_t v[4] = {*(const bf16x8_t*)b, *(const bf16x8_t*)(b + 16), *(const bf16x8_t*)(b + (long long)w * 16),
                           *(const bf16x8_t*)(b + (long long)w * 16 + 16)};
    const bf16x8_t g = *(const bf16x8_t*)(dy + ((ncb * oh + oy) * ow + ox) * 16 + half * 8);
    const int me = (yy & 1) * 2 + (xx & 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int arg = 0;
      float m = (float)v[0][e];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if ((float)v[k][e] > m) {
          m = (float)v[k][e];
          arg = k;
        }
      if (arg == me) o[e] = g[e];
    }
  }
  *(bf16x8_t*)(dx + ((ncb * h + yy) * w + xx) * 16 + half * 8) = o;
}

// dz = gy * (y > 0 ? 1 : slope), 8 elements per thread
__global__ void lrelu_bwd16_kernel(const __bf16* __restrict__ gy, const __bf16* __restrict__ y, __bf16* __restrict__ dz, float slope,
                                   long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const bf16x8_t g = ((const bf16x8_t*)gy)[i], a = ((const bf16x8_t*)y)[i];
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (float)a[e] > 0.f ? g[e] : (__bf16)((float)g[e] * slope);
  ((bf16x8_t*)dz)[i] = o;
}

// F.interpolate(scale_factor=2, mode='bilinear', align_corners=False), see train_ops.hip
__device__ __forceinline__ void bil_taps16(int o, int size, int& i0, int& i1, float& w0, float& w1) {
  float s = (o + 0.5f) * 0.5f - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < size - 1 ? 1 : 0);
  w1 = s - (float)i0;
  w0 = 1.f - w1;
}
// src2 != null: the interpolated tensor is bf16(src + src2) (a skip connection folded into the resampling pass).
// A thread makes the 2x2 output block of one source pixel from its clamped 3x3 neighbourhood; the arithmetic per output is the
// expression of the one-output form.
constexpr int BIL_TH = 8, BIL_TW = 32;  // source tile of a workgroup (256 threads: one source pixel each, 16 channels in two passes)
// SRC2_U2: src2 only exists pixel-unshuffled ([n][4 cblocks][h/2][w/2][16], sr_conv3x3_desc.out_unshuffle2; h, w even).
template <bool SRC2_U2>
__global__ __launch_bounds__(256) void bilinear2x_fwd16_kernel(const __bf16* __restrict__ src, long long src_ns,
                                                               const __bf16* __restrict__ src2, long long src2_ns,
                                                               __bf16* __restrict__ dst, long long dst_ns, int cblocks, int h, int w,
                                                               int tiles_x, int tiles_y) {
  // The clamped (TH+2) x (TW+2) source window of the tile goes through the LDS once (coalesced 16-byte pieces), so every source
  // pixel is fetched ~1.3 times instead of 9 (the register-window form was bound by those L1/L2 reads); each thread then makes
  // the 2x2 output block of its source pixel, 8 channels at a time.
  __shared__ __attribute__((aligned(16))) __bf16 tile[(BIL_TH + 2) * (BIL_TW + 2) * 16];
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  t /= tiles_y;
  const int cb = t % cblocks, n = t / cblocks;
  const int x0 = tx * BIL_TW, y0 = ty * BIL_TH;
  const long long plane = (long long)cb * h * w * 16;
  const __bf16* b = src + n * src_ns + plane;
  const __bf16* c = src2 ? src2 + n * src2_ns + (SRC2_U2 ? 0 : plane) : nullptr;
  constexpr int PIECES = (BIL_TH + 2) * (BIL_TW + 2) * 2;
  for (int q = threadIdx.x; q < PIECES; q += 256) {
    const int pix = q >> 1, hf = q & 1;
    const int ry = pix / (BIL_TW + 2), rx = pix - ry * (BIL_TW + 2);
    const int yy = min(max(y0 + ry - 1, 0), h - 1), xx = min(max(x0 + rx - 1, 0), w - 1);
    const long long off = ((long long)yy * w + xx) * 16 + hf * 8;
    bf16x8_t v = *(const bf16x8_t*)(b + off);
    if (c) {
      long long off2 = off;
      if constexpr (SRC2_U2)
        off2 = ((((long long)(((yy & 1) << 1) | (xx & 1)) * cblocks + cb) * (h >> 1) + (yy >> 1)) * (w >> 1) + (xx >> 1)) * 16 + hf * 8;
      const bf16x8_t u = *(const bf16x8_t*)(c + off2);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] + (float)u[e]);
    }
    *(bf16x8_t*)(tile + pix * 16 + hf * 8) = v;
  }
  __syncthreads();
  const int lx = threadIdx.x % BIL_TW, ly = threadIdx.x / BIL_TW;
  const int sx = x0 + lx, sy = y0 + ly;
  const int W2 = 2 * w, H2 = 2 * h;
  __bf16* out = dst + n * dst_ns + (long long)cb * H2 * W2 * 16;
  // the 2 TH x 2 TW output tile is assembled in the LDS and leaves in contiguous 16-byte pieces (a thread's own four outputs
  // are 16 bytes every 64: a quarter of every line per store instruction)
  __shared__ __attribute__((aligned(16))) __bf16 otile[2 * BIL_TH * 2 * BIL_TW * 16];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    bf16x8_t win[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) win[dy][dx] = *(const bf16x8_t*)(tile + ((ly + dy) * (BIL_TW + 2) + lx + dx) * 16 + hf * 8);
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      int t0, t1;
      float wy0, wy1;
      bil_taps16(2 * min(sy, h - 1) + py, h, t0, t1, wy0, wy1);
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        float wx0, wx1;
        bil_taps16(2 * min(sx, w - 1) + px, w, t0, t1, wx0, wx1);
        // taps lie in the clamped window: rows sy-1+{py, py+1}, columns sx-1+{px, px+1} (a clamped tap repeats its neighbour)
        const bf16x8_t &a00 = win[py][px], &a01 = win[py][px + 1], &a10 = win[py + 1][px], &a11 = win[py + 1][px + 1];
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = (__bf16)(wy0 * (wx0 * (float)a00[e] + wx1 * (float)a01[e]) + wy1 * (wx0 * (float)a10[e] + wx1 * (float)a11[e]));
        *(bf16x8_t*)(otile + ((2 * ly + py) * (2 * BIL_TW) + 2 * lx + px) * 16 + hf * 8) = o;
      }
    }
  }
  __syncthreads();
  constexpr int OPIECES = 2 * BIL_TH * 2 * BIL_TW * 2;
  for (int q = threadIdx.x; q < OPIECES; q += 256) {
    const int pix = q >> 1, hf = q & 1;
    const int oy = pix / (2 * BIL_TW), ox = pix - oy * (2 * BIL_TW);
    const int gy = 2 * y0 + oy, gx = 2 * x0 + ox;
    if (gy < H2 && gx < W2) *(bf16x8_t*)(out + ((long long)gy * W2 + gx) * 16 + hf * 8) = *(const bf16x8_t*)(otile + pix * 16 + hf * 8);
  }
}
// gsrc[y][x] = sum over the (at most 4x4) outputs whose taps touch (y, x): a gather, deterministic.  The (2 TH + 2) x (2 TW + 2)
// window of output gradients of a source tile goes through the LDS once (coalesced), each thread then gathers the 4x4 block
// of its source pixel from there; weights come from the forward's tap rule, so borders (clamped taps) need no special case.
// MASK: the source was the LeakyReLU(slope) output `mask` of a layer with no other consumer of its pre-activation, so the gradient
// leaves multiplied by that LeakyReLU's derivative (the stand-alone lrelu_bwd16 pass of that layer disappears); `plain` (optional)
// also receives the unmasked gradient — what a skip connection added before the resampling gets.
template <bool MASK>
__global__ __launch_bounds__(256) void bilinear2x_bwd16_kernel(const __bf16* __restrict__ g, long long g_ns, __bf16* __restrict__ gsrc,
                                                               long long gsrc_ns, int cblocks, int h, int w, int tiles_x, int tiles_y,
                                                               const __bf16* __restrict__ mask, long long mask_ns, float slope,
                                                               __bf16* __restrict__ plain, long long plain_ns) {
  constexpr int GW = 2 * BIL_TW + 2, GH = 2 * BIL_TH + 2;
  __shared__ __attribute__((aligned(16))) __bf16 tile[GH * GW * 16];
  int t = blockIdx.x;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  t /= tiles_y;
  const int cb = t % cblocks, n = t / cblocks;
  const int x0 = tx * BIL_TW, y0 = ty * BIL_TH;
  const int W2 = 2 * w, H2 = 2 * h;
  const __bf16* b = g + n * g_ns + (long long)cb * H2 * W2 * 16;
  for (int q = threadIdx.x; q < GH * GW * 2; q += 256) {
    const int pix = q >> 1, hf = q & 1;
    const int ry = pix / GW, rx = pix - ry * GW;
    const int oy = 2 * y0 - 1 + ry, ox = 2 * x0 - 1 + rx;
    bf16x8_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
    if (oy >= 0 && oy < H2 && ox >= 0 && ox < W2) v = *(const bf16x8_t*)(b + ((long long)oy * W2 + ox) * 16 + hf * 8);
    *(bf16x8_t*)(tile + pix * 16 + hf * 8) = v;
  }
  __syncthreads();
  const int lx = threadIdx.x % BIL_TW, ly = threadIdx.x / BIL_TW;
  const int x = x0 + lx, y = y0 + ly;
  if (x >= w || y >= h) return;
  float wy[4], wx[4];  // weight of output row 2y-1+k (column 2x-1+k) on source row y (column x)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int t0, t1;
    float u0, u1;
    const int oy = 2 * y - 1 + k, ox = 2 * x - 1 + k;
    bil_taps16(min(max(oy, 0), H2 - 1), h, t0, t1, u0, u1);
    wy[k] = (oy >= 0 && oy < H2) ? (t0 == y ? u0 : 0.f) + (t1 == y ? u1 : 0.f) : 0.f;
    bil_taps16(min(max(ox, 0), W2 - 1), w, t0, t1, u0, u1);
    wx[k] = (ox >= 0 && ox < W2) ? (t0 == x ? u0 : 0.f) + (t1 == x ? u1 : 0.f) : 0.f;
  }
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const bf16x8_t v = *(const bf16x8_t*)(tile + ((2 * ly + ky) * GW + 2 * lx + kx) * 16 + hf * 8);
        const float wgt = wy[ky] * wx[kx];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += wgt * (float)v[e];
      }
    const long long at = (((long long)cb * h + y) * w + x) * 16 + hf * 8;
    bf16x8_t o;
    if constexpr (MASK) {
      if (plain) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[e];
        *(bf16x8_t*)(plain + n * plain_ns + at) = o;
      }
      const bf16x8_t m = *(const bf16x8_t*)(mask + n * mask_ns + at);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)m[e] > 0.f ? acc[e] : acc[e] * slope);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[e];
    }
    *(bf16x8_t*)(gsrc + n * gsrc_ns + at) = o;
  }
}

// W [cout][cin][4][4] <-> W' [cout][4 cin][3][3] (see the header comment); thread per (co, c, ky, kx)
__device__ __forceinline__ void k4_to_t3(int k, int& t, int& r) {
  t = (k + 1) >> 1;   // 0 -> 0, 1 -> 1, 2 -> 1, 3 -> 2
  r = (k + 1) & 1;    // 0 -> 1, 1 -> 0, 2 -> 1, 3 -> 0
}
__global__ void w4_to_w3_kernel(const float* __restrict__ w4, float* __restrict__ w3, int cout, int cin, int adjoint) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cout * cin * 16) return;
  const int kx = i & 3, ky = (i >> 2) & 3, c = (i >> 4) % cin, co = i / (16 * cin);
  int ty, ry, tx, rx;
  k4_to_t3(ky, ty, ry);
  k4_to_t3(kx, tx, rx);
  const long long j = (((long long)co * 4 * cin + (2 * ry + rx) * cin + c) * 3 + ty) * 3 + tx;
  if (adjoint)
    ((float*)w4)[i] = w3[j];  // gradient fold-back: dW[co][c][ky][kx] = dW'[...]
  else
    w3[j] = w4[i];
}
}  // namespace

extern "C" int sr_cb16_unshuffle2_bf16(const void* src, int64_t src_img_stride, void* dst, int64_t dst_img_stride, int n,
                                       int cblocks, int h, int w, int inverse, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(src && dst && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_cb16_unshuffle2_bf16: bad argument");
  const long long total = (long long)n * 4 * cblocks * h * w * 2;
  if (!inverse) {
    hipLaunchKernelGGL(unshuffle2_pairs_kernel, dim3(nblk(total / 2)), dim3(256), 0, stream, (const __bf16*)src, (long long)src_img_stride,
                       (__bf16*)dst, (long long)dst_img_stride, cblocks, h, w, total / 2);
    SR_CHECK_LAUNCH("cb16_unshuffle2");
    return SR_OK;
  }
  hipLaunchKernelGGL(unshuffle2_kernel, dim3(nblk(total)), dim3(256), 0, stream, (const __bf16*)src, (long long)src_img_stride,
                     (__bf16*)dst, (long long)dst_img_stride, cblocks, h, w, inverse, total);
  SR_CHECK_LAUNCH("cb16_unshuffle2");
  return SR_OK;
}

extern "C" int sr_lrelu_fwd_bf16(const void* x, void* y, float slope, int64_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && y && n > 0 && n % 8 == 0, "sr_lrelu_fwd_bf16: n must be a positive multiple of 8");
  hipLaunchKernelGGL(lrelu_fwd16_kernel, dim3(nblk(n / 8)), dim3(256), 0, stream, (const __bf16*)x, (__bf16*)y, slope,
                     (long long)(n / 8));
  SR_CHECK_LAUNCH("lrelu_fwd16");
  return SR_OK;
}

extern "C" int sr_maxpool2x2_fwd_bf16(const void* x, void* y, int n, int cblocks, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && y && n > 0 && cblocks > 0 && h >= 2 && w >= 2, "sr_maxpool2x2_fwd_bf16: bad argument");
  const long long total = (long long)n * cblocks * (h / 2) * (w / 2) * 2;
  hipLaunchKernelGGL(maxpool16_fwd_kernel, dim3(nblk(total)), dim3(256), 0, stream, (const __bf16*)x, (__bf16*)y, h, w, total);
  SR_CHECK_LAUNCH("maxpool16_fwd");
  return SR_OK;
}

extern "C" int sr_maxpool2x2_bwd_bf16(const void* x, const void* dy, void* dx, int n, int cblocks, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && dy && dx && n > 0 && cblocks > 0 && h >= 2 && w >= 2, "sr_maxpool2x2_bwd_bf16: bad argument");
  const long long total = (long long)n * cblocks * h * w * 2;
  hipLaunchKernelGGL(maxpool16_bwd_kernel, dim3(nblk(total)), dim3(256), 0, stream, (const __bf16*)x, (const __bf16*)dy, (__bf16*)dx,
                     h, w, total);
  SR_CHECK_LAUNCH("maxpool16_bwd");
  return SR_OK;
}

extern "C" int sr_cb16_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(a && b && out && n > 0 && n % 8 == 0, "sr_cb16_add_bf16: n must be a positive multiple of 8");
  hipLaunchKernelGGL(add16_kernel, dim3(nblk(n / 8)), dim3(256), 0, stream, (const __bf16*)a, (const __bf16*)b, (__bf16*)out,
                     (long long)(n / 8));
  SR_CHECK_LAUNCH("add16");
  return SR_OK;
}

extern "C" int sr_cb16_fork_bwd_bf16(const void* g_skip, const void* g_u, const void* mask, void* dz, float slope, int n, int cblocks,
                                     int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(g_u && dz && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_cb16_fork_bwd_bf16: bad argument");
  const long long total = (long long)n * cblocks * 2 * h * 2 * w * 2;
  hipLaunchKernelGGL(fork_bwd16_kernel<false>, dim3(nblk(total)), dim3(256), 0, stream, (const __bf16*)g_skip, (const __bf16*)g_u,
                     (const __bf16*)mask, (__bf16*)dz, slope, cblocks, h, w, total);
  SR_CHECK_LAUNCH("fork_bwd16");
  return SR_OK;
}

extern "C" int sr_cb16_fork_bwd_u2_bf16(const void* g_skip, const void* g_u, const void* mask_u2, void* dz, float slope, int n, int cblocks,
                                        int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(g_u && mask_u2 && dz && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_cb16_fork_bwd_u2_bf16: bad argument");
  const long long total = (long long)n * cblocks * 2 * h * 2 * w * 2;
  hipLaunchKernelGGL(fork_bwd16_kernel<true>, dim3(nblk(total)), dim3(256), 0, stream, (const __bf16*)g_skip, (const __bf16*)g_u,
                     (const __bf16*)mask_u2, (__bf16*)dz, slope, cblocks, h, w, total);
  SR_CHECK_LAUNCH("fork_bwd16 (u2 mask)");
  return SR_OK;
}

extern "C" int sr_lrelu_bwd_diff_u2_bf16(const void* gy, const void* xsum, const void* x0_u2, void* dz, float slope, int n, int cblocks,
                                         int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(gy && xsum && x0_u2 && dz && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_lrelu_bwd_diff_u2_bf16: bad argument");
  const long long total = (long long)n * cblocks * 4 * h * w * 2;
  hipLaunchKernelGGL(lrelu_bwd_diff16_u2_kernel, dim3(nblk(total)), dim3(256), 0, stream, (const __bf16*)gy, (const __bf16*)xsum,
                     (const __bf16*)x0_u2, (__bf16*)dz, slope, cblocks, h, w, total);
  SR_CHECK_LAUNCH("lrelu_bwd_diff16_u2");
  return SR_OK;
}

extern "C" int sr_lrelu_bwd_bf16(const void* gy, const void* y, void* dz, float slope, int64_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(gy && y && dz && n > 0 && n % 8 == 0, "sr_lrelu_bwd_bf16: n must be a positive multiple of 8");
  hipLaunchKernelGGL(lrelu_bwd16_kernel, dim3(nblk(n / 8)), dim3(256), 0, stream, (const __bf16*)gy, (const __bf16*)y, (__bf16*)dz,
                     slope, (long long)(n / 8));
  SR_CHECK_LAUNCH("lrelu_bwd16");
  return SR_OK;
}

extern "C" int sr_bilinear2x_fwd_bf16(const void* src, int64_t src_ns, const void* src2, int64_t src2_ns, void* dst, int64_t dst_ns,
                                      int n, int cblocks, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(src && dst && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_bilinear2x_fwd_bf16: bad argument");
  const int tiles_x = (w + BIL_TW - 1) / BIL_TW, tiles_y = (h + BIL_TH - 1) / BIL_TH;
  const long long blocks = (long long)n * cblocks * tiles_x * tiles_y;
  SR_CHECK_ARG(blocks < (1ll << 31), "sr_bilinear2x_fwd_bf16: too many tiles");
  hipLaunchKernelGGL(bilinear2x_fwd16_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, (const __bf16*)src, (long long)src_ns,
                     (const __bf16*)src2, (long long)src2_ns, (__bf16*)dst, (long long)dst_ns, cblocks, h, w, tiles_x, tiles_y);
  SR_CHECK_LAUNCH("bilinear2x_fwd16");
  return SR_OK;
}

extern "C" int sr_bilinear2x_fwd_u2_bf16(const void* src, int64_t src_ns, const void* src2_u2, int64_t src2_ns, void* dst, int64_t dst_ns,
                                         int n, int cblocks, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(src && src2_u2 && dst && n > 0 && cblocks > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0,
               "sr_bilinear2x_fwd_u2_bf16: bad argument (the source size must be even)");
  const int tiles_x = (w + BIL_TW - 1) / BIL_TW, tiles_y = (h + BIL_TH - 1) / BIL_TH;
  const long long blocks = (long long)n * cblocks * tiles_x * tiles_y;
  SR_CHECK_ARG(blocks < (1ll << 31), "sr_bilinear2x_fwd_u2_bf16: too many tiles");
  hipLaunchKernelGGL(bilinear2x_fwd16_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, (const __bf16*)src, (long long)src_ns,
                     (const __bf16*)src2_u2, (long long)src2_ns, (__bf16*)dst, (long long)dst_ns, cblocks, h, w, tiles_x, tiles_y);
  SR_CHECK_LAUNCH("bilinear2x_fwd16 (u2 skip)");
  return SR_OK;
}

extern "C" int sr_cb16_add_u2_bf16(const void* a, const void* b_u2, void* out, int n, int cblocks, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(a && b_u2 && out && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_cb16_add_u2_bf16: bad argument");
  const long long total = (long long)n * cblocks * 4 * h * w * 2;
  hipLaunchKernelGGL(add16_u2_kernel, dim3(nblk(total)), dim3(256), 0, stream, (const __bf16*)a, (const __bf16*)b_u2, (__bf16*)out, cblocks, h,
                     w, total);
  SR_CHECK_LAUNCH("add16 (u2)");
  return SR_OK;
}

extern "C" int sr_bilinear2x_bwd_bf16(const void* g, int64_t g_ns, void* gsrc, int64_t gsrc_ns, int n, int cblocks, int h, int w,
                                      void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(g && gsrc && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_bilinear2x_bwd_bf16: bad argument");
  const int tiles_x = (w + BIL_TW - 1) / BIL_TW, tiles_y = (h + BIL_TH - 1) / BIL_TH;
  const long long blocks = (long long)n * cblocks * tiles_x * tiles_y;
  SR_CHECK_ARG(blocks < (1ll << 31), "sr_bilinear2x_bwd_bf16: too many tiles");
  hipLaunchKernelGGL(bilinear2x_bwd16_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, (const __bf16*)g, (long long)g_ns,
                     (__bf16*)gsrc, (long long)gsrc_ns, cblocks, h, w, tiles_x, tiles_y, (const __bf16*)nullptr, 0ll, 1.f, (__bf16*)nullptr, 0ll);
  SR_CHECK_LAUNCH("bilinear2x_bwd16");
  return SR_OK;
}

extern "C" int sr_bilinear2x_bwd_lrelu_bf16(const void* g, int64_t g_ns, void* gsrc, int64_t gsrc_ns, const void* mask, int64_t mask_ns,
                                            float slope, void* gplain, int64_t gplain_ns, int n, int cblocks, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(g && gsrc && mask && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_bilinear2x_bwd_lrelu_bf16: bad argument");
  const int tiles_x = (w + BIL_TW - 1) / BIL_TW, tiles_y = (h + BIL_TH - 1) / BIL_TH;
  const long long blocks = (long long)n * cblocks * tiles_x * tiles_y;
  SR_CHECK_ARG(blocks < (1ll << 31), "sr_bilinear2x_bwd_lrelu_bf16: too many tiles");
  hipLaunchKernelGGL(bilinear2x_bwd16_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, (const __bf16*)g, (long long)g_ns,
                     (__bf16*)gsrc, (long long)gsrc_ns, cblocks, h, w, tiles_x, tiles_y, (const __bf16*)mask, (long long)mask_ns, slope,
                     (__bf16*)gplain, (long long)gplain_ns);
  SR_CHECK_LAUNCH("bilinear2x_bwd16 (masked)");
  return SR_OK;
}

extern "C" int sr_conv4x4s2_weight_as_3x3_f32(float* w4, float* w3, int cout, int cin, int adjoint, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(w4 && w3 && cout > 0 && cin > 0, "sr_conv4x4s2_weight_as_3x3_f32: bad argument");
  if (!adjoint && hipMemsetAsync(w3, 0, (size_t)cout * 4 * cin * 9 * sizeof(float), stream) != hipSuccess) {
    sr::set_error("sr_conv4x4s2_weight_as_3x3_f32: memset failed");
    return SR_ELAUNCH;
  }
  const int total = cout * cin * 16;
  hipLaunchKernelGGL(w4_to_w3_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w4, w3, cout, cin, adjoint);
  SR_CHECK_LAUNCH("w4_to_w3");
  return SR_OK;
}

// ------------------------------------------------------------------ BatchNorm (+LeakyReLU) on CB16
// nn.BatchNorm2d of VGGStyleDiscriminator128 (discriminator_arch.py:23-49) on bf16 activations: statistics, running
// buffers, gamma / beta and their gradients are fp32; two-pass statistics with fixed-order two-stage reductions
// (train_ops.hip, the fp32 twin).  Workspace >= sr_reduce_workspace_bytes(c).
namespace {
constexpr int RED16 = 64;  // blocks per reduced quantity (== train_ops.hip RED_SPLITS)

__device__ __forceinline__ float block_sum16(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

struct BnRed16 {
  const __bf16* x;
  const __bf16* dy;
  const __bf16* y;
  const float* mean;
  const float* invstd;
  float* part;  // [cblocks16][RED16][16][2]
  long long x_ns, dy_ns, y_ns;
  int n, hw, mode;  // 0: sum x ; 1: sum (x-mean)^2 ; 2: sum dz, sum dz*xhat
  float slope;
};
// grid (cblocks16, RED16), block 256: thread t covers 8 channels (half = t & 1) of pixels t>>1, t>>1 + 128, ...
__global__ __launch_bounds__(256) void bn_reduce16_kernel(const BnRed16 p) {
  __shared__ float sh[4];
  const int cb = blockIdx.x, sp = blockIdx.y;
  const int half = threadIdx.x & 1;
  const long long total = (long long)p.n * p.hw;
  float a[8], b[8], mu[8], is[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = b[e] = 0.f;
    mu[e] = p.mode >= 1 ? p.mean[cb * 16 + half * 8 + e] : 0.f;
    is[e] = p.mode == 2 ? p.invstd[cb * 16 + half * 8 + e] : 1.f;
  }
  for (long long i = (long long)sp * 128 + (threadIdx.x >> 1); i < total; i += 128 * RED16) {
    const int n = (int)(i / p.hw);
    const long long off = ((long long)cb * p.hw + (i - (long long)n * p.hw)) * 16 + half * 8;
    const bf16x8_t xv = *(const bf16x8_t*)(p.x + n * p.x_ns + off);
    if (p.mode == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += (float)xv[e];
    } else if (p.mode == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += ((float)xv[e] - mu[e]) * ((float)xv[e] - mu[e]);
    } else {
      const bf16x8_t gv = *(const bf16x8_t*)(p.dy + n * p.dy_ns + off), yv = *(const bf16x8_t*)(p.y + n * p.y_ns + off);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dz = (float)yv[e] > 0.f ? (float)gv[e] : (float)gv[e] * p.slope;
        a[e] += dz;
        b[e] += dz * ((float)xv[e] - mu[e]) * is[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float s0 = block_sum16(half == 0 ? a[e] : 0.f, sh), s1 = block_sum16(half == 1 ? a[e] : 0.f, sh);
    const float t0 = block_sum16(half == 0 ? b[e] : 0.f, sh), t1 = block_sum16(half == 1 ? b[e] : 0.f, sh);
    if (threadIdx.x == 0) {
      float* o = p.part + (((long long)cb * RED16 + sp) * 16) * 2;
      o[e * 2] = s0;
      o[e * 2 + 1] = t0;
      o[(8 + e) * 2] = s1;
      o[(8 + e) * 2 + 1] = t1;
    }
  }
}
// one thread per channel.  which 0: mean ; 1: var -> invstd (+ running stats) ; 2: dbeta / dgamma
__global__ void bn_finalize16_kernel(const float* part, int c, int which, long long count, float eps, float momentum, float* mean,
                                     float* invstd, float* running_mean, float* running_var, float* dgamma, float* dbeta) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const int cb = ch >> 4, e = ch & 15;
  float s = 0.f, t = 0.f;
  for (int k = 0; k < RED16; ++k) {
    const float* o = part + (((long long)cb * RED16 + k) * 16 + e) * 2;
    s += o[0];
    t += o[1];
  }
  if (which == 0) {
    mean[ch] = s / (float)count;
  } else if (which == 1) {
    const float var = s / (float)count;
    invstd[ch] = rsqrtf(var + eps);
    if (running_mean) {  // nn.BatchNorm2d: running = (1-m)*running + m*batch, with the UNBIASED variance
      const float unbiased = count > 1 ? s / (float)(count - 1) : var;
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean[ch];
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
    }
  } else {
    dbeta[ch] = s;
    dgamma[ch] = t;
  }
}
__global__ void bn_eval_prep16_kernel(const float* rm, const float* rv, float eps, int c, float* mean, float* invstd) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch < c) {
    mean[ch] = rm[ch];
    invstd[ch] = rsqrtf(rv[ch] + eps);
  }
}
// fwd: y = lrelu((x - mean)*invstd*gamma + beta) ; bwd: dx = gamma*invstd*(dz - [train] (dbeta + xhat*dgamma)/M)
__global__ void bn_lrelu16_kernel(const __bf16* __restrict__ x, long long x_ns, const __bf16* __restrict__ dy, long long dy_ns,
                                  const __bf16* __restrict__ yin, long long yin_ns, __bf16* __restrict__ out, long long out_ns,
                                  const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                  float slope, int backward, int train, float inv_count, int c, int cblocks, int hw, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int pix = (int)(r % hw);
  r /= hw;
  const int cb = (int)(r % cblocks), n = (int)(r / cblocks);
  const long long off = ((long long)cb * hw + pix) * 16 + half * 8;
  const bf16x8_t xv = *(const bf16x8_t*)(x + n * x_ns + off);
  bf16x8_t gv, yv, o;
  if (backward) {
    gv = *(const bf16x8_t*)(dy + n * dy_ns + off);
    yv = *(const bf16x8_t*)(yin + n * yin_ns + off);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = cb * 16 + half * 8 + e;
    float v = 0.f;
    if (ch < c) {
      const float xhat = ((float)xv[e] - mean[ch]) * invstd[ch];
      if (!backward) {
        v = xhat * gamma[ch] + beta[ch];
        v = v > 0.f ? v : v * slope;
      } else {
        const float dz = (float)yv[e] > 0.f ? (float)gv[e] : (float)gv[e] * slope;
        v = train ? gamma[ch] * invstd[ch] * (dz - (dbeta[ch] + xhat * dgamma[ch]) * inv_count) : gamma[ch] * invstd[ch] * dz;
      }
    }
    o[e] = (__bf16)v;
  }
  *(bf16x8_t*)(out + n * out_ns + off) = o;
}
}  // namespace

extern "C" int sr_bn_lrelu_fwd_bf16(const void* x, int64_t x_ns, void* y, int64_t y_ns, int n, int c, int h, int w, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, int train, float momentum,
                                    float eps, float slope, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && y && gamma && beta && save_mean && save_invstd && ws && n > 0 && c > 0 && h > 0 && w > 0,
               "sr_bn_lrelu_fwd_bf16: bad argument");
  SR_CHECK_ARG(train || (running_mean && running_var), "sr_bn_lrelu_fwd_bf16: eval mode needs running statistics");
  SR_CHECK_ARG(ws_bytes >= sr_reduce_workspace_bytes(c), "sr_bn_lrelu_fwd_bf16: workspace too small");
  const int cblocks = (c + 15) / 16, hw = h * w;
  const long long count = (long long)n * hw;
  if (train && count <= bnsmall::kBnSmallPixels && sr::bn_small_enabled()) {  // one launch for the whole pass (bn_small.h)
    bnsmall::Params<__bf16> q = {};
    q.x = (const __bf16*)x;
    q.out = (__bf16*)y;
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
    bnsmall::launch<__bf16, 16, false>(q, stream);
    SR_CHECK_LAUNCH("bn_small fwd16");
    return SR_OK;
  }
  float* part = (float*)ws;
  if (train) {
    BnRed16 p = {};
    p.x = (const __bf16*)x;
    p.x_ns = x_ns;
    p.n = n;
    p.hw = hw;
    p.part = part;
    p.mode = 0;
    hipLaunchKernelGGL(bn_reduce16_kernel, dim3(cblocks, RED16), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(bn_finalize16_kernel, dim3(nblk(c)), dim3(256), 0, stream, part, c, 0, count, eps, momentum, save_mean,
                       save_invstd, nullptr, nullptr, nullptr, nullptr);
    p.mode = 1;
    p.mean = save_mean;
    hipLaunchKernelGGL(bn_reduce16_kernel, dim3(cblocks, RED16), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(bn_finalize16_kernel, dim3(nblk(c)), dim3(256), 0, stream, part, c, 1, count, eps, momentum, save_mean,
                       save_invstd, running_mean, running_var, nullptr, nullptr);
  } else {
    hipLaunchKernelGGL(bn_eval_prep16_kernel, dim3(nblk(c)), dim3(256), 0, stream, running_mean, running_var, eps, c, save_mean,
                       save_invstd);
  }
  const long long total = (long long)n * cblocks * hw * 2;
  hipLaunchKernelGGL(bn_lrelu16_kernel, dim3(nblk(total)), dim3(256), 0, stream, (const __bf16*)x, (long long)x_ns, nullptr, 0ll,
                     nullptr, 0ll, (__bf16*)y, (long long)y_ns, save_mean, save_invstd, gamma, beta, nullptr, nullptr, slope, 0, train,
                     0.f, c, cblocks, hw, total);
  SR_CHECK_LAUNCH("bn_lrelu_fwd16");
  return SR_OK;
}

extern "C" int sr_bn_lrelu_bwd_bf16(const void* x, int64_t x_ns, const void* dy, int64_t dy_ns, const void* y, int64_t y_ns, void* dx,
                                    int64_t dx_ns, int n, int c, int h, int w, const float* gamma, const float* save_mean,
                                    const float* save_invstd, int train, float slope, float* dgamma, float* dbeta, void* ws,
                                    size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(x && dy && y && dx && gamma && save_mean && save_invstd && dgamma && dbeta && ws && n > 0 && c > 0,
               "sr_bn_lrelu_bwd_bf16: bad argument");
  SR_CHECK_ARG(ws_bytes >= sr_reduce_workspace_bytes(c), "sr_bn_lrelu_bwd_bf16: workspace too small");
  const int cblocks = (c + 15) / 16, hw = h * w;
  const long long count = (long long)n * hw;
  if (train && count <= bnsmall::kBnSmallPixels && sr::bn_small_enabled()) {
    bnsmall::Params<__bf16> q = {};
    q.x = (const __bf16*)x;
    q.dy = (const __bf16*)dy;
    q.y = (const __bf16*)y;
    q.out = (__bf16*)dx;
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
    bnsmall::launch<__bf16, 16, true>(q, stream);
    SR_CHECK_LAUNCH("bn_small bwd16");
    return SR_OK;
  }
  float* part = (float*)ws;
  BnRed16 p = {};
  p.x = (const __bf16*)x;
  p.x_ns = x_ns;
  p.dy = (const __bf16*)dy;
  p.dy_ns = dy_ns;
  p.y = (const __bf16*)y;
  p.y_ns = y_ns;
  p.mean = save_mean;
  p.invstd = save_invstd;
  p.n = n;
  p.hw = hw;
  p.part = part;
  p.mode = 2;
  p.slope = slope;
  hipLaunchKernelGGL(bn_reduce16_kernel, dim3(cblocks, RED16), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(bn_finalize16_kernel, dim3(nblk(c)), dim3(256), 0, stream, part, c, 2, count, 0.f, 0.f, nullptr, nullptr, nullptr,
                     nullptr, dgamma, dbeta);
  const long long total = (long long)n * cblocks * hw * 2;
  hipLaunchKernelGGL(bn_lrelu16_kernel, dim3(nblk(total)), dim3(256), 0, stream, (const __bf16*)x, (long long)x_ns, (const __bf16*)dy,
                     (long long)dy_ns, (const __bf16*)y, (long long)y_ns, (__bf16*)dx, (long long)dx_ns, save_mean, save_invstd, gamma,
                     nullptr, dgamma, dbeta, slope, 1, train, 1.f / (float)count, c, cblocks, hw, total);
  SR_CHECK_LAUNCH("bn_lrelu_bwd16");
  return SR_OK;
}
