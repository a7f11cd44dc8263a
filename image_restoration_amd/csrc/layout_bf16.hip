// CB16 (bf16) layout conversions, weight images and the small HBM-bound helpers of the bf16 path.
//
// The reference has no reduced precision; BASELINE configs 3-4 name bf16.  Activations and their
// gradients are  __bf16 feat[N][C/16][H][W][16]  (32-byte pixels, like CB8), weights stay fp32 masters and are
// rounded once per optimiser step into the MFMA images packed here; every reduction and epilogue is fp32.
#include "sr_internal.h"

namespace {
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int pos16(int ci, int first_seg, int seg) {  // position of reference channel ci in a CB16 concat
  if (ci < first_seg) return ci;
  const int r = ci - first_seg;
  return (first_seg + 15) / 16 * 16 + (r / seg) * ((seg + 15) / 16 * 16) + r % seg;
}
__host__ __device__ __forceinline__ int group_cout(int cout) { return (((cout + 31) / 32 * 32) % 64 == 0) ? 64 : 32; }

// NCHW fp32 -> CB16 bf16 (pixel_unshuffle fused like nchw_to_cb8): one thread per destination half pixel-block (8 ch)
__global__ void nchw_to_cb16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int C, int H, int W, int u,
                                    int cblocks, long long dst_ns, long long total) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int HW = H * W;
  const int pix = (int)(r % HW);
  r /= HW;
  const int cb = (int)(r % cblocks), n = (int)(r / cblocks);
  const int y = pix / W, x = pix - y * W;
  const int Cu = C * u * u, SH = H * u, SW = W * u;
  bf16x8_t v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cb * 16 + half * 8 + e;
    float val = 0.f;
    if (c < Cu) {
      const int ix = c % u, iy = (c / u) % u, cs = c / (u * u);
      val = src[((long long)(n * C + cs) * SH + (y * u + iy)) * SW + (x * u + ix)];
    }
    v[e] = (__bf16)val;
  }
  *(bf16x8_t*)(dst + n * dst_ns + ((long long)cb * HW + pix) * 16 + half * 8) = v;
}

// CB16 bf16 -> NCHW fp32, pixel_shuffle by u fused (the adjoint of the unshuffle above); thread per destination element
__global__ void cb16_to_nchw_kernel(const __bf16* __restrict__ src, long long src_ns, float* __restrict__ dst, int C, int H,
                                    int W, int u, long long total) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int SW = W * u, SH = H * u;
  const int sx = (int)(i % SW);
  long long r = i / SW;
  const int sy = (int)(r % SH);
  r /= SH;
  const int cs = (int)(r % C), n = (int)(r / C);
  const int c = (cs * u + sy % u) * u + sx % u;
  const long long pix = (long long)(sy / u) * W + sx / u;
  dst[i] = (float)src[n * src_ns + ((long long)(c >> 4) * H * W + pix) * 16 + (c & 15)];
}

// mode 0: OIHW fp32 -> bf16 image wp[g][cb16][tap][gc][16] (g = co / gc; (cb16, c16) = position of ci in the source)
// mode 1: data-gradient image: "output channel" = position of ci, "input channel" = co, taps flipped
__global__ void pack_w16_kernel(const float* __restrict__ w, int cout, int cin, int first_seg, int seg, int cin_pad, int mode,
                                __bf16* __restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cout * cin * 9) return;
  const int tap = i % 9, ci = (i / 9) % cin, co = i / (9 * cin);
  const int pos = pos16(ci, first_seg, seg);
  const __bf16 val = (__bf16)w[i];
  if (mode == 0) {
    const int gc = group_cout(cout), cbs = cin_pad / 16;
    const int g = co / gc, col = co % gc;
    wp[((((long long)g * cbs + (pos >> 4)) * 9 + tap) * gc + col) * 16 + (pos & 15)] = val;
  } else {
    const int gc = group_cout(cin_pad), cbs = (cout + 15) / 16;
    const int g = pos / gc, col = pos % gc;
    wp[((((long long)g * cbs + (co >> 4)) * 9 + (8 - tap)) * gc + col) * 16 + (co & 15)] = val;
  }
}

__global__ void pack_b16_kernel(const float* __restrict__ b, int cout, int cpad, float* __restrict__ bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cpad) bp[i] = (b && i < cout) ? b[i] : 0.f;
}

// Transposed dense block (see layout.hip, pack_dense_dgrad_kernel) in bf16 / CB16 positions.
struct DensePack16 {
  const float* w[5];
  __bf16* out;
  int nf, gc, nfp, gcp, s;
  float scale5;
};
__global__ void pack_dense_dgrad16_kernel(const DensePack16 p) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int slice = p.s == 0 ? p.nf : p.gc;
  const int slice0 = p.s == 0 ? 0 : p.nf + (p.s - 1) * p.gc;
  const int cinp = p.nfp + (4 - p.s) * p.gcp;
  for (int k = 5; k > p.s; --k) {
    const int cout_k = k == 5 ? p.nf : p.gc, cin_k = p.nf + (k - 1) * p.gc;
    const long long cnt = (long long)cout_k * slice * 9;
    if (i < cnt) {
      const int tap = (int)(i % 9);
      const int cil = (int)((i / 9) % slice);
      const int co = (int)(i / (9LL * slice));
      float v = p.w[k - 1][((long long)co * cin_k + slice0 + cil) * 9 + tap];
      if (k == 5) v *= p.scale5;
      const int pos = k == 5 ? co : p.nfp + (4 - k) * p.gcp + co;
      const int gcw = group_cout(slice), cbs = cinp / 16;
      const int g = cil / gcw, col = cil % gcw;
      p.out[((((long long)g * cbs + (pos >> 4)) * 9 + (8 - tap)) * gcw + col) * 16 + (pos & 15)] = (__bf16)v;
      return;
    }
    i -= cnt;
  }
}

// 2x2 sum (backward of the nearest x2 upsample) + optional LeakyReLU backward; thread per destination half pixel-block
__global__ void up2x_bwd16_kernel(const __bf16* __restrict__ g, long long g_ns, __bf16* __restrict__ dst, long long dst_ns,
                                  const __bf16* __restrict__ mask, long long mask_ns, float slope, int cblocks, int h, int w,
                                  long long total) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int half = (int)(i & 1);
  long long r = i >> 1;
  const int x = (int)(r % w);
  r /= w;
  const int y = (int)(r % h);
  r /= h;
  const int cb = (int)(r % cblocks), n = (int)(r / cblocks);
  const int W2 = 2 * w;
  const __bf16* s = g + n * g_ns + (((long long)cb * 2 * h + 2 * y) * W2 + 2 * x) * 16 + half * 8;
  const bf16x8_t a = *(const bf16x8_t*)s, b = *(const bf16x8_t*)(s + 16), c = *(const bf16x8_t*)(s + (long long)W2 * 16),
                 d = *(const bf16x8_t*)(s + (long long)W2 * 16 + 16);
  const long long off = (((long long)cb * h + y) * w + x) * 16 + half * 8;
  bf16x8_t m;
  if (mask) m = *(const bf16x8_t*)(mask + n * mask_ns + off);
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = (float)a[e] + (float)b[e] + (float)c[e] + (float)d[e];
    if (mask && !((float)m[e] > 0.f)) v *= slope;
    o[e] = (__bf16)v;
  }
  *(bf16x8_t*)(dst + n * dst_ns + off) = o;
}

__global__ void cb16_axpby_kernel(__bf16* __restrict__ dst, long long dst_ns, const __bf16* __restrict__ src, long long src_ns,
                                  float a, float b, long long per_img8, long long total) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long n = i / per_img8, o = (i % per_img8) * 8;
  bf16x8_t* d = (bf16x8_t*)(dst + n * dst_ns + o);
  const bf16x8_t s = *(const bf16x8_t*)(src + n * src_ns + o);
  bf16x8_t v = *d;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (__bf16)(a * (float)v[e] + b * (float)s[e]);
  *d = v;
}
}  // namespace

extern "C" int sr_nchw_to_cb16_bf16(const float* src, void* dst, int N, int C, int H, int W, int unshuffle, int dst_cblocks,
                                    int64_t dst_img_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "sr_nchw_to_cb16_bf16: bad argument");
  SR_CHECK_ARG(unshuffle == 1 || unshuffle == 2 || unshuffle == 4, "sr_nchw_to_cb16_bf16: unshuffle must be 1, 2 or 4");
  SR_CHECK_ARG(dst_cblocks * 16 >= C * unshuffle * unshuffle, "sr_nchw_to_cb16_bf16: dst_cblocks too small");
  const long long total = (long long)N * dst_cblocks * H * W * 2;
  hipLaunchKernelGGL(nchw_to_cb16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, (__bf16*)dst, C, H,
                     W, unshuffle, dst_cblocks, (long long)dst_img_stride, total);
  SR_CHECK_LAUNCH("nchw_to_cb16");
  return SR_OK;
}

extern "C" int sr_cb16_to_nchw_f32(const void* src, int64_t src_img_stride, float* dst, int N, int C, int H, int W, int shuffle,
                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "sr_cb16_to_nchw_f32: bad argument");
  SR_CHECK_ARG(shuffle == 1 || shuffle == 2 || shuffle == 4, "sr_cb16_to_nchw_f32: shuffle must be 1, 2 or 4");
  const long long total = (long long)N * C * H * W * shuffle * shuffle;
  hipLaunchKernelGGL(cb16_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const __bf16*)src,
                     (long long)src_img_stride, dst, C, H, W, shuffle, total);
  SR_CHECK_LAUNCH("cb16_to_nchw");
  return SR_OK;
}

extern "C" int sr_conv3x3_cin_pad16(int cin, int first_seg, int seg) {
  if (cin <= 0 || first_seg <= 0 || first_seg > cin) return SR_EINVAL;
  if (first_seg == cin) return (cin + 15) / 16 * 16;
  if (seg <= 0 || (cin - first_seg) % seg != 0) return SR_EINVAL;
  return (first_seg + 15) / 16 * 16 + ((cin - first_seg) / seg) * ((seg + 15) / 16 * 16);
}

extern "C" size_t sr_conv3x3_packed_weight_elems_bf16(int cout, int cin, int first_seg, int seg, int mode) {
  const int cin_pad = sr_conv3x3_cin_pad16(cin, first_seg, seg);
  if (cin_pad <= 0 || cout <= 0) return 0;
  if (mode == 0) return (size_t)((cout + 31) / 32 * 32) * cin_pad * 9;
  return (size_t)((cin_pad + 31) / 32 * 32) * ((cout + 15) / 16 * 16) * 9;
}

extern "C" int sr_conv3x3_pack_bf16(const float* weight, const float* bias, int cout, int cin, int first_seg, int seg, int mode,
                                    void* wpacked, float* bpacked, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(weight && wpacked && cout > 0 && cin > 0 && (mode == 0 || mode == 1), "sr_conv3x3_pack_bf16: bad argument");
  const int cin_pad = sr_conv3x3_cin_pad16(cin, first_seg, seg);
  SR_CHECK_ARG(cin_pad > 0, "sr_conv3x3_pack_bf16: cin=%d is not first_seg=%d + k*seg=%d", cin, first_seg, seg);
  const size_t elems = sr_conv3x3_packed_weight_elems_bf16(cout, cin, first_seg, seg, mode);
  if (hipMemsetAsync(wpacked, 0, elems * 2, stream) != hipSuccess) {
    sr::set_error("sr_conv3x3_pack_bf16: memset failed");
    return SR_ELAUNCH;
  }
  const int total = cout * cin * 9;
  hipLaunchKernelGGL(pack_w16_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, weight, cout, cin, first_seg,
                     seg > 0 ? seg : 1, cin_pad, mode, (__bf16*)wpacked);
  SR_CHECK_LAUNCH("pack_w16");
  if (bpacked && mode == 0) {
    const int cp = (int)sr_conv3x3_packed_bias_floats(cout);
    hipLaunchKernelGGL(pack_b16_kernel, dim3((cp + 255) / 256), dim3(256), 0, stream, bias, cout, cp, bpacked);
    SR_CHECK_LAUNCH("pack_b16");
  }
  return SR_OK;
}

namespace sr {
size_t rdb_dgrad_step_elems16(int nf, int gc, int s) {
  const int nfp = (nf + 15) / 16 * 16, gcp = (gc + 15) / 16 * 16;
  const int slice = s == 0 ? nf : gc;
  return (size_t)((slice + 31) / 32 * 32) * (nfp + (4 - s) * gcp) * 9;
}
int rdb_pack_dgrad_step_bf16(const float* const w[5], int nf, int gc, int s, float scale5, void* out, hipStream_t stream) {
  if (hipMemsetAsync(out, 0, rdb_dgrad_step_elems16(nf, gc, s) * 2, stream) != hipSuccess) {
    set_error("rdb_pack_dgrad_step_bf16: memset failed");
    return SR_ELAUNCH;
  }
  DensePack16 p;
  for (int i = 0; i < 5; ++i) p.w[i] = w[i];
  p.out = (__bf16*)out;
  p.nf = nf;
  p.gc = gc;
  p.nfp = (nf + 15) / 16 * 16;
  p.gcp = (gc + 15) / 16 * 16;
  p.s = s;
  p.scale5 = scale5;
  const int slice = s == 0 ? nf : gc;
  long long total = 0;
  for (int k = 5; k > s; --k) total += (long long)(k == 5 ? nf : gc) * slice * 9;
  hipLaunchKernelGGL(pack_dense_dgrad16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
  SR_CHECK_LAUNCH("pack_dense_dgrad16");
  return SR_OK;
}
}  // namespace sr

extern "C" int sr_upsample2x_bwd_bf16(const void* g, int64_t g_img_stride, void* dst, int64_t dst_img_stride, const void* mask,
                                      int64_t mask_img_stride, float mask_slope, int n, int cblocks, int h, int w,
                                      void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(g && dst && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_upsample2x_bwd_bf16: bad argument");
  const long long total = (long long)n * cblocks * h * w * 2;
  hipLaunchKernelGGL(up2x_bwd16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const __bf16*)g,
                     (long long)g_img_stride, (__bf16*)dst, (long long)dst_img_stride, (const __bf16*)mask,
                     (long long)mask_img_stride, mask_slope, cblocks, h, w, total);
  SR_CHECK_LAUNCH("up2x_bwd16");
  return SR_OK;
}

extern "C" int sr_cb16_axpby_bf16(void* dst, int64_t dst_img_stride, const void* src, int64_t src_img_stride, float a, float b,
                                  int n, int cblocks, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(dst && src && n > 0 && cblocks > 0 && h > 0 && w > 0, "sr_cb16_axpby_bf16: bad argument");
  const long long per_img8 = (long long)cblocks * h * w * 2;
  const long long total = per_img8 * n;
  hipLaunchKernelGGL(cb16_axpby_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (__bf16*)dst,
                     (long long)dst_img_stride, (const __bf16*)src, (long long)src_img_stride, a, b, per_img8, total);
  SR_CHECK_LAUNCH("cb16_axpby");
  return SR_OK;
}
