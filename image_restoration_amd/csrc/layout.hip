// Layout conversion and weight packing kernels (HBM-bound, tiny next to the convs).
//   NCHW <-> CB8 at the module boundary (caller tensors are NCHW),
//   with the reference's pixel_unshuffle (arch_util.py:185-201) fused into the load;
//   OIHW conv weights -> MFMA A-operand image [group][cin/8][tap][group couts][8].
#include "sr_internal.h"

namespace sr {
int conv_group_couts(int cout);
}

namespace {

// One thread per destination pixel-block: 8 channels = 32 B written, 8 strided plane reads.
__global__ void nchw_to_cb8_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W, int u,
                                   int cblocks, long long dst_ns, long long total) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int HW = H * W;
  const int pix = (int)(i % HW);
  long long r = i / HW;
  const int cb = (int)(r % cblocks);
  const int n = (int)(r / cblocks);
  const int y = pix / W, x = pix - y * W;
  const int Cu = C * u * u;
  const int SH = H * u, SW = W * u;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cb * 8 + e;
    float val = 0.f;
    if (c < Cu) {
      // pixel_unshuffle channel order: c = (c_src*u + iy)*u + ix   (arch_util.py:200-201)
      const int ix = c % u, iy = (c / u) % u, cs = c / (u * u);
      val = src[((long long)(n * C + cs) * SH + (y * u + iy)) * SW + (x * u + ix)];
    }
    v[e] = val;
  }
  float4* o = (float4*)(dst + n * dst_ns + ((long long)cb * HW + pix) * 8);
  o[0] = make_float4(v[0], v[1], v[2], v[3]);
  o[1] = make_float4(v[4], v[5], v[6], v[7]);
}

__global__ void cb8_to_nchw_kernel(const float* __restrict__ src, long long src_ns, float* __restrict__ dst, int C, int H,
                                   int W, long long total) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int HW = H * W;
  const int pix = (int)(i % HW);
  long long r = i / HW;
  const int c = (int)(r % C);
  const int n = (int)(r / C);
  dst[i] = src[n * src_ns + ((long long)(c >> 3) * HW + pix) * 8 + (c & 7)];
}

// mode 0: thread per (co, ci, tap) of the OIHW weight, scattered into the (pre-zeroed) image
//         wp[g][cb][tap][co % gc][c8],  g = co / gc, (cb, c8) = position of ci in the CB8 source.
// mode 1: data-gradient image: "output channel" = position of ci, "input channel" = co, tap flipped.
__global__ void pack_w_kernel(const float* __restrict__ w, int cout, int cin, int first_seg, int seg, int cin_pad,
                              int mode, float* __restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cout * cin * 9) return;
  const int tap = i % 9;
  const int ci = (i / 9) % cin;
  const int co = i / (9 * cin);
  int pos = ci;
  if (ci >= first_seg) {
    const int r = ci - first_seg;
    pos = (first_seg + 7) / 8 * 8 + (r / seg) * ((seg + 7) / 8 * 8) + r % seg;
  }
  const float val = w[i];
  if (mode == 0) {
    const int gc = (((cout + 31) / 32 * 32) % 64 == 0) ? 64 : 32;
    const int cbs = cin_pad / 8;
    const int g = co / gc, col = co % gc;
    wp[((((long long)g * cbs + (pos >> 3)) * 9 + tap) * gc + col) * 8 + (pos & 7)] = val;
  } else {
    const int oc = cin_pad;  // output channels of the dgrad conv
    const int gc = (((oc + 31) / 32 * 32) % 64 == 0) ? 64 : 32;
    const int cbs = (cout + 7) / 8;
    const int g = pos / gc, col = pos % gc;
    wp[((((long long)g * cbs + (co >> 3)) * 9 + (8 - tap)) * gc + col) * 8 + (co & 7)] = val;
  }
}

__global__ void pack_b_kernel(const float* __restrict__ b, int cout, int cpad, float* __restrict__ bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cpad) bp[i] = (b && i < cout) ? b[i] : 0.f;
}

}  // namespace

extern "C" int sr_nchw_to_cb8_f32(const float* src, float* dst, int N, int C, int H, int W, int unshuffle,
                                  int dst_cblocks, int64_t dst_img_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "sr_nchw_to_cb8_f32: bad argument");
  SR_CHECK_ARG(unshuffle == 1 || unshuffle == 2 || unshuffle == 4, "sr_nchw_to_cb8_f32: unshuffle must be 1, 2 or 4");
  SR_CHECK_ARG(dst_cblocks * 8 >= C * unshuffle * unshuffle, "sr_nchw_to_cb8_f32: dst_cblocks too small");
  SR_CHECK_ARG((uintptr_t)dst % 16 == 0 && dst_img_stride % 4 == 0, "sr_nchw_to_cb8_f32: dst must be 16-byte aligned");
  const long long total = (long long)N * dst_cblocks * H * W;
  hipLaunchKernelGGL(nchw_to_cb8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, dst, C, H, W,
                     unshuffle, dst_cblocks, (long long)dst_img_stride, total);
  SR_CHECK_LAUNCH("nchw_to_cb8");
  return SR_OK;
}

extern "C" int sr_cb8_to_nchw_f32(const float* src, int64_t src_img_stride, float* dst, int N, int C, int H, int W,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "sr_cb8_to_nchw_f32: bad argument");
  const long long total = (long long)N * C * H * W;
  hipLaunchKernelGGL(cb8_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src,
                     (long long)src_img_stride, dst, C, H, W, total);
  SR_CHECK_LAUNCH("cb8_to_nchw");
  return SR_OK;
}

extern "C" size_t sr_conv3x3_packed_weight_floats(int cout, int cin_pad) {
  const int cp = (cout + 31) / 32 * 32;
  return (size_t)cp * (size_t)cin_pad * 9;
}

extern "C" size_t sr_conv3x3_packed_bias_floats(int cout) { return (size_t)((cout + 31) / 32 * 32); }

extern "C" int sr_conv3x3_cin_pad(int cin, int first_seg, int seg) {
  if (cin <= 0 || first_seg <= 0 || first_seg > cin) return SR_EINVAL;
  if (first_seg == cin) return (cin + 7) / 8 * 8;
  if (seg <= 0 || (cin - first_seg) % seg != 0) return SR_EINVAL;
  return (first_seg + 7) / 8 * 8 + ((cin - first_seg) / seg) * ((seg + 7) / 8 * 8);
}

extern "C" int sr_conv3x3_pack_f32(const float* weight, const float* bias, int cout, int cin, int first_seg, int seg,
                                   int mode, float* wpacked, float* bpacked, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(weight && wpacked && cout > 0 && cin > 0, "sr_conv3x3_pack_f32: bad argument");
  const int cin_pad = sr_conv3x3_cin_pad(cin, first_seg, seg);
  SR_CHECK_ARG(cin_pad > 0, "sr_conv3x3_pack_f32: cin=%d is not first_seg=%d + k*seg=%d", cin, first_seg, seg);
  SR_CHECK_ARG(mode == 0 || mode == 1, "sr_conv3x3_pack_f32: mode must be 0 or 1");
  const size_t wfloats = mode == 0 ? sr_conv3x3_packed_weight_floats(cout, cin_pad)
                                   : sr_conv3x3_packed_weight_floats(cin_pad, (cout + 7) / 8 * 8);
  if (hipMemsetAsync(wpacked, 0, wfloats * sizeof(float), stream) != hipSuccess) {
    sr::set_error("sr_conv3x3_pack_f32: memset failed");
    return SR_ELAUNCH;
  }
  const int total = cout * cin * 9;
  hipLaunchKernelGGL(pack_w_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, weight, cout, cin, first_seg,
                     seg > 0 ? seg : 1, cin_pad, mode, wpacked);
  SR_CHECK_LAUNCH("pack_w");
  if (bpacked && mode == 0) {
    const int cp = (int)sr_conv3x3_packed_bias_floats(cout);
    hipLaunchKernelGGL(pack_b_kernel, dim3((cp + 255) / 256), dim3(256), 0, stream, bias, cout, cp, bpacked);
    SR_CHECK_LAUNCH("pack_b");
  }
  return SR_OK;
}
