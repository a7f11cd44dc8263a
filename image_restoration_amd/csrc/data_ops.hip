// Device side of the training input pipeline: paired crop window + flip / transpose symmetry + uint8 -> float32
// conversion + channel swap + per-channel normalisation of a whole batch in ONE launch per tensor.
//
// Reference, per sample on host cores: paired_random_crop and augment (basicsr/data/transforms.py:26-158), then img2tensor
// (utils/img_util.py:9-35: BGR->RGB, HWC->CHW, float32) on imfrombytes(float32=True) images (uint8 / 255, :128-132) and
// torchvision's normalize (paired_image_dataset.py:94-96).  All of it is index arithmetic plus three IEEE float operations per
// value (u/255, -mean, /std), so the device result is bit-identical to the host pipeline given the same random draws; the host
// only draws (data/transforms.py draw_window / draw_symmetry) and ships uint8: a quarter of the PCIe bytes and no float work.
// HBM-bound: 3 B read + 12 B written per output pixel.
#include "sr_internal.h"

namespace {

struct AugParams {
  const unsigned char* src;  // [n][src_h][src_w][3] uint8, channel order as decoded (BGR)
  long long src_ns;          // bytes between images
  const int* top;            // [n] window origin in SOURCE pixels, or null (0)
  const int* left;
  const int* sym;            // [n] bit0 hflip, bit1 vflip, bit2 transpose (applied in this order), or null (0)
  float* dst;                // [n][3][oh][ow] float32
  int src_h, src_w, ph, pw;  // window size in the source
  int origin_mul;            // top / left are multiplied by this (LQ coordinates driving the GT tensor: the scale)
  int swap_rb;
  float mean[3], inv_scale, stdv[3];
  int normalise;
};

__global__ __launch_bounds__(256) void patch_augment_kernel(const AugParams p) {
  const int n = blockIdx.z;
  const int code = p.sym ? p.sym[n] : 0;
  const bool tr = code & 4;
  const int oh = tr ? p.pw : p.ph, ow = tr ? p.ph : p.pw;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= ow || y >= oh) return;
  // undo transpose, vertical flip, horizontal flip (they were applied as hflip -> vflip -> transpose)
  int wy = tr ? x : y, wx = tr ? y : x;
  if (code & 2) wy = p.ph - 1 - wy;
  if (code & 1) wx = p.pw - 1 - wx;
  const int sy = (p.top ? p.top[n] * p.origin_mul : 0) + wy, sx = (p.left ? p.left[n] * p.origin_mul : 0) + wx;
  const unsigned char* s = p.src + (long long)n * p.src_ns + ((long long)sy * p.src_w + sx) * 3;
  float* d = p.dst + ((long long)n * 3 * oh + y) * ow + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = (float)s[p.swap_rb ? 2 - c : c] / 255.0f;  // IEEE division like numpy's  img.astype(float32) / 255.
    if (p.normalise) v = (v - p.mean[c]) / p.stdv[c];
    d[(long long)c * oh * ow] = v;
  }
}

}  // namespace

extern "C" int sr_patch_augment_u8_f32(const uint8_t* src, int64_t src_img_stride, int src_h, int src_w, const int32_t* top,
                                       const int32_t* left, int origin_mul, const int32_t* sym, float* dst, int n, int patch_h,
                                       int patch_w, int swap_rb, const float* host_mean3, const float* host_std3, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(src && dst && n > 0 && src_h > 0 && src_w > 0 && patch_h > 0 && patch_w > 0 && patch_h <= src_h && patch_w <= src_w,
               "sr_patch_augment_u8_f32: bad shape (window %dx%d in %dx%d)", patch_h, patch_w, src_h, src_w);
  SR_CHECK_ARG(!sym || patch_h == patch_w, "sr_patch_augment_u8_f32: transposition needs a square window");
  SR_CHECK_ARG((top == nullptr) == (left == nullptr) && origin_mul >= 1, "sr_patch_augment_u8_f32: top / left go together");
  SR_CHECK_ARG(n <= 65535, "sr_patch_augment_u8_f32: batch too large for one launch");
  AugParams p = {};
  p.src = src;
  p.src_ns = src_img_stride > 0 ? src_img_stride : (long long)src_h * src_w * 3;
  p.top = top;
  p.left = left;
  p.sym = sym;
  p.dst = dst;
  p.src_h = src_h;
  p.src_w = src_w;
  p.ph = patch_h;
  p.pw = patch_w;
  p.origin_mul = origin_mul;
  p.swap_rb = swap_rb;
  p.normalise = (host_mean3 || host_std3) ? 1 : 0;
  for (int c = 0; c < 3; ++c) {
    p.mean[c] = host_mean3 ? host_mean3[c] : 0.f;
    p.stdv[c] = host_std3 ? host_std3[c] : 1.f;
  }
  const int side = patch_h > patch_w ? patch_h : patch_w;
  hipLaunchKernelGGL(patch_augment_kernel, dim3(sr::cdiv(side, 64), sr::cdiv(side, 4), n), dim3(256), 0, stream, p);
  SR_CHECK_LAUNCH("patch_augment");
  return SR_OK;
}
