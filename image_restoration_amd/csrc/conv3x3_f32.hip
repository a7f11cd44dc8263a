// Fused 3x3 convolution on fp32 MFMA for gfx950 (MI355X).
//
// Replaces, on the RRDBNet path, nn.Conv2d(k3,s1,p1) + LeakyReLU(0.2) + torch.cat +
// the 0.2 residual scale-adds of the reference
// (Car_Plate-Restoration/basicsr/archs/rrdbnet_arch.py:32-39, :58-63, :112-118).
//
// Formulation: implicit GEMM  D[cout][pixel] += W[cout][k] * X[k][pixel],  k = (cin, tap),
// on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).  A = weights, B = activations, so
// that the 32x32 accumulator has the PIXEL on the lane and 4 consecutive couts in 4
// consecutive registers: one accumulator quad is a 16-byte store into the CB8 layout
// [N][C/8][H][W][8], and 64 lanes write 1 KiB contiguous.
//
// Workgroup = 4 waves; output tile = (4*PT rows) x 32 columns x (32*COT couts).
// Wave w owns PT rows.  K is walked in chunks of one 8-channel CB8 block:
//   LDS X image  [TH+2][34][8] floats   (halo'd tile, pixel = 32 B, rows contiguous in HBM)
//   LDS W image  [9 taps][32*COT][8]    (pre-packed, contiguous in HBM)
// both filled by global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction), double
// buffered, one barrier per chunk.  Zero padding and the nearest-x2 upsample of the head
// (rrdbnet_arch.py:116-117) are per-lane SOURCE addresses of the DMA (pad lanes read a
// 16-byte zero line), so neither costs a pass over memory.
// Per chunk and wave: 9*(COT+PT) ds_read_b128 feed 36*COT*PT MFMAs (64 cycles each).
#include "sr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __attribute__((aligned(64))) float g_zero_line[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

struct ConvParams {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  const float* res1;
  const float* res2;
  const float* mask;
  long long in_ns, out_ns, res1_ns, res2_ns, mask_ns;
  int cin_blocks;   // Cin / 8
  int cout_blocks;  // valid 8-channel blocks of the destination
  int cout;         // real cout (NCHW store)
  int in_h, in_w;   // source spatial size
  int H, W;         // output spatial size
  int tiles_x, tiles_y;
  int mask_cb0, mask_cb1;
  int res_cb1;      // residuals apply to destination blocks [0, res_cb1)
  float slope, alpha, beta1, beta2, mask_slope;
  int accumulate;
};

__device__ __forceinline__ void glds16(const float* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int COT, int PT, bool UPS, bool NCHW_OUT>
__global__ __launch_bounds__(256) void conv3x3_f32_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TH = 4 * PT, XROW = 34, XPIX = (TH + 2) * XROW;
  constexpr int XBYTES = ((XPIX * 32 + 1023) / 1024) * 1024;
  constexpr int NXU = XBYTES / 1024, NWU = 9 * COT;
  constexpr int WBYTES = NWU * 1024, STAGE = XBYTES + WBYTES;
  constexpr int NXR = (NXU + 3) / 4, NWR = (NWU + 3) / 4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each
  // XCD a contiguous run of tiles (neighbouring tiles share halo rows and weights in L2).
  int t;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int cog = blockIdx.y;
  const int x0 = tx * 32, y0 = ty * TH;
  const int HWin = p.in_h * p.in_w;
  const float* in_n = p.in + (long long)n * p.in_ns;
  const float* wg = p.w + (size_t)cog * p.cin_blocks * (WBYTES / 4);

  // Per-lane source offsets (floats, inside one channel-block plane) of the X pieces this
  // wave moves; -1 = zero padding.
  int xoff[NXR];
#pragma unroll
  for (int r = 0; r < NXR; ++r) {
    const int u = r * 4 + wave;
    const int q = u * 64 + lane;
    const int pix = q >> 1, half = q & 1;
    const int row = pix / XROW, col = pix - row * XROW;
    const int gy = y0 - 1 + row, gx = x0 - 1 + col;
    const bool valid = (pix < XPIX) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const int sy = UPS ? (gy >> 1) : gy, sx = UPS ? (gx >> 1) : gx;
    xoff[r] = valid ? ((sy * p.in_w + sx) * 8 + half * 4) : -1;
  }

  auto stage = [&](int buf, int cb) {
    char* xs = smem + buf * STAGE;
    char* ws = xs + XBYTES;
    const float* plane = in_n + (size_t)cb * HWin * 8;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int u = r * 4 + wave;
      if (u < NXU) {
        const float* src = xoff[r] >= 0 ? plane + xoff[r] : g_zero_line;
        glds16(src, xs + u * 1024);
      }
    }
    const float* wsrc = wg + (size_t)cb * (WBYTES / 4) + lane * 4;
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
      const int u = r * 4 + wave;
      if (u < NWU) glds16(wsrc + u * 256, ws + u * 1024);
    }
  };

  f32x16 acc[COT][PT];
#pragma unroll
  for (int a = 0; a < COT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int xlane = ((wave * PT) * XROW + j) * 32 + h * 16;  // byte offset of this lane's B operand, tap (0,0), row 0
  const int wlane = j * 32 + h * 16;                         // byte offset of this lane's A operand, tap 0, cot 0

  auto compute = [&](int buf) {
    const char* xs = smem + buf * STAGE + xlane;
    const char* ws = smem + buf * STAGE + XBYTES + wlane;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int tap = dy * 3 + dx;
        f32x4 a[COT], b[PT];
#pragma unroll
        for (int c = 0; c < COT; ++c) a[c] = *(const f32x4*)(ws + (tap * COT + c) * 1024);
#pragma unroll
        for (int r = 0; r < PT; ++r) b[r] = *(const f32x4*)(xs + ((r + dy) * XROW + dx) * 32);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < PT; ++r)
              acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][s], b[r][s], acc[c][r], 0, 0, 0);
      }
    }
  };

  const int nchunk = p.cin_blocks;
  stage(0, 0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    if (c + 1 < nchunk) stage((c + 1) & 1, c + 1);
    compute(c & 1);
    __syncthreads();  // drains the LDS-DMA of chunk c+1 and frees buffer c&1
  }

  // ---- epilogue: bias, LeakyReLU, residual scale-adds, optional accumulate / LReLU-backward mask
  const int x = x0 + j;
  const long long HW = (long long)p.H * p.W;
#pragma unroll
  for (int r = 0; r < PT; ++r) {
    const int y = y0 + wave * PT + r;
    if (y >= p.H || x >= p.W) continue;
    const long long pixoff = (long long)y * p.W + x;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = (cog * COT + c) * 4 + g;
        if (cb >= p.cout_blocks) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[c][r][g * 4 + e];
        if (p.bias) {
          const f32x4 bv = *(const f32x4*)(p.bias + cb * 8 + h * 4);
          v += bv;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
        v *= p.alpha;
        const long long off = (cb * HW + pixoff) * 8 + h * 4;
        if (p.res1 && cb < p.res_cb1) v += p.beta1 * *(const f32x4*)(p.res1 + (long long)n * p.res1_ns + off);
        if (p.res2 && cb < p.res_cb1) v += p.beta2 * *(const f32x4*)(p.res2 + (long long)n * p.res2_ns + off);
        if constexpr (NCHW_OUT) {
          if (h == 0 && cb == 0) {
            float* o = p.out + (long long)n * p.out_ns + pixoff;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < p.cout) o[e * HW] = v[e];
          }
        } else {
          float* o = p.out + (long long)n * p.out_ns + off;
          if (p.accumulate) v += *(const f32x4*)o;
          if (p.mask && cb >= p.mask_cb0 && cb < p.mask_cb1) {
            const f32x4 m = *(const f32x4*)(p.mask + (long long)n * p.mask_ns + ((cb - p.mask_cb0) * HW + pixoff) * 8 + h * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : v[e] * p.mask_slope;
          }
          *(f32x4*)o = v;
        }
      }
    }
  }
}

template <int COT, int PT>
constexpr int conv_lds_bytes() {
  return 2 * ((((4 * PT + 2) * 34 * 32 + 1023) / 1024) * 1024 + 9 * COT * 1024);
}

template <int COT, int PT, bool UPS, bool NCHW_OUT>
int launch(const ConvParams& p, int n, int groups, hipStream_t stream, const sr_conv3x3_desc* d) {
  constexpr int lds = conv_lds_bytes<COT, PT>();
  static bool attr_set = false;
  auto kern = conv3x3_f32_kernel<COT, PT, UPS, NCHW_OUT>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      sr::set_error("hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
      return SR_ELAUNCH;
    }
    attr_set = true;
  }
  dim3 grid(p.tiles_x * p.tiles_y * n, groups);
  const bool prof = sr::prof_on();
  if (prof) {
    sr_launch_record r = {};
    r.kernel_id = (COT - 1) * 4 + (UPS ? 2 : 0) + (NCHW_OUT ? 1 : 0);
    r.cin = d->cin_real > 0 ? d->cin_real : d->cin_pad;
    r.cout = d->cout;
    r.n = n;
    r.h = p.H;
    r.w = p.W;
    const double px = (double)n * p.H * p.W;
    r.flops = 2.0 * 9.0 * r.cin * r.cout * px;
    const double in_px = (double)n * p.in_h * p.in_w;
    double fl = in_px * r.cin + px * r.cout;  // source once, destination once
    if (d->res1) fl += px * r.cout;
    if (d->res2) fl += px * r.cout;
    if (d->accumulate) fl += px * r.cout;
    if (d->mask_src) fl += px * d->mask_cbn * 8;
    r.bytes = 4.0 * fl;
    sr::prof_begin(stream, r);
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
  if (prof) sr::prof_end(stream);
  SR_CHECK_LAUNCH("conv3x3_f32 launch");
  return SR_OK;
}

}  // namespace

namespace sr {
// group width (couts per workgroup) rule shared with the weight packer
int conv_group_couts(int cout) {
  const int cp = (cout + 31) / 32 * 32;
  return (cp % 64 == 0) ? 64 : 32;
}
}  // namespace sr

extern "C" int sr_conv3x3_f32(const sr_conv3x3_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(d != nullptr, "sr_conv3x3_f32: null descriptor");
  SR_CHECK_ARG(d->in && d->wpacked && d->out, "sr_conv3x3_f32: null in/wpacked/out");
  SR_CHECK_ARG(d->cin_pad > 0 && d->cin_pad % 8 == 0, "sr_conv3x3_f32: cin_pad=%d must be a positive multiple of 8",
               d->cin_pad);
  SR_CHECK_ARG(d->cout > 0 && d->n > 0 && d->in_h > 0 && d->in_w > 0, "sr_conv3x3_f32: bad shape");
  SR_CHECK_ARG(!d->out_nchw || d->cout <= 4, "sr_conv3x3_f32: out_nchw needs cout <= 4 (got %d)", d->cout);
  SR_CHECK_ARG(!(d->out_nchw && (d->accumulate || d->mask_src)), "sr_conv3x3_f32: out_nchw excludes accumulate/mask");
  SR_CHECK_ARG(((uintptr_t)d->in | (uintptr_t)d->wpacked | (uintptr_t)d->out | (uintptr_t)d->res1 |
                (uintptr_t)d->res2 | (uintptr_t)d->mask_src | (uintptr_t)d->bpacked) % 16 == 0,
               "sr_conv3x3_f32: pointers must be 16-byte aligned");
  ConvParams p;
  p.in = d->in;
  p.w = d->wpacked;
  p.bias = d->bpacked;
  p.out = d->out;
  p.res1 = d->res1;
  p.res2 = d->res2;
  p.mask = d->mask_src;
  p.in_ns = d->in_img_stride;
  p.out_ns = d->out_img_stride;
  p.res1_ns = d->res1_img_stride;
  p.res2_ns = d->res2_img_stride;
  p.mask_ns = d->mask_img_stride;
  p.cin_blocks = d->cin_pad / 8;
  p.cout_blocks = (d->cout + 7) / 8;
  p.cout = d->cout;
  p.in_h = d->in_h;
  p.in_w = d->in_w;
  p.H = d->upsample ? 2 * d->in_h : d->in_h;
  p.W = d->upsample ? 2 * d->in_w : d->in_w;
  p.res_cb1 = d->res_cbn > 0 ? d->res_cbn : (1 << 30);
  p.mask_cb0 = d->mask_cb0;
  p.mask_cb1 = d->mask_cb0 + d->mask_cbn;
  p.slope = d->act_slope;
  p.alpha = d->alpha;
  p.beta1 = d->beta1;
  p.beta2 = d->beta2;
  p.mask_slope = d->mask_slope;
  p.accumulate = d->accumulate;
  SR_CHECK_ARG((long long)p.H * p.W * 8 * (long long)(p.cout_blocks > p.cin_blocks ? p.cout_blocks : p.cin_blocks) <
                   (1ll << 31),
               "sr_conv3x3_f32: image too large for 32-bit plane offsets");
  const int gc = sr::conv_group_couts(d->cout);
  const int groups = ((d->cout + 31) / 32 * 32) / gc;
  constexpr int PT = 2;
  p.tiles_x = sr::cdiv(p.W, 32);
  p.tiles_y = sr::cdiv(p.H, 4 * PT);
  SR_CHECK_ARG((long long)p.tiles_x * p.tiles_y * d->n < (1ll << 31), "sr_conv3x3_f32: grid too large");
  if (d->out_nchw) {
    if (d->upsample) {
      sr::set_error("sr_conv3x3_f32: out_nchw with upsample is not instantiated");
      return SR_EINVAL;
    }
    return launch<1, PT, false, true>(p, d->n, groups, stream, d);
  }
  if (gc == 64) {
    return d->upsample ? launch<2, PT, true, false>(p, d->n, groups, stream, d)
                       : launch<2, PT, false, false>(p, d->n, groups, stream, d);
  }
  return d->upsample ? launch<1, PT, true, false>(p, d->n, groups, stream, d)
                     : launch<1, PT, false, false>(p, d->n, groups, stream, d);
}

extern "C" const char* sr_kernel_name(int id) {
  static const char* names[8] = {
      "conv3x3_f32_kernelILi1ELi2ELb0ELb0E", "conv3x3_f32_kernelILi1ELi2ELb0ELb1E", "conv3x3_f32_kernelILi1ELi2ELb1ELb0E",
      "conv3x3_f32_kernelILi1ELi2ELb1ELb1E", "conv3x3_f32_kernelILi2ELi2ELb0ELb0E", "conv3x3_f32_kernelILi2ELi2ELb0ELb1E",
      "conv3x3_f32_kernelILi2ELi2ELb1ELb0E", "conv3x3_f32_kernelILi2ELi2ELb1ELb1E"};
  static const char* wnames[5] = {"wgrad3x3_f32_kernelILi1ELi1ELi1E", "wgrad3x3_f32_kernelILi1ELi2ELi1E",
                                  "wgrad3x3_f32_kernelILi1ELi4ELi1E", "wgrad3x3_f32_kernelILi2ELi2ELi1E",
                                  "wgrad3x3_f32_kernelILi2ELi1ELi1E"};
  if (id >= 8 && id < 13) return wnames[id - 8];
  return (id >= 0 && id < 8) ? names[id] : "";
}
