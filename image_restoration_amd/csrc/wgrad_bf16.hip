// Weight gradient of the fused 3x3 convolution on bf16 MFMA (gfx950) — bf16 training path (extension, see
// layout_bf16.hip).  Same contract as wgrad_f32.hip (autograd counterpart of nn.Conv2d's weight/bias gradient,
// rrdbnet_arch.py:21-25,94-101 through esrgan_model.py:47):
//     dW[co][ci][tap] = sum_p dY[co][p] * X[ci][p + tap],   db[co] = sum_p dY[co][p],
// with X and dY in bf16 (CB16) and dW, db accumulated and written in fp32.
//
// GEMM view: D[cout][cin] (per tap) += A[cout][k] * B[k][cin], k = 16 consecutive PIXELS of a row, on
// v_mfma_f32_32x32x16_bf16.  The operands are k-major (a lane needs 8 pixels of ONE channel) while CB16 is
// channel-minor, so both come out of the pixel-major LDS images through ds_read_b64_tr_b16 (the hardware transpose
// read: a 16-lane group reads 4 pixels x 16 channels and each lane receives its channel's 4 pixels).
// One wave owns one (32-cout, 32-cin) tile pair for all 9 taps (9 x 16 fp32 accumulators); a workgroup is 8 waves =
// P pairs x KS k-splits that share the staged rows and walks DOWN a 64-pixel-wide column strip; rows arrive by LDS-DMA
// into row rings ([plane][68 px][16 ch], plane stride 2176 B = 128 mod 256 so the two channel blocks a 32-lane half
// reads fall on disjoint banks).  The KS partial tiles of a pair are summed through the LDS before they leave the
// workgroup: the slab (one fp32 tile per pair per workgroup) is the dominant HBM traffic of this kernel, so the grid is
// one workgroup per CU and not more.  The slab is reduced by wgrad_f32.hip's deterministic two-stage reduction.
#include "sr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __attribute__((aligned(64))) float g_zero_line_wh[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

struct WgradParamsH {
  const __bf16* x;   // forward source activation, CB16
  const __bf16* dy;  // gradient wrt the conv's (pre-activation) output, CB16
  float* slab;       // [workgroup][pair][9][1024]
  float* bslab;      // [workgroup][CT][32] or null
  long long x_ns, dy_ns;  // image strides in elements
  int x_h, x_w;      // source spatial size
  int H, W;          // output spatial size
  int cin_blocks;    // valid 16-channel blocks of x
  int cout_blocks;   // valid 16-channel blocks of dy
  int cin_tile0, cout_tile0;
  int strips, rows_per_wg, row_splits;
  int src_shift;     // 1: x is read through the nearest x2 upsample
};

__device__ __forceinline__ void glds16wh(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ s16x4 tr_read(const char* lds) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)lds);
}

constexpr int PLP = 136;           // 16-byte pieces per plane row: 68 pixels (64 + halo, padded) x 2 halves
constexpr int PLB = PLP * 16;      // 2176 bytes

template <int CT, int IT, int KS>
__global__ __launch_bounds__(512) void wgrad_bf16_kernel(const WgradParamsH p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int P = CT * IT;
  static_assert(P * KS == 8, "8 waves");
  constexpr int R = KS == 8 ? 2 : 1;        // output rows per step
  constexpr int UPW = R * 4 / KS;           // 16-pixel k-steps per wave per step
  constexpr int XUNITS = (IT * 2 * PLP + 63) / 64, YUNITS = (CT * 2 * PLP + 63) / 64;
  constexpr int XROWB = XUNITS * 1024, YROWB = YUNITS * 1024;
  constexpr int NXR = 2 * R + 2, NYR = 2 * R;
  constexpr int XRING = NXR * XROWB;
  constexpr int UNITS_PER_STEP = R * (XUNITS + YUNITS);
  constexpr int UW = (UNITS_PER_STEP + 7) / 8;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave / (IT * KS), it = (wave / KS) % IT, ks = wave % KS;

  int t = blockIdx.x;
  const int rs = t % p.row_splits;
  t /= p.row_splits;
  const int strip = t % p.strips;
  const int n = t / p.strips;
  const int x0 = strip * 64;
  const int y_begin = rs * p.rows_per_wg;
  const int y_end = min(y_begin + p.rows_per_wg, p.H);

  const __bf16* xn = p.x + (long long)n * p.x_ns;
  const __bf16* dyn = p.dy + (long long)n * p.dy_ns;
  const long long xplane = (long long)p.x_h * p.x_w * 16, yplane = (long long)p.H * p.W * 16;
  char* xring = smem;
  char* yring = smem + XRING;

  // Stage X tap-rows u in [ux, ux+R) (source row u - 1, ring slot u % NXR) and, with_dy, dY rows [yy, yy+R).
  auto stage = [&](int ux, int yy, bool with_dy) {
#pragma unroll
    for (int uu = 0; uu < UW; ++uu) {
      const int u = uu * 8 + wave;
      if (u >= UNITS_PER_STEP) break;
      const int r = u / (XUNITS + YUNITS), v = u % (XUNITS + YUNITS);
      if (v >= XUNITS && !with_dy) continue;
      if (v < XUNITS) {
        const int row = ux + r;
        const int vy = row - 1;
        const int q = v * 64 + lane;
        const int plane = q / PLP, within = q - plane * PLP;
        const int px = within >> 1, half = within & 1;
        const int cb = p.cin_tile0 * 2 + plane;
        const int gx = x0 - 1 + px;
        const bool ok = plane < IT * 2 && px < 66 && vy >= 0 && vy < p.H && gx >= 0 && gx < p.W && cb < p.cin_blocks;
        const int sy = vy >> p.src_shift, sx = gx >> p.src_shift;
        const void* src = ok ? (const void*)(xn + cb * xplane + ((long long)sy * p.x_w + sx) * 16 + half * 8) : (const void*)g_zero_line_wh;
        glds16wh(src, xring + (row % NXR) * XROWB + v * 1024);
      } else {
        const int vv = v - XUNITS;
        const int y = yy + r;
        const int q = vv * 64 + lane;
        const int plane = q / PLP, within = q - plane * PLP;
        const int px = within >> 1, half = within & 1;
        const int cb = p.cout_tile0 * 2 + plane;
        const int gx = x0 + px;
        const bool ok = plane < CT * 2 && px < 64 && y < p.H && gx < p.W && cb < p.cout_blocks;
        const void* src = ok ? (const void*)(dyn + cb * yplane + ((long long)y * p.W + gx) * 16 + half * 8) : (const void*)g_zero_line_wh;
        glds16wh(src, yring + (y % NYR) * YROWB + vv * 1024);
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float bsum = 0.f;

  // transposed-read lane roles: 16-lane group grp = (kh << 1) | (channel block of the 32-tile); lane 4q+pp of the group
  // addresses pixel q, channels 4pp..4pp+3 of its block, and receives channel (lane & 15) of pixels q = 0..3.
  const int grp = lane >> 4, li = lane & 15;
  const int tq = li >> 2, tp = li & 3;
  const int kh = grp >> 1, blk = grp & 1;
  const int a_lane = (ct * 2 + blk) * PLB + (kh * 8 + tq) * 32 + tp * 8;
  const int b_lane = (it * 2 + blk) * PLB + (kh * 8 + tq) * 32 + tp * 8;

  stage(y_begin, y_begin, true);
  for (int u0 = y_begin + R; u0 < y_begin + 2 + R; u0 += R) stage(u0, 0, false);
  __syncthreads();
  for (int y = y_begin; y < y_end; y += R) {
    stage(y + 2 + R, y + R, true);
#pragma unroll
    for (int uw = 0; uw < UPW; ++uw) {
      const int unit = ks * UPW + uw;
      const int row = y + unit / 4, seg = unit % 4;
      if (row < y_end) {
        const char* ya = yring + (row % NYR) * YROWB + a_lane + seg * 512;
        const s16x4 a0 = tr_read(ya), a1 = tr_read(ya + 128);
        const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        if (it == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum += (float)a[e];
        }
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const char* xb = xring + ((row + ty) % NXR) * XROWB + b_lane + seg * 512;
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const s16x4 b0 = tr_read(xb + tx * 32), b1 = tr_read(xb + tx * 32 + 128);
            const bf16x8 b = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ty * 3 + tx], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  // sum the KS partial tiles of each pair through the LDS (the rings are dead: every load was drained by the last
  // barrier), tap by tap: waves ks > 0 park their tile, wave ks = 0 adds them in fixed order.
  if constexpr (KS > 1) {
    float* red = (float*)smem;  // [pair][ks-1][1024] floats per round (<= 28 KB)
    const int pair_w = ct * IT + it;
#pragma unroll
    for (int tap = 0; tap < 10; ++tap) {
      if (ks > 0) {
        float* dst = red + ((pair_w * (KS - 1) + (ks - 1)) * 1024) + lane * 4;
        if (tap < 9) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[tap < 9 ? tap : 0][g * 4 + e];
            *(f32x4*)(dst + g * 256) = v;
          }
        } else {
          dst[0] = bsum;
        }
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int k = 0; k < KS - 1; ++k) {
          const float* src = red + ((pair_w * (KS - 1) + k) * 1024) + lane * 4;
          if (tap < 9) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 v = *(const f32x4*)(src + g * 256);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[tap < 9 ? tap : 0][g * 4 + e] += v[e];
            }
          } else {
            bsum += src[0];
          }
        }
      }
      __syncthreads();
    }
  }

  if (ks == 0) {
    // this pair's tile:  slab[workgroup][pair][tap][g][lane][4]
    const int pair = ct * IT + it;
    float* dst = p.slab + (((long long)blockIdx.x * P + pair) * 9) * 1024 + lane * 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[tap][g * 4 + e];
        *(f32x4*)(dst + tap * 1024 + g * 256) = v;
      }
    if (p.bslab && it == 0) {
      // lane (blk, li, kh) summed cout blk*16 + li over its kh half of the pixels
      bsum += __shfl_xor(bsum, 32);
      if (kh == 0) p.bslab[((long long)blockIdx.x * CT + ct) * 32 + blk * 16 + li] = bsum;
    }
  }
}

template <int CT, int IT, int KS>
constexpr int wgrad_bf16_lds() {
  constexpr int R = KS == 8 ? 2 : 1;
  constexpr int XU = (IT * 2 * PLP + 63) / 64, YU = (CT * 2 * PLP + 63) / 64;
  constexpr int ring = (2 * R + 2) * XU * 1024 + 2 * R * YU * 1024;
  constexpr int red = CT * IT * (KS - 1) * 4096;
  return ring > red ? ring : red;
}

struct SlabCarve {
  float *slab, *bslab, *part, *bpart;
  size_t wslab_bytes;
};
constexpr size_t kPartFloats = (size_t)64 * 8 * 9 * 1024;  // stage-1 partials: 64 chunks of a P = 8 launch
constexpr size_t kBpartFloats = 64 * 64;

size_t max_wgs(int n, int w) { return 256 + (size_t)n * ((w + 63) / 64); }

SlabCarve carve_slab(void* base, size_t bytes) {
  SlabCarve c;
  const size_t tail = (kPartFloats + kBpartFloats) * sizeof(float) + 4096;
  const size_t head = bytes > tail ? bytes - tail : 0;
  c.wslab_bytes = head / 65 * 64 / 256 * 256;  // 1/65 of the head is bias partials
  c.slab = (float*)base;
  c.bslab = (float*)((char*)base + c.wslab_bytes);
  c.part = (float*)((char*)base + (head / 256 * 256));
  c.bpart = c.part + kPartFloats;
  return c;
}

template <int CT, int IT, int KS>
int launch_group(const sr_conv3x3_wgrad_desc* d, WgradParamsH p, int cout_tile0, int cin_tile0, const SlabCarve& sc,
                 bool want_bias, hipStream_t stream) {
  constexpr int P = CT * IT, R = KS == 8 ? 2 : 1;
  constexpr int lds = wgrad_bf16_lds<CT, IT, KS>();
  static_assert(lds <= 160 * 1024, "rings do not fit the LDS");
  p.cin_tile0 = cin_tile0;
  p.cout_tile0 = cout_tile0;
  // one workgroup per CU: every workgroup leaves P fp32 tiles of 36 KB, so more workgroups = more slab traffic
  const long long strips_total = (long long)d->n * p.strips;
  const int want = (int)(256 / strips_total) > 1 ? (int)(256 / strips_total) : 1;
  int rows = sr::cdiv(p.H, want);
  rows = (rows + R - 1) / R * R;
  p.rows_per_wg = rows;
  p.row_splits = sr::cdiv(p.H, rows);
  const long long nwg = strips_total * p.row_splits;
  if ((size_t)nwg * P * 9 * 1024 * sizeof(float) > sc.wslab_bytes || (size_t)nwg * CT * 32 * sizeof(float) > sc.wslab_bytes / 64) {
    sr::set_error("sr_conv3x3_wgrad_bf16: slab too small (need %zu B of tiles)", (size_t)nwg * P * 9 * 1024 * sizeof(float));
    return SR_ENOSPACE;
  }
  p.slab = sc.slab;
  p.bslab = want_bias ? sc.bslab : nullptr;
  auto kern = wgrad_bf16_kernel<CT, IT, KS>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      sr::set_error("wgrad_bf16: hipFuncSetAttribute(%d) failed", lds);
      return SR_ELAUNCH;
    }
    attr_set = true;
  }
  const bool prof = sr::prof_on();
  if (prof) {
    sr_launch_record r = {};
    r.kernel_id = 32 + (CT == 2 ? (IT == 4 ? 5 : (IT == 2 ? 4 : 3)) : (IT == 4 ? 2 : (IT == 2 ? 1 : 0)));
    r.cin = 32 * IT;
    r.cout = 32 * CT;
    r.n = d->n;
    r.h = p.H;
    r.w = p.W;
    const double px = (double)d->n * p.H * p.W;
    const int cin_eff = min(32 * IT, d->cin_pad - 32 * cin_tile0), cout_eff = min(32 * CT, d->cout - 32 * cout_tile0);
    r.flops = 2.0 * 9 * cin_eff * cout_eff * px;
    r.bytes = 2.0 * px * (cin_eff + cout_eff) + 2.0 * nwg * P * 9 * 4096;
    sr::prof_begin(stream, r);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(512), lds, stream, p);
  if (prof) sr::prof_end(stream);
  SR_CHECK_LAUNCH("wgrad_bf16 launch");
  sr::WgradReduce rr = {};
  rr.slab = sc.slab;
  rr.bslab = want_bias ? sc.bslab : nullptr;
  rr.part = sc.part;
  rr.bpart = sc.bpart;
  rr.splits = nwg;
  rr.P = P;
  rr.IT = IT;
  rr.CT = CT;
  rr.ntap = 9;
  rr.ks = 3;
  rr.kdim = 3;
  rr.t_mul = 1;
  rr.cin_tile0 = cin_tile0;
  rr.cout_tile0 = cout_tile0;
  rr.cout = d->cout;
  rr.cin = d->cin;
  rr.first_seg = d->first_seg;
  rr.seg = d->seg;
  rr.seg_pad = 16;
  rr.scale = d->scale;
  rr.accumulate = d->accumulate;
  rr.dw = d->dweight;
  rr.db = want_bias ? d->dbias : nullptr;
  return sr::wgrad_reduce(rr, stream);
}

}  // namespace

extern "C" size_t sr_conv3x3_wgrad_slab_bytes_bf16(int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  const size_t wbytes = max_wgs(n, w) * 8 * 9 * 1024 * sizeof(float);
  return (wbytes + wbytes / 64 + (kPartFloats + kBpartFloats) * sizeof(float) + 3 * 4096) / 256 * 256;
}

extern "C" int sr_conv3x3_wgrad_bf16(const sr_conv3x3_wgrad_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(d && d->x && d->dy && d->dweight && d->slab, "sr_conv3x3_wgrad_bf16: null argument");
  SR_CHECK_ARG(d->cout > 0 && d->cin > 0 && d->n > 0 && d->in_h > 0 && d->in_w > 0, "sr_conv3x3_wgrad_bf16: bad shape");
  SR_CHECK_ARG(((uintptr_t)d->x | (uintptr_t)d->dy | (uintptr_t)d->slab) % 16 == 0,
               "sr_conv3x3_wgrad_bf16: pointers must be 16-byte aligned");
  const int cin_pad = sr_conv3x3_cin_pad16(d->cin, d->first_seg, d->seg);
  SR_CHECK_ARG(cin_pad > 0 && cin_pad == d->cin_pad, "sr_conv3x3_wgrad_bf16: cin_pad=%d does not match cin=%d/%d/%d", d->cin_pad,
               d->cin, d->first_seg, d->seg);
  WgradParamsH p = {};
  p.x = (const __bf16*)d->x;
  p.dy = (const __bf16*)d->dy;
  p.x_ns = d->x_img_stride;
  p.dy_ns = d->dy_img_stride;
  p.x_h = d->in_h;
  p.x_w = d->in_w;
  p.H = d->upsample ? 2 * d->in_h : d->in_h;
  p.W = d->upsample ? 2 * d->in_w : d->in_w;
  p.src_shift = d->upsample ? 1 : 0;
  p.cin_blocks = cin_pad / 16;
  p.cout_blocks = (d->cout + 15) / 16;
  p.strips = sr::cdiv(p.W, 64);
  const SlabCarve sc = carve_slab(d->slab, d->slab_bytes);
  SR_CHECK_ARG(sc.wslab_bytes > 0, "sr_conv3x3_wgrad_bf16: slab too small");
  // walk the (cout tile, cin tile) grid in workgroup-sized groups
  const int cts = sr::cdiv(d->cout, 32), its = sr::cdiv(cin_pad, 32);
  for (int c0 = 0; c0 < cts;) {
    const int cn = (cts - c0 >= 2) ? 2 : 1;
    for (int i0 = 0; i0 < its;) {
      const int left = its - i0;
      const bool bias = d->dbias != nullptr && i0 == 0;
      int rc, in;
      if (cn == 2) {
        if (left >= 4) {
          in = 4;
          rc = launch_group<2, 4, 1>(d, p, c0, i0, sc, bias, stream);
        } else if (left >= 2) {
          in = 2;
          rc = launch_group<2, 2, 2>(d, p, c0, i0, sc, bias, stream);
        } else {
          in = 1;
          rc = launch_group<2, 1, 4>(d, p, c0, i0, sc, bias, stream);
        }
      } else {
        if (left >= 4) {
          in = 4;
          rc = launch_group<1, 4, 2>(d, p, c0, i0, sc, bias, stream);
        } else if (left >= 2) {
          in = 2;
          rc = launch_group<1, 2, 4>(d, p, c0, i0, sc, bias, stream);
        } else {
          in = 1;
          rc = launch_group<1, 1, 8>(d, p, c0, i0, sc, bias, stream);
        }
      }
      if (rc) return rc;
      i0 += in;
    }
    c0 += cn;
  }
  return SR_OK;
}
