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
// reads fall on disjoint banks).  A step (R output rows) is only 0.1-0.5 us of MFMA work — shorter than the memory
// latency — so rows are prefetched NSTG-1 steps ahead into the rings and retired with a counted s_waitcnt vmcnt and a
// raw s_barrier (an LDS-DMA is a pending VM op: __syncthreads() would drain the prefetch); every wave issues exactly
// L loads per stage (surplus units read the zero line into a dump KB) so one immediate count serves all waves.
// The KS partial tiles of a pair are summed through the LDS before they leave the
// workgroup: the slab (one fp32 tile per pair per workgroup) is the dominant HBM traffic of this kernel, so the grid is
// one workgroup per CU and not more.  The slab is reduced by wgrad_f32.hip's deterministic two-stage reduction.
#include "sr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {


struct WgradParamsH {
  const void* zero;  // sr::zero_line()
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
  long long* dbg;    // development: per-wave phase clocks (sr_dev_wgrad_bf16_phase_clocks)
};

__device__ __forceinline__ void glds16wh(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// ds_read_b64_tr_b16 through inline asm: behind the builtin hipcc (ROCm 7.2) waits vmcnt(0) before the first LDS read of
// every step (it cannot tell the read from the rows still arriving by LDS-DMA), which drains the row prefetch.  The asm
// form is invisible to that analysis; its completion is awaited by tr_wait() (explicit s_waitcnt + a scheduling barrier:
// tying the fragments to the wait as in/out operands instead costs ~150 register moves per step).
template <int OFF>
__device__ __forceinline__ s16x4 tr_read(unsigned lds_addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void tr_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // nothing (the MFMAs in particular) moves above the wait
}

constexpr int PLP = 136;           // 16-byte pieces per plane row: 68 pixels (64 + halo, padded) x 2 halves
constexpr int PLB = PLP * 16;      // 2176 bytes

template <int N>
__device__ __forceinline__ void wait_vmcnt_w() {
  // gfx9 s_waitcnt simm16: vmcnt[3:0] = bits 3:0, vmcnt[5:4] = bits 15:14, expcnt 6:4 (7 = no wait), lgkmcnt 11:8 (15 = no wait)
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

template <int CT, int IT, int KS, int R, int NSTG>
struct WgradGeom {
  static constexpr int P = CT * IT, NW = P * KS;
  static constexpr int UPW = R * 4 / KS;  // 16-pixel k-steps per wave per step
  static constexpr int XUNITS = (IT * 2 * PLP + 63) / 64, YUNITS = (CT * 2 * PLP + 63) / 64;
  static constexpr int XROWB = XUNITS * 1024, YROWB = YUNITS * 1024;
  static constexpr int NXR = NSTG * R + 2, NYR = NSTG * R;
  static constexpr int XRING = NXR * XROWB, YRING = NYR * YROWB;
  static constexpr int DUMP = XRING + YRING;
  static constexpr int UNITS_PER_STAGE = R * (XUNITS + YUNITS);
  static constexpr int L = (UNITS_PER_STAGE + NW - 1) / NW;  // loads per wave per stage
  static constexpr int RED = P * (KS - 1) * 4096;
  static constexpr int LDS = (DUMP + 1024) > RED ? (DUMP + 1024) : RED;
  static_assert(NW <= 8 && R * 4 % KS == 0 && UPW >= 1, "bad wave split");
  static_assert(L * (NSTG - 2) < 64, "vmcnt is a 6-bit counter");
};

template <int CT, int IT, int KS, int R, int NSTG>
__global__ __launch_bounds__(CT * IT * KS * 64) void wgrad_bf16_kernel(const WgradParamsH p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = WgradGeom<CT, IT, KS, R, NSTG>;
  constexpr int P = G::P, NW = G::NW, UPW = G::UPW, XUNITS = G::XUNITS, YUNITS = G::YUNITS, XROWB = G::XROWB, YROWB = G::YROWB;
  constexpr int NXR = G::NXR, NYR = G::NYR, XRING = G::XRING, DUMP = G::DUMP, L = G::L;

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
  const int nsteps = (y_end - y_begin + R - 1) / R;

  const __bf16* xn = p.x + (long long)n * p.x_ns;
  const __bf16* dyn = p.dy + (long long)n * p.dy_ns;
  const long long xplane = (long long)p.x_h * p.x_w * 16, yplane = (long long)p.H * p.W * 16;
  char* xring = smem;
  char* yring = smem + XRING;
  const unsigned lds_base = (unsigned)(size_t)smem;  // LDS byte address of the dynamic segment

  // one 1 KiB unit of X tap-row `row` (source row row - 1, ring slot row % NXR) / of dY row y
  auto load_x = [&](int row, int v) {
    const int vy = row - 1;
    const int q = v * 64 + lane;
    const int plane = q / PLP, within = q - plane * PLP;
    const int px = within >> 1, half = within & 1;
    const int cb = p.cin_tile0 * 2 + plane;
    const int gx = x0 - 1 + px;
    const bool ok = plane < IT * 2 && px < 66 && vy >= 0 && vy < p.H && gx >= 0 && gx < p.W && cb < p.cin_blocks;
    const int sy = vy >> p.src_shift, sx = gx >> p.src_shift;
    const void* src = ok ? (const void*)(xn + cb * xplane + ((long long)sy * p.x_w + sx) * 16 + half * 8) : p.zero;
    glds16wh(src, xring + (row % NXR) * XROWB + v * 1024);
  };
  auto load_y = [&](int y, int v) {
    const int q = v * 64 + lane;
    const int plane = q / PLP, within = q - plane * PLP;
    const int px = within >> 1, half = within & 1;
    const int cb = p.cout_tile0 * 2 + plane;
    const int gx = x0 + px;
    // rows at or below y_end belong to the next workgroup: they arrive as zeros, so the k-steps need no row guard
    // (a guard around the MFMAs makes hipcc copy all 144 accumulator registers twice per step)
    const bool ok = plane < CT * 2 && px < 64 && y < y_end && gx < p.W && cb < p.cout_blocks;
    const void* src = ok ? (const void*)(dyn + cb * yplane + ((long long)y * p.W + gx) * 16 + half * 8) : p.zero;
    glds16wh(src, yring + (y % NYR) * YROWB + v * 1024);
  };
  // stage s = X tap-rows [y_begin + 2 + sR, +R) and dY rows [y_begin + sR, +R); exactly L loads per wave
  auto issue = [&](int s) {
    const bool real = s < nsteps;
#pragma unroll
    for (int uu = 0; uu < L; ++uu) {
      const int u = uu * NW + wave;
      if (real && u < G::UNITS_PER_STAGE) {
        const int r = u / (XUNITS + YUNITS), v = u % (XUNITS + YUNITS);
        if (v < XUNITS)
          load_x(y_begin + 2 + s * R + r, v);
        else
          load_y(y_begin + s * R + r, v - XUNITS);
      } else {
        glds16wh(p.zero, smem + DUMP);
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

  long long tk[5] = {0, 0, 0, 0, 0};
  if (p.dbg) tk[0] = __builtin_readcyclecounter();
  // prologue: the two halo tap-rows, then NSTG-1 stages in flight
  for (int u = wave; u < 2 * XUNITS; u += NW) load_x(y_begin + u / XUNITS, u % XUNITS);
#pragma unroll
  for (int s = 0; s < NSTG - 1; ++s) issue(s);
  for (int s = 0; s < nsteps; ++s) {
    long long tw = 0;
    if (p.dbg) tw = __builtin_readcyclecounter();
    wait_vmcnt_w<L*(NSTG - 2)>();    // this wave's loads of stage s (and everything older) have landed
    __builtin_amdgcn_s_barrier();     // ... and everybody's; everybody is done reading the rows of step s-1
    if (p.dbg) {
      const long long tn = __builtin_readcyclecounter();
      if (s == 0) tk[1] = tn - tk[0]; else tk[2] += tn - tw;
    }
    issue(s + NSTG - 1);              // overwrite exactly those rows
    const int y = y_begin + s * R;
#pragma unroll
    for (int uw = 0; uw < UPW; ++uw) {
      const int unit = ks * UPW + uw;
      const int row = y + unit / 4, seg = unit % 4;
      {
        // all 20 operand fragments of this 16-pixel k-step (A: dY; B: X at the 9 tap shifts), one wait, 9 MFMAs
        s16x4 f[20];
        const unsigned ya = lds_base + XRING + (row % NYR) * YROWB + a_lane + seg * 512;
        f[0] = tr_read<0>(ya);
        f[1] = tr_read<128>(ya);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const unsigned xb = lds_base + ((row + ty) % NXR) * XROWB + b_lane + seg * 512;
          f[2 + 6 * ty] = tr_read<0>(xb);
          f[3 + 6 * ty] = tr_read<128>(xb);
          f[4 + 6 * ty] = tr_read<32>(xb);
          f[5 + 6 * ty] = tr_read<160>(xb);
          f[6 + 6 * ty] = tr_read<64>(xb);
          f[7 + 6 * ty] = tr_read<192>(xb);
        }
        tr_wait();
        const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(f[0], f[1], 0, 1, 2, 3, 4, 5, 6, 7));
        if (it == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum += (float)a[e];
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const bf16x8 b = __builtin_bit_cast(bf16x8, __builtin_shufflevector(f[2 + 2 * tap], f[3 + 2 * tap], 0, 1, 2, 3, 4, 5, 6, 7));
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[tap], 0, 0, 0);
        }
      }
    }
  }
  wait_vmcnt_w<0>();  // the tail's dummy loads target the dump KB; the rings are reused below
  __syncthreads();
  if (p.dbg) tk[3] = __builtin_readcyclecounter() - tk[0];

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
  if (p.dbg && lane == 0) {
    long long* o = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
    o[0] = tk[0]; o[1] = tk[1]; o[2] = tk[2]; o[3] = tk[3]; o[4] = __builtin_readcyclecounter() - tk[0]; o[5] = nsteps;
  }
}

long long* g_wgrad_phase_clocks = nullptr;

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

template <int CT, int IT, int KS, int R, int NSTG>
int launch_group(const sr_conv3x3_wgrad_desc* d, WgradParamsH p, int cout_tile0, int cin_tile0, const SlabCarve& sc,
                 bool want_bias, hipStream_t stream) {
  using G = WgradGeom<CT, IT, KS, R, NSTG>;
  constexpr int P = G::P;
  constexpr int lds = G::LDS;
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
  p.dbg = g_wgrad_phase_clocks;
  auto kern = wgrad_bf16_kernel<CT, IT, KS, R, NSTG>;
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
    r.kernel_id = 32 + (CT == 2 ? (IT == 4 ? 5 : (IT == 2 ? 4 : 3)) : (IT == 5 ? 7 : (IT == 3 ? 6 : (IT == 4 ? 2 : (IT == 2 ? 1 : 0)))));
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
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(G::NW * 64), lds, stream, p);
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

// Development aid (tools/bf16_phase.py; not part of the ABI): per-wave phase clocks of the next launches.
extern "C" void sr_dev_wgrad_bf16_phase_clocks(void* buf) { g_wgrad_phase_clocks = (long long*)buf; }

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
  p.zero = sr::zero_line();
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
          rc = launch_group<2, 4, 1, 1, 4>(d, p, c0, i0, sc, bias, stream);
        } else if (left >= 2) {
          in = 2;
          rc = launch_group<2, 2, 2, 2, 3>(d, p, c0, i0, sc, bias, stream);
        } else {
          in = 1;
          rc = launch_group<2, 1, 4, 2, 4>(d, p, c0, i0, sc, bias, stream);
        }
      } else {
        if (left == 5 || left >= 9) {  // conv4 of a dense block: one 5-wave pass instead of 4 + 1
          in = 5;
          rc = launch_group<1, 5, 1, 1, 4>(d, p, c0, i0, sc, bias, stream);
        } else if (left >= 4) {
          in = 4;
          rc = launch_group<1, 4, 2, 1, 4>(d, p, c0, i0, sc, bias, stream);
        } else if (left == 3) {
          in = 3;
          rc = launch_group<1, 3, 2, 1, 4>(d, p, c0, i0, sc, bias, stream);
        } else if (left == 2) {
          in = 2;
          rc = launch_group<1, 2, 4, 2, 4>(d, p, c0, i0, sc, bias, stream);
        } else {
          in = 1;
          rc = launch_group<1, 1, 8, 2, 4>(d, p, c0, i0, sc, bias, stream);
        }
      }
      if (rc) return rc;
      i0 += in;
    }
    c0 += cn;
  }
  return SR_OK;
}
