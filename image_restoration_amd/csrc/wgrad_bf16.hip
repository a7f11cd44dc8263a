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
  int gi;            // cin groups per cout-group row: blockIdx.y = (row, column) of a grid of same-shaped tile groups
  int strips, rows_per_wg, row_splits;
  int n, imgs_per_wg;  // a workgroup walks imgs_per_wg images (small images: the 36 KB tiles it leaves are its fixed cost)
  int src_shift;     // 1: x is read through the nearest x2 upsample
  long long* dbg;    // development: per-wave phase clocks (sr_dev_wgrad_bf16_phase_clocks)
};

__device__ __forceinline__ void glds16wh(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// LDS-DMA through a buffer descriptor (scalar base, 32-bit lane offset, scalar row offset): see conv_f32.hip.  A descriptor with
// num_records = 0 delivers zeros without touching memory: rows outside the image and the padding loads of the counted waits.
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, char* lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
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

// NSEG: 16-pixel k-steps of a strip row that can hold pixels (4 = the 64-pixel strip; 2 / 1 for images at most 32 / 16 pixels wide —
// the deep layers of the VGG discriminator at 32x32 ... 4x4 — where the others would multiply zeros).
template <int CT, int IT, int KS, int R, int NSTG, int NSEG = 4>
struct WgradGeom {
  static constexpr int P = CT * IT, NW = P * KS;
  static constexpr int UPW = R * NSEG / KS;  // 16-pixel k-steps per wave per step
  static constexpr int XUNITS = (IT * 2 * PLP + 63) / 64, YUNITS = (CT * 2 * PLP + 63) / 64;
  static constexpr int XROWB = XUNITS * 1024, YROWB = YUNITS * 1024;
  static constexpr int NXR = NSTG * R + 2, NYR = NSTG * R;
  static constexpr int XRING = NXR * XROWB, YRING = NYR * YROWB;
  static constexpr int DUMP = XRING + YRING;
  static constexpr int UNITS_PER_STAGE = R * (XUNITS + YUNITS);
  static constexpr int L = (UNITS_PER_STAGE + NW - 1) / NW;  // loads per wave per stage
  static constexpr int RED = P * (KS - 1) * 4096;
  static constexpr int LDS = (DUMP + 1024) > RED ? (DUMP + 1024) : RED;
  static_assert(NW <= 8 && R * NSEG % KS == 0 && UPW >= 1, "bad wave split");
  static_assert(L * (NSTG - 2) < 64, "vmcnt is a 6-bit counter");
};

template <int CT, int IT, int KS, int R, int NSTG, int NSEG = 4>
__global__ __launch_bounds__(CT * IT * KS * 64) void wgrad_bf16_kernel(const WgradParamsH p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = WgradGeom<CT, IT, KS, R, NSTG, NSEG>;
  constexpr int P = G::P, NW = G::NW, UPW = G::UPW, XUNITS = G::XUNITS, YUNITS = G::YUNITS, XROWB = G::XROWB, YROWB = G::YROWB;
  constexpr int NXR = G::NXR, NYR = G::NYR, XRING = G::XRING, DUMP = G::DUMP, L = G::L;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave / (IT * KS), it = (wave / KS) % IT, ks = wave % KS;
  const int grow = blockIdx.y / p.gi, gcol = blockIdx.y - grow * p.gi;  // several tile groups per launch (grid.y)
  const int cout_tile0 = p.cout_tile0 + grow * CT, cin_tile0 = p.cin_tile0 + gcol * IT;

  int t = blockIdx.x;
  const int rs = t % p.row_splits;
  t /= p.row_splits;
  const int strip = t % p.strips;
  const int n_first = (t / p.strips) * p.imgs_per_wg, n_last = min(n_first + p.imgs_per_wg, p.n);
  const int x0 = strip * 64;
  const int y_begin = rs * p.rows_per_wg;
  const int y_end = min(y_begin + p.rows_per_wg, p.H);
  const int nsteps = (y_end - y_begin + R - 1) / R;

  const __bf16* xn = p.x + (long long)n_first * p.x_ns;  // (the image under the loop below)
  const __bf16* dyn = p.dy + (long long)n_first * p.dy_ns;
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
    const int cb = cin_tile0 * 2 + plane;
    const int gx = x0 - 1 + px;
    const bool ok = plane < IT * 2 && px < 66 && vy >= 0 && vy < p.H && gx >= 0 && gx < p.W && cb < p.cin_blocks;
    const int sy = vy >> p.src_shift, sx = gx >> p.src_shift;
    const void* src = ok ? (const void*)(xn + cb * xplane + ((long long)sy * p.x_w + sx) * 16 + half * 8) : p.zero;
    glds16wh(src, xring + (row % NXR) * XROWB + v * 1024);
  };
  // stage s = X tap-rows [y_begin + 2 + sR, +R) and dY rows [y_begin + sR, +R); exactly L loads per wave.  Which unit of which
  // row a wave's uu-th load moves never changes, so its per-lane byte offset (channel block, column, half; beyond the buffer for
  // lanes outside the image) is computed once; a step only adds the row as the scalar offset of the buffer load.
  __amdgpu_buffer_rsrc_t x_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (unsigned)((long long)p.cin_blocks * xplane * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t y_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)dyn, 0, (unsigned)((long long)p.cout_blocks * yplane * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t null_rs = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, 0, 0x00020000);
  unsigned lane_off[L];
#pragma unroll
  for (int uu = 0; uu < L; ++uu) {
    const int u = uu * NW + wave;
    const int v = u % (XUNITS + YUNITS);
    const bool is_x = v < XUNITS;
    const int q = (is_x ? v : v - XUNITS) * 64 + lane;
    const int plane = q / PLP, within = q - plane * PLP;
    const int px = within >> 1, half = within & 1;
    unsigned off = 0xfffffff0u;
    if (is_x) {
      const int cb = cin_tile0 * 2 + plane, gx = x0 - 1 + px;
      if (plane < IT * 2 && px < 66 && gx >= 0 && gx < p.W && cb < p.cin_blocks)
        off = (unsigned)((cb * xplane + (long long)(gx >> p.src_shift) * 16 + half * 8) * 2);
    } else {
      const int cb = cout_tile0 * 2 + plane, gx = x0 + px;
      if (plane < CT * 2 && px < 64 && gx < p.W && cb < p.cout_blocks) off = (unsigned)((cb * yplane + (long long)gx * 16 + half * 8) * 2);
    }
    lane_off[uu] = off;
  }
  auto issue = [&](int s) {
    const bool real = s < nsteps;
#pragma unroll
    for (int uu = 0; uu < L; ++uu) {
      const int u = uu * NW + wave;
      if (real && u < G::UNITS_PER_STAGE) {
        const int r = u / (XUNITS + YUNITS), v = u % (XUNITS + YUNITS);
        if (v < XUNITS) {
          const int row = y_begin + 2 + s * R + r, vy = row - 1;
          const bool ok = vy >= 0 && vy < p.H;
          blds16(ok ? x_rs : null_rs, lane_off[uu], ok ? (unsigned)(vy >> p.src_shift) * (unsigned)p.x_w * 32u : 0u,
                 xring + (row % NXR) * XROWB + v * 1024);
        } else {
          // rows at or below y_end belong to the next workgroup: they arrive as zeros, so the k-steps need no row guard
          const int y = y_begin + s * R + r;
          const bool ok = y < y_end;
          blds16(ok ? y_rs : null_rs, lane_off[uu], ok ? (unsigned)y * (unsigned)p.W * 32u : 0u, yring + (y % NYR) * YROWB + (v - XUNITS) * 1024);
        }
      } else {
        blds16(null_rs, 0u, 0u, smem + DUMP);
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
  for (int n = n_first; n < n_last; ++n) {
  if (n > n_first) {  // the next image of this workgroup: the rings are free (every load drained, every read behind the barrier)
    wait_vmcnt_w<0>();
    __syncthreads();
    xn = p.x + (long long)n * p.x_ns;
    dyn = p.dy + (long long)n * p.dy_ns;
    x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (unsigned)((long long)p.cin_blocks * xplane * 2), 0x00020000);
    y_rs = __builtin_amdgcn_make_buffer_rsrc((void*)dyn, 0, (unsigned)((long long)p.cout_blocks * yplane * 2), 0x00020000);
  }
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
      const int row = y + unit / NSEG, seg = unit % NSEG;
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
  }  // images of this workgroup
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
#pragma unroll 1  // (rolled: unrolled sevenfold for KS = 8 the loads of all partial tiles were live at once and the kernel spilled)
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
    float* dst = p.slab + ((((long long)blockIdx.y * gridDim.x + blockIdx.x) * P + pair) * 9) * 1024 + lane * 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[tap][g * 4 + e];
        *(f32x4*)(dst + tap * 1024 + g * 256) = v;
      }
    if (p.bslab && it == 0 && gcol == 0) {
      // lane (blk, li, kh) summed cout blk*16 + li over its kh half of the pixels
      bsum += __shfl_xor(bsum, 32);
      if (kh == 0) p.bslab[(((long long)grow * gridDim.x + blockIdx.x) * CT + ct) * 32 + blk * 16 + li] = bsum;
    }
  }
  if (p.dbg && lane == 0) {
    long long* o = p.dbg + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8;
    o[0] = tk[0]; o[1] = tk[1]; o[2] = tk[2]; o[3] = tk[3]; o[4] = __builtin_readcyclecounter() - tk[0]; o[5] = nsteps;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// All five weight gradients of one residual dense block in ONE launch.  At bf16 rates these kernels run at the HBM
// share of a CU (~25 GB/s), so what matters is bytes: a workgroup's 8 waves each own one (cout tile, cin tile) pair for
// all taps (no k-split), and the block's 26 pairs (nf = 64, gc = 32) are packed into 4 ITEMS of <= 8 pairs that share
// their rows — {conv5 tile0 x X0-5, conv1 x X0-1}, {conv5 tile1 x X0-5}, {conv4 x X0-4, conv2 x X0-2}, {conv3 x X0-3} —
// so X is read 4 times and dY once per item instead of once per (conv, group) launch, the slab shrinks (one tile per
// pair per workgroup, 64 pixel-splits) and items x splits = 256 workgroups = one round of the chip.  X is the block's
// concat buffer for every item, dY one or two 32-channel windows of the gradient concat buffer.
struct RdbItem {
  int ndy;                 // 32-cout tiles of dY held by the item (1 or 2)
  int dy_cb0_0, dy_cb0_1, dy_cbn_0, dy_cbn_1;  // first 16-channel block in D and valid blocks (<= 2) of each
  int x_tile0, nx;         // cin tiles [x_tile0, x_tile0 + nx) of the concat buffer, nx <= 6
  int npairs;              // <= 8: wave w owns pair w = (dY tile pa(w), X tile pb(w) relative to x_tile0)
  unsigned pa_bits, pb_nib;  // pa(w) = bit w of pa_bits, pb(w) = nibble w of pb_nib (no arrays: a dynamically indexed
                             // kernel-argument array is copied to scratch)
  int bias_wave_0, bias_wave_1;  // wave that accumulates the bias gradient of dY tile a (or -1)
  long long slab_off, bslab_off;  // float offsets of this item's tiles / bias partials
};
struct RdbWgradParams {
  const void* zero;
  const __bf16* x;
  const __bf16* dy;
  float* slab;
  float* bslab;
  long long x_ns, dy_ns;
  int H, W, cin_blocks, strips, rows_per_wg, row_splits, nitems, splits;
  int xcd_map;     // 1: the items of a pixel split sit on ONE XCD (below)
  long long* dbg;  // development: per-wave phase clocks
};
struct RdbItems {
  RdbItem it[8];
};

constexpr int RDB_NSTG = 3, RDB_MAXX = 6;
constexpr int RDB_XUNITS = (RDB_MAXX * 2 * PLP + 63) / 64, RDB_YUNITS = (2 * 2 * PLP + 63) / 64;
constexpr int RDB_XROWB = RDB_XUNITS * 1024, RDB_YROWB = RDB_YUNITS * 1024;
constexpr int RDB_NXR = RDB_NSTG + 2, RDB_NYR = RDB_NSTG;
constexpr int RDB_XRING = RDB_NXR * RDB_XROWB, RDB_DUMP = RDB_XRING + RDB_NYR * RDB_YROWB;
constexpr int RDB_L = (RDB_XUNITS + RDB_YUNITS + 7) / 8;
constexpr int RDB_LDS = RDB_DUMP + 1024;
static_assert(RDB_LDS <= 160 * 1024 && RDB_L * (RDB_NSTG - 2) < 64, "rdb wgrad geometry");

// The item table travels as eight separate by-value arguments: an array inside the parameter struct makes hipcc copy
// the whole struct to scratch and re-load fields inside the loop behind vmcnt(0) waits (which drain the row prefetch).
// NSEG: 16-pixel k-steps of a row that hold pixels (4 = a 64-pixel strip; 2 for images at most 32 pixels wide — the reference recipe's
// 32x32 patches, train_ESRGAN_x4.yml:24 — where the other two would multiply zeros: half of the launch's MFMAs and operand reads).
template <int NSEG>
__global__ __launch_bounds__(512) void wgrad_rdb_bf16_kernel(const RdbWgradParams p, const RdbItem i0, const RdbItem i1,
                                                             const RdbItem i2, const RdbItem i3, const RdbItem i4,
                                                             const RdbItem i5, const RdbItem i6, const RdbItem i7) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NSTG = RDB_NSTG, NXR = RDB_NXR, NYR = RDB_NYR, XROWB = RDB_XROWB, YROWB = RDB_YROWB;
  constexpr int XRING = RDB_XRING, DUMP = RDB_DUMP, L = RDB_L;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The items of one pixel split read the same rows of X at the same pace.  Workgroups are dealt round-robin over the 8 XCDs, each
  // with its own L2: with the items as the fastest index the siblings land on different XCDs and every item fetches its rows from
  // HBM for itself (measured: HBM traffic = the staged bytes).  Here workgroup b runs on XCD b & 7 and the items of a split are 8
  // ids apart: same XCD, dispatched back to back, so the first sibling's fetch serves the others out of that L2.
  int split_i, item_i;
  if (p.xcd_map) {
    const int k = blockIdx.x >> 3;
    item_i = k % p.nitems;
    split_i = (k / p.nitems) * 8 + (blockIdx.x & 7);
  } else {
    split_i = blockIdx.x / p.nitems;
    item_i = blockIdx.x - split_i * p.nitems;
  }
#define RDB_SEL(f) \
  (item_i == 0 ? i0.f : item_i == 1 ? i1.f : item_i == 2 ? i2.f : item_i == 3 ? i3.f : item_i == 4 ? i4.f : item_i == 5 ? i5.f : item_i == 6 ? i6.f : i7.f)
  struct {
    int ndy, dy_cb0_0, dy_cb0_1, dy_cbn_0, dy_cbn_1, x_tile0, nx, npairs;
    long long slab_off, bslab_off;
  } im;
  im.ndy = RDB_SEL(ndy);
  im.dy_cb0_0 = RDB_SEL(dy_cb0_0);
  im.dy_cb0_1 = RDB_SEL(dy_cb0_1);
  im.dy_cbn_0 = RDB_SEL(dy_cbn_0);
  im.dy_cbn_1 = RDB_SEL(dy_cbn_1);
  im.x_tile0 = RDB_SEL(x_tile0);
  im.nx = RDB_SEL(nx);
  im.npairs = RDB_SEL(npairs);
  im.slab_off = RDB_SEL(slab_off);
  im.bslab_off = RDB_SEL(bslab_off);
  const unsigned pa_bits = RDB_SEL(pa_bits), pb_nib = RDB_SEL(pb_nib);
  const int bw0 = RDB_SEL(bias_wave_0), bw1 = RDB_SEL(bias_wave_1);
#undef RDB_SEL
  const bool active = wave < im.npairs;
  const int pa = active ? (pa_bits >> wave) & 1 : 0, pb = active ? (pb_nib >> (4 * wave)) & 15 : 0;
  const bool bias_wave = active && (pa ? bw1 : bw0) == wave;
  const int xunits = (im.nx * 2 * PLP + 63) / 64, yunits = (im.ndy * 2 * PLP + 63) / 64;

  int t = split_i;
  const int rs = t % p.row_splits;
  t /= p.row_splits;
  const int strip = t % p.strips;
  const int n = t / p.strips;
  const int x0 = strip * 64;
  const int y_begin = rs * p.rows_per_wg;
  const int y_end = min(y_begin + p.rows_per_wg, p.H);
  const int nsteps = y_end - y_begin;

  const __bf16* xn = p.x + (long long)n * p.x_ns;
  const __bf16* dyn = p.dy + (long long)n * p.dy_ns;
  const long long plane_e = (long long)p.H * p.W * 16;
  char* xring = smem;
  char* yring = smem + XRING;
  const unsigned lds_base = (unsigned)(size_t)smem;

  auto load_x = [&](int row, int v) __attribute__((always_inline)) {
    const int vy = row - 1;
    const int q = v * 64 + lane;
    const int plane = q / PLP, within = q - plane * PLP;
    const int px = within >> 1, half = within & 1;
    const int cb = im.x_tile0 * 2 + plane;
    const int gx = x0 - 1 + px;
    const bool ok = plane < im.nx * 2 && px < 66 && vy >= 0 && vy < p.H && gx >= 0 && gx < p.W && cb < p.cin_blocks;
    const void* src = ok ? (const void*)(xn + cb * plane_e + ((long long)vy * p.W + gx) * 16 + half * 8) : p.zero;
    glds16wh(src, xring + (row % NXR) * XROWB + v * 1024);
  };
  // stage s = X tap-row y_begin + 2 + s and dY row y_begin + s; exactly L loads per wave (load uu of the stage).  The unit a
  // wave's uu-th load moves never changes: its per-lane byte offset is computed once, a step adds the row as the scalar offset
  // of a buffer load (conv_f32.hip); a descriptor with num_records = 0 serves rows outside the image and the padding loads.
  const __amdgpu_buffer_rsrc_t x_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (unsigned)((long long)p.cin_blocks * plane_e * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc((void*)dyn, 0, 0x7ffffff0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t null_rs = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, 0, 0x00020000);
  unsigned lane_off[L];
#pragma unroll
  for (int uu = 0; uu < L; ++uu) {
    const int u = uu * 8 + wave;
    const bool is_x = u < xunits;
    const int q = (is_x ? u : u - xunits) * 64 + lane;
    const int plane = q / PLP, within = q - plane * PLP;
    const int px = within >> 1, half = within & 1;
    unsigned off = 0xfffffff0u;
    if (is_x) {
      const int cb = im.x_tile0 * 2 + plane, gx = x0 - 1 + px;
      if (plane < im.nx * 2 && px < 66 && gx >= 0 && gx < p.W && cb < p.cin_blocks) off = (unsigned)((cb * plane_e + (long long)gx * 16 + half * 8) * 2);
    } else if (u < xunits + yunits) {
      const int a = plane >> 1, sub = plane & 1, gx = x0 + px;
      const int cbn = a ? im.dy_cbn_1 : im.dy_cbn_0, cb0 = a ? im.dy_cb0_1 : im.dy_cb0_0;
      if (a < im.ndy && sub < cbn && px < 64 && gx < p.W) off = (unsigned)(((cb0 + sub) * plane_e + (long long)gx * 16 + half * 8) * 2);
    }
    lane_off[uu] = off;
  }
  auto issue_one = [&](int s, int uu) __attribute__((always_inline)) {
    const bool real = s < nsteps;
    const int u = uu * 8 + wave;
    if (real && u < xunits) {
      const int row = y_begin + 2 + s, vy = row - 1;
      const bool ok = vy >= 0 && vy < p.H;
      blds16(ok ? x_rs : null_rs, lane_off[uu], ok ? (unsigned)vy * (unsigned)p.W * 32u : 0u, xring + (row % NXR) * XROWB + u * 1024);
    } else if (real && u < xunits + yunits) {
      const int y = y_begin + s;
      const bool ok = y < y_end;
      blds16(ok ? y_rs : null_rs, lane_off[uu], ok ? (unsigned)y * (unsigned)p.W * 32u : 0u, yring + (y % NYR) * YROWB + (u - xunits) * 1024);
    } else {
      blds16(null_rs, 0u, 0u, smem + DUMP);
    }
  };
  auto issue = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int uu = 0; uu < L; ++uu) issue_one(s, uu);
  };

  f32x16 acc[9];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float bsum = 0.f;

  const int grp = lane >> 4, li = lane & 15;
  const int tq = li >> 2, tp = li & 3;
  const int kh = grp >> 1, blk = grp & 1;
  const int a_lane = (pa * 2 + blk) * PLB + (kh * 8 + tq) * 32 + tp * 8;
  const int b_lane = (pb * 2 + blk) * PLB + (kh * 8 + tq) * 32 + tp * 8;

  long long tk[5] = {0, 0, 0, 0, 0};
  if (p.dbg) tk[0] = __builtin_readcyclecounter();
  for (int u = wave; u < 2 * xunits; u += 8) load_x(y_begin + u / xunits, u % xunits);
#pragma unroll
  for (int s = 0; s < NSTG - 1; ++s) issue(s);
  for (int s = 0; s < nsteps; ++s) {
    long long tw = 0;
    if (p.dbg) tw = __builtin_readcyclecounter();
    wait_vmcnt_w<L*(NSTG - 2)>();
    __builtin_amdgcn_s_barrier();
    long long ti = 0;
    if (p.dbg) {
      ti = __builtin_readcyclecounter();
      if (s == 0) tk[1] = ti - tk[0]; else tk[2] += ti - tw;
    }
    // The refill of the rows freed by this barrier is NOT issued here in one go: the chip-wide HBM share of a CU is
    // what these loads wait for (they block at issue once the queue is full), and a block of loads right after the
    // barrier stalls all eight waves at once while no MFMA runs.  They are dealt out between the k-steps below, so a
    // wave blocked on a load leaves the SIMD to the other wave's MFMAs.
    if (!active) issue(s + NSTG - 1);
    if (active) {
      // Software pipeline over (16-pixel k-step, tap row): the 6 X fragments of tap row ty+1 (and, behind the last tap
      // row, the next k-step's dY fragments and first tap row) are in flight while the 3 MFMAs of tap row ty run, retired
      // by counted lgkmcnt waits (LDS reads complete in order).  One wave has ~100-300 cycles of MFMA work per wait and
      // the LDS latency under eight reading waves is longer, so the second wave of the SIMD fills the rest.
      const int row = y_begin + s;
      const unsigned ya = lds_base + XRING + (row % NYR) * YROWB + a_lane;
      unsigned xb[3];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) xb[ty] = lds_base + ((row + ty) % NXR) * XROWB + b_lane;
      s16x4 fa[2][2], fb[2][6];
      auto read_a = [&](int seg, s16x4 (&d)[2]) __attribute__((always_inline)) {
        d[0] = tr_read<0>(ya + seg * 512);
        d[1] = tr_read<128>(ya + seg * 512);
      };
      auto read_b = [&](int seg, int ty, s16x4 (&d)[6]) __attribute__((always_inline)) {
        const unsigned b = xb[ty] + seg * 512;
        d[0] = tr_read<0>(b);
        d[1] = tr_read<128>(b);
        d[2] = tr_read<32>(b);
        d[3] = tr_read<160>(b);
        d[4] = tr_read<64>(b);
        d[5] = tr_read<192>(b);
      };
      read_a(0, fa[0]);
      read_b(0, 0, fb[0]);
#pragma unroll
      for (int seg = 0; seg < NSEG; ++seg) {
        // (no clock reads in here: s_memtime is a scalar-memory operation, and with one possibly outstanding the compiler drains
        // lgkmcnt to 0 around it — the read pipeline with it)
#pragma unroll
        for (int uu = seg; uu < L; uu += NSEG) issue_one(s + NSTG - 1, uu);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const int cur = (seg * 3 + ty) & 1;
          // prefetch the next group into the other buffer, then wait for the current one
          if (ty < 2) {
            read_b(seg, ty + 1, fb[cur ^ 1]);
            asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
          } else if (seg < NSEG - 1) {
            read_a(seg + 1, fa[(seg + 1) & 1]);
            read_b(seg + 1, 0, fb[cur ^ 1]);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 a = __builtin_bit_cast(bf16x8, __builtin_shufflevector(fa[seg & 1][0], fa[seg & 1][1], 0, 1, 2, 3, 4, 5, 6, 7));
          if (ty == 0 && bias_wave) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum += (float)a[e];
          }
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            const bf16x8 b = __builtin_bit_cast(bf16x8, __builtin_shufflevector(fb[cur][2 * tx], fb[cur][2 * tx + 1], 0, 1, 2, 3, 4, 5, 6, 7));
            acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ty * 3 + tx], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  wait_vmcnt_w<0>();  // the tail's dummy loads still target the dump KB
  if (p.dbg) tk[3] = __builtin_readcyclecounter() - tk[0];

  if (active) {
    float* dst = p.slab + im.slab_off + (((long long)split_i * im.npairs + wave) * 9) * 1024 + lane * 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[tap][g * 4 + e];
        *(f32x4*)(dst + tap * 1024 + g * 256) = v;
      }
    if (bias_wave) {
      bsum += __shfl_xor(bsum, 32);
      if (kh == 0) p.bslab[im.bslab_off + ((long long)split_i * im.ndy + pa) * 32 + blk * 16 + li] = bsum;
    }
  }
  if (p.dbg && lane == 0) {
    long long* o = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = tk[0]; o[1] = tk[1]; o[2] = tk[2]; o[3] = tk[3]; o[4] = __builtin_readcyclecounter() - tk[0]; o[5] = nsteps; o[6] = tk[4]; o[7] = item_i;
  }
}

long long* g_wgrad_phase_clocks = nullptr;
int g_rdb_xcd_map = 1;  // development switch (sr_dev_set_wgrad_xcd_map)

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

// Images a workgroup should walk at least: it leaves P tiles of 36 KB (~28 thousand cycles of its HBM share) whatever it multiplied,
// and an image costs it H rows x 9 MFMAs per live 16-pixel k-step plus ~2 thousand cycles of ring prologue.
int min_imgs_per_wg(int n, int H, int W) {
  const int nseg = W <= 16 ? 1 : W <= 32 ? 2 : 4;
  const long long per_image = (long long)H * nseg * 288 + 2000;
  const long long m = (28000 + per_image - 1) / per_image;
  if (m < 3) return 1;  // (32x32 and larger: measured faster with one image, or a row range of one, per workgroup)
  return (int)(m > n ? n : m);
}

template <int CT, int IT, int KS, int R, int NSTG, int NSEG>
int launch_group_n(const sr_conv3x3_wgrad_desc* d, WgradParamsH p, int cout_tile0, int cin_tile0, int grows, int gi,
                   const SlabCarve& sc, bool want_bias, hipStream_t stream) {
  using G = WgradGeom<CT, IT, KS, R, NSTG, NSEG>;
  constexpr int P = G::P;
  constexpr int lds = G::LDS;
  static_assert(lds <= 160 * 1024, "rings do not fit the LDS");
  const int groups = grows * gi;  // same-shaped tile groups of this launch (grid.y), one slab reduction for all
  p.cin_tile0 = cin_tile0;
  p.cout_tile0 = cout_tile0;
  p.gi = gi;
  // about one workgroup per CU over the whole launch: every workgroup leaves P fp32 tiles of 36 KB, whatever it multiplied — a
  // workgroup of the VGG discriminator's 512-channel layers used to own ONE 8x8 or 4x4 image (1024 workgroups, a 300 MB slab for
  // a 9 MB gradient); small images are walked several per workgroup
  p.n = d->n;
  long long ipw = (long long)d->n * p.strips * groups / 256;
  const int ipw_min = min_imgs_per_wg(d->n, p.H, p.W);
  ipw = ipw < ipw_min ? ipw_min : ipw > d->n ? d->n : ipw;
  p.imgs_per_wg = (int)ipw;
  const long long strips_total = (long long)sr::cdiv(d->n, (int)ipw) * p.strips;
  // (rows of an image are split over workgroups only where one image is more than a workgroup's worth: never below ipw_min's bound)
  const int want = ipw_min > 1 ? 1 : (int)(256 / (strips_total * groups)) > 1 ? (int)(256 / (strips_total * groups)) : 1;
  int rows = sr::cdiv(p.H, want);
  rows = (rows + R - 1) / R * R;
  p.rows_per_wg = rows;
  p.row_splits = sr::cdiv(p.H, rows);
  const long long nwg = strips_total * p.row_splits;
  if ((size_t)groups * nwg * P * 9 * 1024 * sizeof(float) > sc.wslab_bytes ||
      (size_t)grows * nwg * CT * 32 * sizeof(float) > sc.wslab_bytes / 64) {
    sr::set_error("sr_conv3x3_wgrad_bf16: slab too small (need %zu B of tiles)", (size_t)groups * nwg * P * 9 * 1024 * sizeof(float));
    return SR_ENOSPACE;
  }
  p.slab = sc.slab;
  p.bslab = want_bias ? sc.bslab : nullptr;
  p.dbg = g_wgrad_phase_clocks;
  auto kern = wgrad_bf16_kernel<CT, IT, KS, R, NSTG, NSEG>;
  if (int rc = sr::ensure_dynamic_lds((const void*)kern, lds)) return rc;  // once per (kernel, device)
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
    r.flops = 2.0 * 9 * cin_eff * cout_eff * px * groups;
    r.bytes = (2.0 * px * (cin_eff + cout_eff) + 2.0 * nwg * P * 9 * 4096) * groups;
    sr::prof_begin(stream, r);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg, (unsigned)groups), dim3(G::NW * 64), lds, stream, p);
  if (prof) sr::prof_end(stream);
  SR_CHECK_LAUNCH("wgrad_bf16 launch");
  sr::WgradReduce rr = {};
  rr.slab = sc.slab;
  rr.bslab = want_bias ? sc.bslab : nullptr;
  rr.part = sc.part;
  rr.bpart = sc.bpart;
  rr.splits = nwg;
  rr.groups = groups;
  rr.gi = gi;
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

// the instance with as many k-steps per strip row as the image is wide (where the wave split of the shape allows it)
template <int CT, int IT, int KS, int R, int NSTG>
int launch_group(const sr_conv3x3_wgrad_desc* d, WgradParamsH p, int cout_tile0, int cin_tile0, int grows, int gi,
                 const SlabCarve& sc, bool want_bias, hipStream_t stream) {
  if constexpr (R % KS == 0) {
    if (p.W <= 16) return launch_group_n<CT, IT, KS, R, NSTG, 1>(d, p, cout_tile0, cin_tile0, grows, gi, sc, want_bias, stream);
  }
  if constexpr (R * 2 % KS == 0) {
    if (p.W <= 32) return launch_group_n<CT, IT, KS, R, NSTG, 2>(d, p, cout_tile0, cin_tile0, grows, gi, sc, want_bias, stream);
  }
  return launch_group_n<CT, IT, KS, R, NSTG, 4>(d, p, cout_tile0, cin_tile0, grows, gi, sc, want_bias, stream);
}

}  // namespace

namespace sr {
namespace {
struct RdbRow {  // the pairs of one (conv, cout tile) over a window of cin tiles
  int conv, cout_tile, x_tile0, nx, dy_cb0, dy_cbn;
};
struct RdbPlan {
  int nitems, nrows;
  RdbItem items[8];
  RdbRow rows[32];
  int row_item[32], row_pair0[32], row_a[32];
  long long pairs;
};
// Greedy packing, largest rows first: a row joins an item with room for its pairs (<= 8), a free dY slot (<= 2) and the same first
// cin tile — among those the one that ends up staging the FEWEST tile rows per step (X tiles + dY tiles).  The kernel runs at the
// rate its workgroups can stage rows (its waves sit blocked in LDS-DMA issue for a quarter of a step, tools/rdb_wgrad_phase.py) and
// every workgroup walks the same number of rows, so the item that stages the most per row sets the launch time: first-fit gave
// {conv5 t0, conv1} 8 tile rows against {conv3} 5; this gives 7 / 7 / 7 / 6.  Returns false if the block needs more than 8 items.
bool rdb_plan(int nf, int gc, float* const* dparams, RdbPlan* P) {
  const int nfp = (nf + 15) / 16 * 16, gcp = (gc + 15) / 16 * 16;
  P->nitems = P->nrows = 0;
  P->pairs = 0;
  for (int k = 5; k >= 1; --k) {  // conv5, conv4, ...: descending cin = descending row size
    if (dparams && !dparams[2 * (k - 1)]) continue;
    const int cout = k == 5 ? nf : gc;
    const int its = cdiv(nfp + (k - 1) * gcp, 32), cts = cdiv(cout, 32);
    const int dy_ch0 = k == 5 ? 0 : nfp + (4 - k) * gcp;
    for (int c = 0; c < cts; ++c)
      for (int i0 = 0; i0 < its; i0 += RDB_MAXX) {
        if (P->nrows >= 32) return false;
        RdbRow& r = P->rows[P->nrows++];
        r.conv = k;
        r.cout_tile = c;
        r.x_tile0 = i0;
        r.nx = its - i0 < RDB_MAXX ? its - i0 : RDB_MAXX;
        r.dy_cb0 = dy_ch0 / 16 + c * 2;
        const int left = (cout + 15) / 16 - c * 2;
        r.dy_cbn = left < 2 ? left : 2;
      }
  }
  for (int ri = 0; ri < P->nrows; ++ri) {
    const RdbRow& r = P->rows[ri];
    int dst = -1, best = 1 << 30;
    for (int i = 0; i < P->nitems; ++i) {
      const RdbItem& im = P->items[i];
      if (!(im.npairs + r.nx <= 8 && im.ndy < 2 && im.x_tile0 == r.x_tile0)) continue;
      const int staged = (im.nx > r.nx ? im.nx : r.nx) + im.ndy + 1;
      if (staged < best) {
        best = staged;
        dst = i;
      }
    }
    if (dst < 0) {
      if (P->nitems >= 8) return false;
      dst = P->nitems++;
      RdbItem& im = P->items[dst];
      im = RdbItem{};
      im.x_tile0 = r.x_tile0;
      im.bias_wave_0 = im.bias_wave_1 = -1;
    }
    RdbItem& im = P->items[dst];
    const int a = im.ndy++;
    (a ? im.dy_cb0_1 : im.dy_cb0_0) = r.dy_cb0;
    (a ? im.dy_cbn_1 : im.dy_cbn_0) = r.dy_cbn;
    if (r.nx > im.nx) im.nx = r.nx;
    P->row_item[ri] = dst;
    P->row_pair0[ri] = im.npairs;
    P->row_a[ri] = a;
    for (int b = 0; b < r.nx; ++b) {
      im.pa_bits |= (unsigned)a << im.npairs;
      im.pb_nib |= (unsigned)b << (4 * im.npairs);
      ++im.npairs;
    }
    P->pairs += r.nx;
  }
  return true;
}
void rdb_grid(const RdbPlan& P, int n, int h, int w, int* rows_per_wg, int* row_splits, long long* splits) {
  const long long strips_total = (long long)n * cdiv(w, 64);
  long long want = 256 / (strips_total * P.nitems);
  if (want < 1) want = 1;
  int rows = cdiv(h, (int)(want < h ? want : h));
  *rows_per_wg = rows;
  *row_splits = cdiv(h, rows);
  *splits = strips_total * *row_splits;
}
}  // namespace

size_t rdb_wgrad_slab_bytes(int n, int h, int w, int nf, int gc) {
  RdbPlan P;
  if (n <= 0 || h <= 0 || w <= 0 || nf <= 0 || gc <= 0 || !rdb_plan(nf, gc, nullptr, &P) || P.nitems == 0) return 0;
  int rows, rsplits;
  long long splits;
  rdb_grid(P, n, h, w, &rows, &rsplits, &splits);
  const size_t wbytes = (size_t)splits * P.pairs * 9 * 1024 * sizeof(float);
  const size_t bbytes = (size_t)splits * 16 * 32 * sizeof(float);
  const size_t head = wbytes > 64 * bbytes ? wbytes : 64 * bbytes;
  return (head + head / 64 + (kPartFloats + kBpartFloats) * sizeof(float) + 3 * 4096) / 256 * 256;
}

// Weight gradients of the five convs of one residual dense block (reference rrdbnet_arch.py:21-25) in one launch:
// cat = the block's concat buffer [x | x1..x4] (CB16), D = its gradient concat buffer [dY5 | dY4 | dY3 | dY2 | dY1]
// (both with image stride ns elements); dparams[2k], dparams[2k+1] = fp32 dweight / dbias of conv k+1 (null weight:
// skipped); scale5 multiplies conv5's gradient (D[0:nf] holds dL/d(out), dY5 = scale5 * that).
int rdb_wgrad_bf16(const void* cat, const void* D, long long ns, int n, int h, int w, int nf, int gc, float* const* dparams,
                   float scale5, int accumulate, void* slab, size_t slab_bytes, hipStream_t stream) {
  const int nfp = (nf + 15) / 16 * 16, gcp = (gc + 15) / 16 * 16;
  RdbPlan P;
  if (!rdb_plan(nf, gc, dparams, &P)) {
    set_error("rdb_wgrad_bf16: the block does not fit 8 work items (nf=%d gc=%d)", nf, gc);
    return SR_EINVAL;
  }
  if (P.nitems == 0) return SR_OK;
  const SlabCarve sc = carve_slab(slab, slab_bytes);
  RdbWgradParams p = {};
  RdbItems items = {};
  p.zero = zero_line();
  p.x = (const __bf16*)cat;
  p.dy = (const __bf16*)D;
  p.slab = sc.slab;
  p.bslab = sc.bslab;
  p.x_ns = p.dy_ns = ns;
  p.H = h;
  p.W = w;
  p.cin_blocks = (nfp + 4 * gcp) / 16;
  p.strips = cdiv(w, 64);
  p.nitems = P.nitems;
  p.dbg = g_wgrad_phase_clocks;
  long long splits;
  rdb_grid(P, n, h, w, &p.rows_per_wg, &p.row_splits, &splits);
  p.splits = (int)splits;
  p.xcd_map = g_rdb_xcd_map && splits % 8 == 0;
  long long soff = 0, boff = 0;
  for (int i = 0; i < P.nitems; ++i) {
    items.it[i] = P.items[i];
    items.it[i].slab_off = soff;
    items.it[i].bslab_off = boff;
    soff += splits * P.items[i].npairs * 9 * 1024;
    boff += splits * P.items[i].ndy * 32;
  }
  for (int ri = 0; ri < P.nrows; ++ri) {  // the first pair of a row that starts at cin tile 0 also sums its bias gradient
    const RdbRow& r = P.rows[ri];
    if (r.x_tile0 == 0 && dparams[2 * (r.conv - 1) + 1])
      (P.row_a[ri] ? items.it[P.row_item[ri]].bias_wave_1 : items.it[P.row_item[ri]].bias_wave_0) = P.row_pair0[ri];
  }
  if ((size_t)soff * sizeof(float) > sc.wslab_bytes || (size_t)boff * sizeof(float) > sc.wslab_bytes / 64) {
    set_error("rdb_wgrad_bf16: slab too small (need %zu B of tiles)", (size_t)soff * sizeof(float));
    return SR_ENOSPACE;
  }
  const auto kern = w <= 32 ? wgrad_rdb_bf16_kernel<2> : wgrad_rdb_bf16_kernel<4>;
  if (int rc = sr::ensure_dynamic_lds((const void*)kern, RDB_LDS)) return rc;  // once per (kernel, device)
  const bool prof = prof_on();
  if (prof) {
    sr_launch_record r = {};
    r.kernel_id = 40;
    r.cin = nfp + 4 * gcp;
    r.cout = nf + 4 * gc;
    r.n = n;
    r.h = h;
    r.w = w;
    const double px = (double)n * h * w;
    double fl = 0, by = 0;
    for (int k = 1; k <= 5; ++k)
      if (dparams[2 * (k - 1)]) fl += 2.0 * 9 * (nf + (k - 1) * gc) * (k == 5 ? nf : gc) * px;
    for (int i = 0; i < P.nitems; ++i)
      by += 2.0 * px * 32 * (P.items[i].nx + P.items[i].ndy) + 2.0 * splits * P.items[i].npairs * 9 * 4096;
    r.flops = fl;
    r.bytes = by;
    prof_begin(stream, r);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(splits * P.nitems)), dim3(512), RDB_LDS, stream, p, items.it[0],
                     items.it[1], items.it[2], items.it[3], items.it[4], items.it[5], items.it[6], items.it[7]);
  if (prof) prof_end(stream);
  SR_CHECK_LAUNCH("wgrad_rdb_bf16 launch");
  WgradReduce rows[8];  // all rows of the block reduce in two launches
  int nr = 0;
  for (int ri = 0; ri < P.nrows; ++ri) {
    const RdbRow& r = P.rows[ri];
    const RdbItem& im = items.it[P.row_item[ri]];
    const int k = r.conv;
    const bool bias = r.x_tile0 == 0 && dparams[2 * (k - 1) + 1];
    WgradReduce rr = {};
    rr.slab = sc.slab + im.slab_off + (long long)P.row_pair0[ri] * 9 * 1024;
    rr.split_stride = (long long)im.npairs * 9 * 1024;
    rr.bslab = bias ? sc.bslab + im.bslab_off + P.row_a[ri] * 32 : nullptr;
    rr.bsplit_stride = im.ndy * 32;
    rr.part = sc.part;
    rr.bpart = sc.bpart;
    rr.splits = splits;
    rr.P = r.nx;
    rr.IT = r.nx;
    rr.CT = 1;
    rr.ntap = 9;
    rr.ks = 3;
    rr.kdim = 3;
    rr.t_mul = 1;
    rr.cin_tile0 = r.x_tile0;
    rr.cout_tile0 = r.cout_tile;
    rr.cout = k == 5 ? nf : gc;
    rr.cin = nf + (k - 1) * gc;
    rr.first_seg = nf;
    rr.seg = gc;
    rr.seg_pad = 16;
    rr.scale = k == 5 ? scale5 : 1.f;
    rr.accumulate = accumulate;
    rr.dw = dparams[2 * (k - 1)];
    rr.db = bias ? dparams[2 * (k - 1) + 1] : nullptr;
    if (!rr.dw) continue;  // frozen parameter
    rows[nr++] = rr;
    if (nr == 8) {
      if (int rc = wgrad_reduce_rows(rows, nr, stream)) return rc;
      nr = 0;
    }
  }
  return wgrad_reduce_rows(rows, nr, stream);
}
}  // namespace sr

// Development aid (tools/bf16_phase.py; not part of the ABI): per-wave phase clocks of the next launches.
extern "C" void sr_dev_wgrad_bf16_phase_clocks(void* buf) { g_wgrad_phase_clocks = (long long*)buf; }
// Development switch (not part of the ABI): 0 = the dense block's weight-gradient items in plain order (items fastest).
extern "C" void sr_dev_set_wgrad_xcd_map(int on) { g_rdb_xcd_map = on; }

extern "C" size_t sr_conv3x3_wgrad_slab_bytes_bf16(int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  // per-conv launches: <= max_wgs workgroups of 8 tiles; the dense-block launch (sr_rdb_wgrad_bf16 at the same n, h, w):
  // <= 256 + strips workgroups... of <= 3 tiles each, i.e. never more than the first bound
  const size_t wbytes = max_wgs(n, w) * 8 * 9 * 1024 * sizeof(float);
  return (wbytes + wbytes / 64 + (kPartFloats + kBpartFloats) * sizeof(float) + 3 * 4096) / 256 * 256;
}

extern "C" size_t sr_rdb_wgrad_slab_bytes_bf16(int n, int h, int w, int nf, int gc) { return sr::rdb_wgrad_slab_bytes(n, h, w, nf, gc); }

extern "C" int sr_rdb_wgrad_bf16(const void* cat, const void* D, int64_t img_stride, int n, int h, int w, int nf, int gc,
                                 float* const* host_dparams, float scale5, int accumulate, void* slab, size_t slab_bytes,
                                 void* stream) {
  SR_CHECK_ARG(cat && D && host_dparams && slab && n > 0 && h > 0 && w > 0 && nf > 0 && gc > 0, "sr_rdb_wgrad_bf16: bad argument");
  SR_CHECK_ARG(((uintptr_t)cat | (uintptr_t)D | (uintptr_t)slab) % 16 == 0, "sr_rdb_wgrad_bf16: pointers must be 16-byte aligned");
  SR_CHECK_ARG((long long)h * w * 2 * (nf + 4 * gc + 15) < (1ll << 31), "sr_rdb_wgrad_bf16: image too large for 32-bit buffer offsets");
  return sr::rdb_wgrad_bf16(cat, D, img_stride, n, h, w, nf, gc, host_dparams, scale5, accumulate, slab, slab_bytes,
                            (hipStream_t)stream);
}

extern "C" int sr_conv3x3_wgrad_bf16(const sr_conv3x3_wgrad_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(d && d->x && d->dy && d->dweight && d->slab, "sr_conv3x3_wgrad_bf16: null argument");
  SR_CHECK_ARG(d->cout > 0 && d->cin > 0 && d->n > 0 && d->in_h > 0 && d->in_w > 0, "sr_conv3x3_wgrad_bf16: bad shape");
  SR_CHECK_ARG((long long)d->in_h * d->in_w * (d->upsample ? 4 : 1) * 2 * ((d->cout > d->cin_pad ? d->cout : d->cin_pad) + 15) < (1ll << 31),
               "sr_conv3x3_wgrad_bf16: image too large for 32-bit buffer offsets");
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
  // walk the (cout tile, cin tile) grid in workgroup-sized groups; the 2 x 4 groups of wide layers (the discriminators'
  // 512-channel convs: up to 64 of them) go out as one launch per chunk of cout rows with one slab reduction
  const int cts = sr::cdiv(d->cout, 32), its = sr::cdiv(cin_pad, 32);
  const long long strips_total = (long long)d->n * p.strips;
  for (int c0 = 0; c0 < cts;) {
    const int cn = (cts - c0 >= 2) ? 2 : 1;
    int i0 = 0;
    if (cn == 2 && its >= 4 && its / 4 <= 32) {
      const int gi = its / 4, rows_left = (cts - c0) / 2;
      // slab: groups * workgroups tiles of 8 pairs; partial buffers: groups * 8 pairs <= 256 and rows * 2 * 32 bias floats
      // (workgroups per tile group: at least min_imgs_per_wg images each — launch_group_n may give a workgroup more, never fewer)
      const long long wgs_per_group = (long long)sr::cdiv(d->n, min_imgs_per_wg(d->n, p.H, p.W)) * p.strips;
      long long cap = (long long)(sc.wslab_bytes / ((size_t)8 * 9 * 1024 * sizeof(float))) / (wgs_per_group > 0 ? wgs_per_group : 1);
      if (cap > 32) cap = 32;
      SR_CHECK_ARG(cap >= 1, "sr_conv3x3_wgrad_bf16: slab too small for one tile group of %lld strips", strips_total);
      long long rows = 1;
      if (cap >= gi) {  // whole rows of cin groups per launch
        rows = cap / gi;
        if (rows > rows_left) rows = rows_left;
        int rc = launch_group<2, 4, 1, 1, 4>(d, p, c0, 0, (int)rows, gi, sc, d->dbias != nullptr, stream);
        if (rc) return rc;
      } else {  // a very wide layer (the 2048-channel unshuffled input of the last strided conv): one row in cin chunks
        for (int g0 = 0; g0 < gi; g0 += (int)cap) {
          const int gn = gi - g0 < (int)cap ? gi - g0 : (int)cap;
          int rc = launch_group<2, 4, 1, 1, 4>(d, p, c0, g0 * 4, 1, gn, sc, d->dbias != nullptr && g0 == 0, stream);
          if (rc) return rc;
        }
      }
      i0 = gi * 4;
      // the remaining cin tiles of these rows, row by row
      for (int r = 0; r < rows; ++r) {
        int j0 = i0;
        while (j0 < its) {
          const int left = its - j0;
          int in, rc2;
          if (left >= 2) {
            in = 2;
            rc2 = launch_group<2, 2, 2, 2, 3>(d, p, c0 + 2 * r, j0, 1, 1, sc, false, stream);
          } else {
            in = 1;
            rc2 = launch_group<2, 1, 4, 2, 4>(d, p, c0 + 2 * r, j0, 1, 1, sc, false, stream);
          }
          if (rc2) return rc2;
          j0 += in;
        }
      }
      c0 += 2 * (int)rows;
      continue;
    }
    for (; i0 < its;) {
      const int left = its - i0;
      const bool bias = d->dbias != nullptr && i0 == 0;
      int rc, in;
      if (cn == 2) {
        if (left >= 2) {
          in = 2;
          rc = launch_group<2, 2, 2, 2, 3>(d, p, c0, i0, 1, 1, sc, bias, stream);
        } else {
          in = 1;
          rc = launch_group<2, 1, 4, 2, 4>(d, p, c0, i0, 1, 1, sc, bias, stream);
        }
      } else {
        if (left == 5 || left >= 9) {  // conv4 of a dense block: one 5-wave pass instead of 4 + 1
          in = 5;
          rc = launch_group<1, 5, 1, 1, 4>(d, p, c0, i0, 1, 1, sc, bias, stream);
        } else if (left >= 4) {
          in = 4;
          rc = launch_group<1, 4, 2, 1, 4>(d, p, c0, i0, 1, 1, sc, bias, stream);
        } else if (left == 3) {
          in = 3;
          rc = launch_group<1, 3, 2, 1, 4>(d, p, c0, i0, 1, 1, sc, bias, stream);
        } else if (left == 2) {
          in = 2;
          rc = launch_group<1, 2, 4, 2, 4>(d, p, c0, i0, 1, 1, sc, bias, stream);
        } else {
          in = 1;
          rc = launch_group<1, 1, 8, 2, 4>(d, p, c0, i0, 1, 1, sc, bias, stream);
        }
      }
      if (rc) return rc;
      i0 += in;
    }
    c0 += cn;
  }
  return SR_OK;
}
