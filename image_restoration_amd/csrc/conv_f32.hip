// Fused convolution on fp32 MFMA for gfx950 (MI355X): 3x3/s1 (generator, discriminators) and, as four
// parity passes of a 2x2 tap grid, 4x4/s2 (discriminators) and its data gradient.
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

// A/B switches (tools/ab_f32.sh builds the variants; measured on one box, BASELINE config 2, 20 steps):
#ifndef SR_F32_PIPELINE
#define SR_F32_PIPELINE 0  // operand reads of tap t+1 issued before the MFMAs of tap t: 216.5 vs 219.0 img/s without (two waves per
#endif                     // SIMD from different workgroups already cover the LDS latency; the extra registers cost more)
#ifndef SR_F32_BUFDMA
#define SR_F32_BUFDMA 1  // LDS-DMA through buffer descriptors (scalar base + 32-bit lane offset) instead of 64-bit flat addresses
#endif
#ifndef SR_F32_SWIZZLE
#define SR_F32_SWIZZLE 1  // LDS bank swizzle of the two 16-byte halves of a pixel / cout (see conv_bf16.hip)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {


struct ConvParams {
  const void* zero;  // sr::zero_line()
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
  int in_h, in_w;   // real source spatial size
  int H, W;         // compute-space output size (tiles cover this)
  int tiles_x, tiles_y;
  // Source map: tap (ty,tx) of compute pixel (y,x) reads VIRTUAL pixel (y+tap_oy+ty, x+tap_ox+tx), valid inside
  // [0,vH)x[0,vW), which is REAL pixel ((v*src_mul)>>src_shift)+src_o.  3x3/s1: identity, tap_o = -1; nearest x2
  // upsample: shift 1; parity sub-image of a stride-2 conv: mul 2, off = parity.
  int vH, vW, src_mul, src_shift, src_oy, src_ox, tap_oy, tap_ox;
  // Destination map: compute pixel (y,x) is REAL pixel (y*dst_mul+dst_oy, x*dst_mul+dst_ox) of an oH x oW image.
  int oH, oW, dst_mul, dst_oy, dst_ox;
  int acc_first;    // accumulate the old destination value BEFORE bias/activation (last pass of a multi-pass conv)
  int mask_cb0, mask_cb1;
  int res_cb1;      // residuals apply to destination blocks [0, res_cb1)
  float slope, alpha, beta1, beta2, mask_slope;
  int accumulate;
  // Stacked mode (small feature maps, WT < 32): the n images are treated as ONE image of n*stack_hs rows — image i
  // occupies virtual rows [i*stack_hs, i*stack_hs + H) and row i*stack_hs + H is a separator that reads as zero padding
  // for both neighbours (one pad row is all a 3x3 / parity-2x2 tap grid needs) — so tiles are not confined to one tiny
  // image.  0 = off (tiles of one image, image index from the grid).
  int stack_hs, stack_n;
  // Weight image addressing (floats): chunk cb, tap t, cout sub-tile c of the workgroup's tile start at
  // w + cb * w_chunk + t * w_tap + c * 256.  0 = dense image of this kernel's own COT (w_chunk = taps * COT * 256, w_tap = COT * 256);
  // the chain kernel runs 32-cout tiles on images packed for 64-cout groups (w_tap = 512, w + 256 for the upper half).
  int w_chunk, w_tap, w_cog;  // w_cog: floats between the images of consecutive cout groups (0 = cin_blocks * w_chunk)
};

__device__ __forceinline__ void glds16(const float* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// ABL: timing-only ablation bits for tools/conv_ablate.hip (never instantiated non-zero in the library):
//   1 = no LDS-DMA refill in the loop, 2 = no per-chunk barrier, 8 = no output store
// NW = waves per workgroup (4 or 8): 8 waves share one staged weight chunk over twice the rows.
// WT = tile width in pixels (32, 16, 8): an MFMA column group of 32 pixels is 32/WT tile rows x WT columns, so maps
//      narrower than 32 pixels (the 16x16 ... 4x4 layers of VGGStyleDiscriminator128) do not idle 50-88 % of the lanes.
// 16-byte store; write-through (sc1) when another workgroup of the same launch will read the tile (chain kernel).
__device__ __forceinline__ void store16f(float* ptr, f32x4 v, bool write_through) {
  if (write_through)
    // + two wait states: a 16-byte store reads its data registers after it has issued, and behind an asm the compiler does not
    // know that the next vector instruction must not overwrite them yet (the VMEM store-data hazard)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(ptr), "v"(v) : "memory");
  else
    *(f32x4*)ptr = v;
}

// One output tile of one conv: the body of conv_f32_kernel, also run once per work item by the persistent chain kernel.
template <int COT, int PT, int KS, bool NCHW_OUT, int ABL, int NW, int WT, bool WT_OUT>
__device__ __forceinline__ void conv_tile_f32(const ConvParams p, const int cog, const int tx, const int ty, const int n_grid,
                                              char* smem) {
  constexpr int RPG = 32 / WT;  // tile rows per MFMA column group
  constexpr int TH = NW * PT * RPG, XROW = WT + KS - 1, XPIX = (TH + KS - 1) * XROW;
  constexpr int XBYTES = ((XPIX * 32 + 1023) / 1024) * 1024;
  constexpr int NXU = XBYTES / 1024, NWU = KS * KS * COT;
  constexpr int WBYTES = NWU * 1024, STAGE = XBYTES + WBYTES;
  constexpr int NXR = (NXU + NW - 1) / NW, NWR = (NWU + NW - 1) / NW;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int jr = j / WT, jc = j % WT;  // this lane's pixel inside a column group

  const int n = n_grid;
  const int x0 = tx * WT, y0 = ty * TH;
  const int HWin = p.in_h * p.in_w;
  const bool stacked = WT < 32 && p.stack_hs > 0;
  const float* in_n = p.in + (stacked ? 0 : (long long)n * p.in_ns);
  // weight image addressing: dense (constants) in the per-conv kernels; general (ConvParams::w_chunk / w_tap / w_cog) in the chain
  // kernel, which runs 32-cout tiles on images packed for 64-cout groups
  const int w_chunk = WT_OUT && p.w_chunk ? p.w_chunk : WBYTES / 4, w_tap = WT_OUT && p.w_tap ? p.w_tap : COT * 256;
  const float* wg = p.w + (size_t)cog * (WT_OUT && p.w_cog ? (size_t)p.w_cog : (size_t)p.cin_blocks * (WBYTES / 4));

  // Per-lane source offsets (floats, inside one channel-block plane) of the X pieces this
  // wave moves; -1 = zero padding.
  int xoff[NXR];
#pragma unroll
  for (int r = 0; r < NXR; ++r) {
    const int u = r * NW + wave;
    const int q = u * 64 + lane;
    const int pix = q >> 1, half = q & 1;
    const int row = pix / XROW, col = pix - row * XROW;
    int gy = y0 + p.tap_oy + row;
    const int gx = x0 + p.tap_ox + col;
    int img_off = 0;
    bool valid = (pix < XPIX) && gx >= 0 && gx < p.vW;
    if (stacked) {  // virtual row -> (image, row); the separator row and everything outside the stack read as zero
      const int img = gy >= 0 ? gy / p.stack_hs : -1;
      gy -= img * p.stack_hs;
      valid = valid && img >= 0 && img < p.stack_n;
      img_off = img * (int)p.in_ns;
    }
    valid = valid && gy >= 0 && gy < p.vH;
    const int sy = ((gy * p.src_mul) >> p.src_shift) + p.src_oy, sx = ((gx * p.src_mul) >> p.src_shift) + p.src_ox;
    // LDS bank swizzle (WT == 32): the two 16-byte halves of tile column c are stored swapped when bit 3 of c is set, so the
    // 32 lanes of one half (32-byte stride) cover all 64 banks per ds_read_b128 lane group instead of every slot twice
    // (PMC before: SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE); see conv_bf16.hip's header
    const int hsw = (SR_F32_SWIZZLE && WT == 32) ? (half ^ ((col >> 3) & 1)) : half;
    xoff[r] = valid ? (img_off + (sy * p.in_w + sx) * 8 + hsw * 4) : -1;
  }

#if SR_F32_BUFDMA
  // LDS-DMA through buffer descriptors: base (image / weight image) in scalar registers, a 32-bit per-lane byte offset computed
  // ONCE per tile, the chunk's offset as the scalar `soffset` — a piece is one instruction with no address arithmetic, against a
  // 64-bit per-lane address (select zero line / add plane) per piece of the flat form.  Padding lanes carry an offset beyond
  // num_records: the hardware returns zeros for them.
  const unsigned x_range = (unsigned)((stacked ? (long long)p.stack_n * p.in_ns : (long long)p.cin_blocks * HWin * 8) * 4);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)in_n, 0, x_range, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, (unsigned)((long long)p.cin_blocks * w_chunk * 4), 0x00020000);
  unsigned xvo[NXR];
#pragma unroll
  for (int r = 0; r < NXR; ++r) xvo[r] = xoff[r] >= 0 ? (unsigned)xoff[r] * 4u : 0xfffffff0u;
  const unsigned wvo = (SR_F32_SWIZZLE ? (lane ^ ((lane >> 4) & 1)) : lane) * 16;
  auto stage = [&](int buf, int cb) {
    char* xs = smem + buf * STAGE;
    char* ws = xs + XBYTES;
    const unsigned xso = (unsigned)cb * (unsigned)HWin * 32u, wso = (unsigned)cb * (unsigned)w_chunk * 4u;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int u = r * NW + wave;
      if (u < NXU)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (__attribute__((address_space(3))) void*)(xs + u * 1024), 16, xvo[r], xso, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
      const int u = r * NW + wave;  // unit = tap * COT + cout sub-tile
      if (u < NWU)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (__attribute__((address_space(3))) void*)(ws + u * 1024), 16, wvo,
                                                 wso + (unsigned)(WT_OUT ? (u / COT) * w_tap + (u % COT) * 256 : u * 256) * 4u, 0, 0);
    }
  };
#else
  auto stage = [&](int buf, int cb) {
    char* xs = smem + buf * STAGE;
    char* ws = xs + XBYTES;
    const float* plane = in_n + (size_t)cb * HWin * 8;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int u = r * NW + wave;
      if (u < NXU) {
        const float* src = xoff[r] >= 0 ? plane + xoff[r] : (const float*)p.zero;
        glds16(src, xs + u * 1024);
      }
    }
    const float* wsrc = wg + (size_t)cb * w_chunk + (SR_F32_SWIZZLE ? (lane ^ ((lane >> 4) & 1)) : lane) * 4;  // unit (cout i, half) <- half ^ bit3(i)
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
      const int u = r * NW + wave;  // unit = tap * COT + cout sub-tile
      if (u < NWU) glds16(wsrc + (WT_OUT ? (u / COT) * w_tap + (u % COT) * 256 : u * 256), ws + u * 1024);
    }
  };

#endif

  f32x16 acc[COT][PT];
#pragma unroll
  for (int a = 0; a < COT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // byte offset of this lane's B operand at tap column dx (tap row 0, group 0): tile column jc + dx, swizzled half
  const int xrow0 = ((wave * PT * RPG + jr) * XROW + jc) * 32;
  int xlane[KS];
#pragma unroll
  for (int dx = 0; dx < KS; ++dx)
    xlane[dx] = xrow0 + dx * 32 + (((SR_F32_SWIZZLE && WT == 32) ? (h ^ (((jc + dx) >> 3) & 1)) : h) * 16);
  const int wlane = j * 32 + ((SR_F32_SWIZZLE ? (h ^ ((j >> 3) & 1)) : h) * 16);  // byte offset of this lane's A operand, tap 0, cot 0

  // Operand reads are software-pipelined by tap: the LDS reads of tap t+1 are issued BEFORE the MFMAs of tap t (two register
  // sets, pinned with scheduling barriers).  Left to itself hipcc issues a tap's reads behind the last MFMAs of the previous tap
  // and then waits lgkmcnt(0): ~100 of every ~1100 cycles of a wave with nothing on the matrix pipe.
#if SR_F32_PIPELINE
  struct Ops {
    f32x4 a[COT], b[PT];
  };
  auto load = [&](Ops& o, const char* xb, const char* ws, int tap) {
    const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
    for (int c = 0; c < COT; ++c) o.a[c] = *(const f32x4*)(ws + (tap * COT + c) * 1024);
#pragma unroll
    for (int r = 0; r < PT; ++r) o.b[r] = *(const f32x4*)(xb + xlane[dx] + (r * RPG + dy) * XROW * 32);
  };
  auto mfma = [&](const Ops& o) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int r = 0; r < PT; ++r)
          acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[c][s], o.b[r][s], acc[c][r], 0, 0, 0);
  };
#endif
  auto compute = [&](int buf) {
    const char* xb = smem + buf * STAGE;
    const char* ws = smem + buf * STAGE + XBYTES + wlane;
#if SR_F32_PIPELINE
    Ops o[2];
    load(o[0], xb, ws, 0);
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
      if (tap + 1 < KS * KS) load(o[(tap + 1) & 1], xb, ws, tap + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma(o[tap & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
        const int tap = dy * KS + dx;
        f32x4 a[COT], b[PT];
#pragma unroll
        for (int c = 0; c < COT; ++c) a[c] = *(const f32x4*)(ws + (tap * COT + c) * 1024);
#pragma unroll
        for (int r = 0; r < PT; ++r) b[r] = *(const f32x4*)(xb + xlane[dx] + (r * RPG + dy) * XROW * 32);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < PT; ++r)
              acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][s], b[r][s], acc[c][r], 0, 0, 0);
      }
    }
#endif
  };

  const int nchunk = p.cin_blocks;
  stage(0, 0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    if constexpr (!(ABL & 1)) {
      if (c + 1 < nchunk) stage((c + 1) & 1, c + 1);
    }
    compute(c & 1);
    if constexpr (!(ABL & 2)) __syncthreads();  // drains the LDS-DMA of chunk c+1 and frees buffer c&1
  }

  // ---- epilogue: [old +] bias, LeakyReLU, residual scale-adds, optional accumulate / LReLU-backward mask
  const int x = x0 + jc;
  const long long HW = (long long)p.oH * p.oW;
  const int rx = x * p.dst_mul + p.dst_ox;
#pragma unroll
  for (int r = 0; r < PT; ++r) {
    int y = y0 + (wave * PT + r) * RPG + jr;
    int n = n_grid;
    if (stacked) {
      n = y / p.stack_hs;
      y -= n * p.stack_hs;
      if (n >= p.stack_n) continue;
    }
    const int ry = y * p.dst_mul + p.dst_oy;
    if (y >= p.H || x >= p.W || ry >= p.oH || rx >= p.oW) continue;
    const long long pixoff = (long long)ry * p.oW + rx;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = (cog * COT + c) * 4 + g;
        if (cb >= p.cout_blocks) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[c][r][g * 4 + e];
        const long long off = (cb * HW + pixoff) * 8 + h * 4;
        float* o = p.out + (long long)n * p.out_ns + off;
        if constexpr (!NCHW_OUT) {
          if (p.accumulate && p.acc_first) v += *(const f32x4*)o;
        }
        if (p.bias) {
          const f32x4 bv = *(const f32x4*)(p.bias + cb * 8 + h * 4);
          v += bv;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
        v *= p.alpha;
        if (p.res1 && cb < p.res_cb1) v += p.beta1 * *(const f32x4*)(p.res1 + (long long)n * p.res1_ns + off);
        if (p.res2 && cb < p.res_cb1) v += p.beta2 * *(const f32x4*)(p.res2 + (long long)n * p.res2_ns + off);
        if constexpr (NCHW_OUT) {
          if (h == 0 && cb == 0) {
            float* on = p.out + (long long)n * p.out_ns + pixoff;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < p.cout) on[e * HW] = v[e];
          }
        } else {
          if (p.accumulate && !p.acc_first) v += *(const f32x4*)o;
          if (p.mask && cb >= p.mask_cb0 && cb < p.mask_cb1) {
            const f32x4 m = *(const f32x4*)(p.mask + (long long)n * p.mask_ns + ((cb - p.mask_cb0) * HW + pixoff) * 8 + h * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : v[e] * p.mask_slope;
          }
          if constexpr (ABL & 8) {
            asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));  // timing-only: keep the value, skip the store
          } else {
            store16f(o, v, WT_OUT);
          }
        }
      }
    }
  }
}

template <int COT, int PT, int KS, bool NCHW_OUT, int ABL = 0, int NW = 4, int WT = 32>
__global__ __launch_bounds__(NW * 64, (COT == 2 && PT == 4) ? 2 : 1) void conv_f32_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
  const int n_grid = t / p.tiles_y;
  conv_tile_f32<COT, PT, KS, NCHW_OUT, ABL, NW, WT, false>(p, blockIdx.y, tx, ty, n_grid, smem);
}

// ------------------------------------------------------------------------------------------------ persistent conv chain
// The convs of a residual dense block (rrdbnet_arch.py:32-39) as ONE launch: work items (conv k, tile, cout half) are claimed from a
// counter in k-major order and wait until conv k-1 has finished on the tile's 3x3 neighbourhood.  Same scheme and hand-off as
// conv_chain_bf16_kernel (conv_bf16.hip) — claimed items never depend on unclaimed ones, write-through tile stores + drained
// waves + barrier + agent-scope progress word, one acquire before the dependent loads, bounded spins.  Why, in fp32: a conv1-4
// launch is ONE round of 512 workgroups that start together and finish apart, so 10 % of its wave-slot time is empty (PMC:
// SQ_WAVE_CYCLES against capacity, profiles/r02_pmc_sq_fp32.txt) and the two workgroups of a CU change tiles at the same moment;
// in a chain the next conv's tiles fill those holes.  Every item is a 16x32 tile of 32 couts (conv5 = two items per tile).
#define SR_CHAIN_MAX_F32 5
struct ChainParamsF {
  ConvParams lv[SR_CHAIN_MAX_F32];
  int halves[SR_CHAIN_MAX_F32];   // 32-cout items per tile of conv k (1 or 2)
  int first[SR_CHAIN_MAX_F32 + 1];  // first item of conv k; first[nconv] = number of items
  int need[SR_CHAIN_MAX_F32];     // progress a neighbour tile must show before conv k may read it (items of convs < k)
  int nconv, ntiles, tiles_x, tiles_y;
  int* head;
  int* done;
  int* abort;
  int epoch;
};

__global__ __launch_bounds__(256) void conv_chain_f32_kernel(const ChainParamsF P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_item;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nitems = P.first[P.nconv];
  auto conv_of = [&](int item) {
    int k = 0;
    while (k + 1 < P.nconv && item >= P.first[k + 1]) ++k;
    return k;
  };
  for (;;) {
    if (wave == 0) {
      int item = 0;
      if (lane == 0) item = atomicAdd(P.head, 1);
      item = __builtin_amdgcn_readfirstlane(item);
      if (item < nitems && item >= P.first[1]) {
        const int k = conv_of(item);
        int t = (item - P.first[k]) / P.halves[k];
        const int tx = t % P.tiles_x;
        t /= P.tiles_x;
        const int ty = t % P.tiles_y, n = t / P.tiles_y;
        bool gave_up = false;
        if (lane < 9) {
          const int ny = ty + lane / 3 - 1, nx = tx + lane % 3 - 1;
          if (ny >= 0 && ny < P.tiles_y && nx >= 0 && nx < P.tiles_x) {
            const int* f = P.done + (n * P.tiles_y + ny) * P.tiles_x + nx;
            const int want = P.epoch + P.need[k];
            int spins = 0;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
              __builtin_amdgcn_s_sleep(2);
              if ((++spins & 255) == 0 && (spins > (1 << 22) || __hip_atomic_load(P.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(P.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = true;
                break;
              }
            }
          }
        }
        if (__builtin_amdgcn_ballot_w64(gave_up)) item = nitems;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (lane == 0) s_item = item;
    }
    __syncthreads();
    const int item = s_item;
    if (item >= nitems) break;
    const int k = conv_of(item);
    const int sub = (item - P.first[k]) % P.halves[k];
    int t = (item - P.first[k]) / P.halves[k];
    const int tile = t;
    const int tx = t % P.tiles_x;
    t /= P.tiles_x;
    const int ty = t % P.tiles_y, n = t / P.tiles_y;
    conv_tile_f32<1, 4, 3, false, 0, 4, 32, true>(P.lv[k], sub, tx, ty, n, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have landed
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(P.done + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Few-output-channel 3x3 conv (conv_last: 64 -> 3, rrdbnet_arch.py:118) on v_mfma_f32_4x4x1_16b_f32.
// The 32x32 tile would pad 3 couts to 32 (10x wasted MFMA work, 1.08 ms of a 73 ms step); the 16-block 4x4x1 form
// pads to 4: block b = lane/4 computes D[4 couts][4 pixels] += A[4 couts][1] * B[1][4 pixels], so lane l IS pixel l
// of a 64-pixel group and holds its 4 couts in 4 registers.  A = W[cout l%4][k] (broadcast within a block column),
// B = X[k][pixel l].  With K = 9*Cin steps of 8 cycles the layer is HBM-bound (67 MB read per image).
// Workgroup = 4 waves, tile 16 rows x 32 columns; wave w owns rows 4w..4w+3 as two 64-pixel groups that share the
// weight fragments.  X tile staged exactly like the main kernel; the weight chunk [9 taps][4 couts][8 ch] is
// gathered by LDS-DMA out of the standard packed image (first 4 couts of every tap).
template <int KS>
__global__ __launch_bounds__(256) void conv_fewcout_f32_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TH = 16, XROW = 32 + KS - 1, XPIX = (TH + KS - 1) * XROW;
  constexpr int XBYTES = ((XPIX * 32 + 1023) / 1024) * 1024;
  constexpr int NXU = XBYTES / 1024, NT = KS * KS;
  constexpr int WBYTES = ((NT * 4 * 32 + 1023) / 1024) * 1024, NWU = WBYTES / 1024, STAGE = XBYTES + WBYTES;
  constexpr int NXR = (NXU + 3) / 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const int x0 = tx * 32, y0 = ty * TH;
  const int HWin = p.in_h * p.in_w;
  const float* in_n = p.in + (long long)n * p.in_ns;

  int xoff[NXR];
#pragma unroll
  for (int r = 0; r < NXR; ++r) {
    const int u = r * 4 + wave;
    const int q = u * 64 + lane;
    const int pix = q >> 1, half = q & 1;
    const int row = pix / XROW, col = pix - row * XROW;
    const int gy = y0 + p.tap_oy + row, gx = x0 + p.tap_ox + col;
    const bool valid = (pix < XPIX) && gy >= 0 && gy < p.vH && gx >= 0 && gx < p.vW;
    const int sy = ((gy * p.src_mul) >> p.src_shift) + p.src_oy, sx = ((gx * p.src_mul) >> p.src_shift) + p.src_ox;
    // LDS bank swizzle (as conv_tile_f32): the two 16-byte halves of staged pixel column c are stored swapped when bit 3 of c is
    // set, applied here by the source address; the readers below undo it.  The 32 lanes of a row stride 32 B, so without it every
    // ds_read_b128 lane group hits each 16-byte bank slot twice (PMC round 2: 38 % of this kernel's LDS cycles were conflicts).
    xoff[r] = valid ? ((sy * p.in_w + sx) * 8 + (half ^ ((col >> 3) & 1)) * 4) : -1;
  }
  // weight piece q of the chunk: tap = q/8, cout = (q%8)/2, half = q%2  ->  packed image [tap][32 couts][8]
  const int wq = wave * 64 + lane;
  const int woff = (wq < NT * 8) ? ((wq >> 3) * 32 + ((wq & 7) >> 1)) * 8 + (wq & 1) * 4 : -1;

  auto stage = [&](int buf, int cb) {
    char* xs = smem + buf * STAGE;
    const float* plane = in_n + (size_t)cb * HWin * 8;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int u = r * 4 + wave;
      if (u < NXU) glds16(xoff[r] >= 0 ? plane + xoff[r] : (const float*)p.zero, xs + u * 1024);
    }
    if (wave < NWU) {
      const float* wchunk = p.w + (size_t)cb * (NT * 32 * 8);
      glds16(woff >= 0 ? wchunk + woff : (const float*)p.zero, xs + XBYTES + wave * 1024);
    }
  };

  f32x4 acc[2];
  acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int prow = lane >> 5, pcol = lane & 31;
  const int xlane = ((wave * 4 + prow) * XROW + pcol) * 32;  // this lane's pixel, group 0, tap (0,0)
  const int wlane = (lane & 3) * 32;                         // this lane's cout row of the weight chunk

  auto compute = [&](int buf) {
    const char* xs = smem + buf * STAGE + xlane;
    const char* ws = smem + buf * STAGE + XBYTES + wlane;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
        const int tap = dy * KS + dx;
        const f32x4 w0 = *(const f32x4*)(ws + tap * 128), w1 = *(const f32x4*)(ws + tap * 128 + 16);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const char* xp = xs + ((2 * g + dy) * XROW + dx) * 32;
          const int sw = (((pcol + dx) >> 3) & 1) * 16;  // where this column's first half lives
          const f32x4 a0 = *(const f32x4*)(xp + sw), a1 = *(const f32x4*)(xp + (sw ^ 16));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w0[e], a0[e], acc[g], 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w1[e], a1[e], acc[g], 0, 0, 0);
        }
      }
  };

  const int nchunk = p.cin_blocks;
  stage(0, 0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    if (c + 1 < nchunk) stage((c + 1) & 1, c + 1);
    compute(c & 1);
    __syncthreads();
  }
  // epilogue: bias, LeakyReLU, scale; NCHW store (lane = pixel: 32 consecutive x per row and channel)
  const long long HW = (long long)p.oH * p.oW;
  const int x = x0 + pcol;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int y = y0 + wave * 4 + 2 * g + prow;
    if (y >= p.H || x >= p.W) continue;
    float* o = p.out + (long long)n * p.out_ns + (long long)y * p.oW + x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e >= p.cout) break;
      float v = acc[g][e] + (p.bias ? p.bias[e] : 0.f);
      v = v > 0.f ? v : v * p.slope;
      o[e * HW] = v * p.alpha;
    }
  }
}

template <int COT, int PT, int KS, int NW = 4, int WT = 32>
constexpr int conv_lds_bytes() {
  return 2 * ((((NW * PT * (32 / WT) + KS - 1) * (WT + KS - 1) * 32 + 1023) / 1024) * 1024 + KS * KS * COT * 1024);
}

template <int COT, int PT, int KS, bool NCHW_OUT, int WT = 32>
int launch(const ConvParams& p, int n, int groups, hipStream_t stream, const sr_conv3x3_desc* d) {
  constexpr int lds = conv_lds_bytes<COT, PT, KS, 4, WT>();
  auto kern = conv_f32_kernel<COT, PT, KS, NCHW_OUT, 0, 4, WT>;
  if (int rc = sr::ensure_dynamic_lds((const void*)kern, lds)) return rc;  // once per (kernel, device)
  dim3 grid(p.tiles_x * p.tiles_y * n, groups);
  const bool prof = sr::prof_on();
  if (prof) {
    sr_launch_record r = {};
    r.kernel_id = WT < 32 ? 44 + (KS == 2 ? 2 : 0) + (WT == 8 ? 1 : 0)
                          : PT == 4 ? 13 : PT == 1 ? (COT == 1 ? 15 : 41) : (COT - 1) * 4 + (KS == 2 ? 2 : 0) + (NCHW_OUT ? 1 : 0);
    r.cin = d->cin_real > 0 ? d->cin_real : d->cin_pad;
    r.cout = d->cout;
    r.n = n;
    r.h = p.H;
    r.w = p.W;
    const double px = (double)n * p.H * p.W;
    r.flops = 2.0 * KS * KS * r.cin * r.cout * px;
    const double in_px = KS == 3 ? (double)n * p.in_h * p.in_w : px;  // a parity pass reads 1/4 of its source
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
  SR_CHECK_LAUNCH("conv_f32 launch");
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

namespace {

// Fills the fields every entry point shares and validates the descriptor.
int fill_common(const sr_conv3x3_desc* d, ConvParams* pp, const char* who) {
  SR_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  SR_CHECK_ARG(d->in && d->wpacked && d->out, "%s: null in/wpacked/out", who);
  SR_CHECK_ARG(d->cin_pad > 0 && d->cin_pad % 8 == 0, "%s: cin_pad=%d must be a positive multiple of 8", who, d->cin_pad);
  SR_CHECK_ARG(d->cout > 0 && d->n > 0 && d->in_h > 0 && d->in_w > 0, "%s: bad shape", who);
  SR_CHECK_ARG(!d->out_nchw || d->cout <= 4, "%s: out_nchw needs cout <= 4 (got %d)", who, d->cout);
  SR_CHECK_ARG(!(d->out_nchw && (d->accumulate || d->mask_src)), "%s: out_nchw excludes accumulate/mask", who);
  SR_CHECK_ARG(((uintptr_t)d->in | (uintptr_t)d->wpacked | (uintptr_t)d->out | (uintptr_t)d->res1 | (uintptr_t)d->res2 |
                (uintptr_t)d->mask_src | (uintptr_t)d->bpacked) % 16 == 0,
               "%s: pointers must be 16-byte aligned", who);
  ConvParams& p = *pp;
  p = ConvParams{};
  p.zero = sr::zero_line();
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
  p.res_cb1 = d->res_cbn > 0 ? d->res_cbn : (1 << 30);
  p.mask_cb0 = d->mask_cb0;
  p.mask_cb1 = d->mask_cb0 + d->mask_cbn;
  p.slope = d->act_slope;
  p.alpha = d->alpha;
  p.beta1 = d->beta1;
  p.beta2 = d->beta2;
  p.mask_slope = d->mask_slope;
  p.accumulate = d->accumulate;
  p.src_mul = 1;
  p.dst_mul = 1;
  return SR_OK;
}

int check_sizes(const ConvParams& p, int n, const char* who) {
  const long long blocks = p.cout_blocks > p.cin_blocks ? p.cout_blocks : p.cin_blocks;
  SR_CHECK_ARG((long long)p.oH * p.oW * 8 * blocks < (1ll << 31) && (long long)p.in_h * p.in_w * 8 * blocks < (1ll << 31),
               "%s: image too large for 32-bit plane offsets", who);
  SR_CHECK_ARG((long long)p.tiles_x * p.tiles_y * n < (1ll << 31), "%s: grid too large", who);
  return SR_OK;
}

// Small feature maps (W <= 16: the 16x16 ... 4x4 layers of VGGStyleDiscriminator128, discriminator_arch.py:33-46): a
// 32-pixel-wide tile would idle 50-88 % of its lanes and tiles could not be taller than one tiny image.  Narrow tiles
// (WT = 16 / 8) over the vertically STACKED batch (ConvParams::stack_hs) fix both.  Returns -1 when not applicable.
template <int KS>
int launch_small(ConvParams q, const sr_conv3x3_desc* d, int groups, int gc, hipStream_t stream) {
  if (gc != 64 || d->out_nchw || q.W > 16 || q.vH != q.H || q.vW != q.W) return -1;
  if ((long long)d->n * q.in_ns + (long long)q.in_h * q.in_w * 8 >= (1ll << 31)) return -1;
  q.stack_hs = q.H + 1;
  q.stack_n = d->n;
  const long long vrows = (long long)d->n * q.stack_hs;
  if (q.W > 8) {
    q.tiles_x = sr::cdiv(q.W, 16);
    const bool big = vrows / 16 * q.tiles_x * groups >= 128;  // 4 waves x PT x 2 rows per tile
    q.tiles_y = (int)sr::cdiv((int)vrows, big ? 16 : 8);
    return big ? launch<2, 2, KS, false, 16>(q, 1, groups, stream, d) : launch<2, 1, KS, false, 16>(q, 1, groups, stream, d);
  }
  q.tiles_x = sr::cdiv(q.W, 8);
  const bool big = vrows / 32 * q.tiles_x * groups >= 128;  // 4 waves x PT x 4 rows per tile
  q.tiles_y = (int)sr::cdiv((int)vrows, big ? 32 : 16);
  return big ? launch<2, 2, KS, false, 8>(q, 1, groups, stream, d) : launch<2, 1, KS, false, 8>(q, 1, groups, stream, d);
}

}  // namespace

// Off by default: 221.0 vs 223.0 img/s conv by conv on the same box (BASELINE config 2).  The chain fills the 10 % of empty
// wave-slot time of the one-round conv1-4 launches, but runs conv5 as two 32-cout items and pays the hand-offs.
static bool g_chain_f32_enabled = false;
static bool g_f32_tall64 = false;
static int g_f32_rows8 = 1;  // 32-cout convs on 8-row tiles (1, default: two rounds of workgroups) / 4-row tiles (2) / 16-row tiles (0)
extern "C" int sr_dev_set_f32_rows8(int on) {  // development switch: 32-cout convs on 8-row tiles (two rounds of workgroups)
  g_f32_rows8 = on;
  return SR_OK;
}
extern "C" int sr_dev_set_f32_tall64(int on) {  // development switch (not in the ABI header)
  g_f32_tall64 = on != 0;
  return SR_OK;
}
extern "C" int sr_set_conv_chain_f32(int enabled) {
  g_chain_f32_enabled = enabled != 0;
  return SR_OK;
}

// fp32 twin of sr_conv3x3_chain_bf16 (conv_bf16.hip): same contract, same sync block (sr_conv3x3_chain_sync_ints).
extern "C" int sr_conv3x3_chain_f32(const sr_conv3x3_desc* d, int nconv, int32_t* sync, int call_index, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SR_CHECK_ARG(d && nconv >= 1, "sr_conv3x3_chain_f32: bad argument");
  constexpr int kEpochs = 256;  // SR_CHAIN_EPOCHS of conv_bf16.hip: layout of the sync block
  bool one_launch = g_chain_f32_enabled && sync && nconv >= 2 && nconv <= SR_CHAIN_MAX_F32 && call_index >= 0 && call_index < kEpochs &&
                    !sr::prof_on();
  for (int k = 0; k < nconv && one_launch; ++k) {
    const sr_conv3x3_desc& c = d[k];
    one_launch = c.n == d[0].n && c.in_h == d[0].in_h && c.in_w == d[0].in_w && !c.upsample && !c.out_nchw && !c.accumulate &&
                 c.cout <= 64 && c.in_h % 16 == 0 && c.s2_channels == 0;
  }
  const int conc = sr::launch_concurrency();
  if (one_launch) one_launch = (long long)sr::cdiv(d[0].in_w, 32) * (d[0].in_h / 16) * d[0].n * conc >= 512;
  if (!one_launch) {
    for (int k = 0; k < nconv; ++k)
      if (int rc = sr_conv3x3_f32(&d[k], stream_)) return rc;
    return SR_OK;
  }
  ChainParamsF P = {};
  P.nconv = nconv;
  int items = 0, progress = 0;
  for (int k = 0; k < nconv; ++k) {
    ConvParams& p = P.lv[k];
    if (int rc = fill_common(&d[k], &p, "sr_conv3x3_chain_f32")) return rc;
    p.H = p.vH = p.oH = d[k].in_h;
    p.W = p.vW = p.oW = d[k].in_w;
    p.tap_oy = p.tap_ox = -1;
    p.tiles_x = sr::cdiv(p.W, 32);
    p.tiles_y = p.H / 16;
    if (int rc = check_sizes(p, d[k].n, "sr_conv3x3_chain_f32")) return rc;
    const int gc = sr::conv_group_couts(d[k].cout);  // how the weight image was packed: 32- or 64-cout groups
    P.halves[k] = gc == 64 ? 2 : 1;
    if (gc == 64) {
      p.w_chunk = 9 * 2 * 256;
      p.w_tap = 2 * 256;
      p.w_cog = 256;
    }
    P.first[k] = items;
    P.need[k] = progress;
    if (k == 0) {
      P.tiles_x = p.tiles_x;
      P.tiles_y = p.tiles_y;
      P.ntiles = p.tiles_x * p.tiles_y * d[0].n;
    }
    items += P.ntiles * P.halves[k];
    progress += P.halves[k];
  }
  P.first[nconv] = items;

  P.abort = sync;
  P.head = sync + 1 + call_index;
  P.done = sync + 1 + kEpochs;
  // the progress words count finished items and are never reset between the calls that share the block: every call adds
  // `progress` per tile, so call i starts from i * progress (calls sharing a block have the same chain shape)
  P.epoch = call_index * progress;
  constexpr int lds = conv_lds_bytes<1, 4, 3, 4, 32>();
  auto kern = conv_chain_f32_kernel;
  if (int rc = sr::ensure_dynamic_lds((const void*)kern, lds)) return rc;
  static int slots[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (slots[dev] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    slots[dev] = (per_cu > 2 ? 2 : per_cu) * cus;
  }
  long long grid = slots[dev] / (conc > 1 ? conc : 1);
  if (grid > P.ntiles) grid = P.ntiles;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, P);
  SR_CHECK_LAUNCH("conv_chain_f32 launch");
  return SR_OK;
}

extern "C" int sr_conv3x3_f32(const sr_conv3x3_desc* d, void* stream_) {
  SR_CHECK_ARG(!d || !(d->out_unshuffle2 || d->res1_u2 || d->res1_keep_sign), "sr_conv3x3_f32: out_unshuffle2 / res1_u2 / res1_keep_sign are options of sr_conv3x3_bf16");
  hipStream_t stream = (hipStream_t)stream_;
  ConvParams p;
  int rc = fill_common(d, &p, "sr_conv3x3_f32");
  if (rc) return rc;
  p.H = p.vH = p.oH = d->upsample ? 2 * d->in_h : d->in_h;
  p.W = p.vW = p.oW = d->upsample ? 2 * d->in_w : d->in_w;
  p.src_shift = d->upsample ? 1 : 0;
  p.tap_oy = p.tap_ox = -1;
  const int gc = sr::conv_group_couts(d->cout);
  const int groups = ((d->cout + 31) / 32 * 32) / gc;
  constexpr int PT = 2;
  p.tiles_x = sr::cdiv(p.W, 32);
  p.tiles_y = sr::cdiv(p.H, 4 * PT);
  rc = check_sizes(p, d->n, "sr_conv3x3_f32");
  if (rc) return rc;
  if (d->out_nchw && !d->res1 && !d->res2) {
    // few couts, plain NCHW output: the 4x4x1 kernel (16-row tiles)
    constexpr int XB = (((16 + 2) * 34 * 32 + 1023) / 1024) * 1024, WB = ((9 * 4 * 32 + 1023) / 1024) * 1024;
    constexpr int lds = 2 * (XB + WB);
    auto kern = conv_fewcout_f32_kernel<3>;
    if (int rc = sr::ensure_dynamic_lds((const void*)kern, lds)) return rc;  // once per (kernel, device)
    p.tiles_y = sr::cdiv(p.H, 16);
    const bool prof = sr::prof_on();
    if (prof) {
      sr_launch_record r = {};
      r.kernel_id = 14;
      r.cin = d->cin_real > 0 ? d->cin_real : d->cin_pad;
      r.cout = d->cout;
      r.n = d->n;
      r.h = p.H;
      r.w = p.W;
      const double px = (double)d->n * p.H * p.W;
      r.flops = 2.0 * 9 * r.cin * r.cout * px;
      r.bytes = 4.0 * px * (r.cin + r.cout);
      sr::prof_begin(stream, r);
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_x * p.tiles_y * d->n), dim3(256), lds, stream, p);
    if (prof) sr::prof_end(stream);
    SR_CHECK_LAUNCH("conv_fewcout_f32 launch");
    return SR_OK;
  }
  if (d->out_nchw) return launch<1, PT, 3, true>(p, d->n, groups, stream, d);
  // Small inputs (single plate crops, 32x32 training patches): with 8-row tiles the launch has fewer workgroups than the
  // chip has CUs and every layer is a serial walk over K on a few CUs; 4-row tiles double the workgroups (more halo and
  // weight refill per MFMA, irrelevant while CUs idle).
  if (int rs = launch_small<3>(p, d, groups, gc, stream); rs >= 0) return rs;
  if ((long long)p.tiles_x * p.tiles_y * d->n * groups < 256 && p.H > 4) {
    p.tiles_y = sr::cdiv(p.H, 4);
    return gc == 64 ? launch<2, 1, 3, false>(p, d->n, groups, stream, d) : launch<1, 1, 3, false>(p, d->n, groups, stream, d);
  }
  if (gc == 64) {
    // 64-cout groups on 16-row tiles (PT = 4: 128 accumulator registers): half the weight refill per MFMA and one round of
    // workgroups instead of two, when the launch still fills the chip (A/B: sr_set_f32_tall64)
    const long long wg16 = (long long)p.tiles_x * sr::cdiv(p.H, 16) * d->n * groups;
    if (g_f32_tall64 && p.H % 16 == 0 && wg16 >= 512) {
      p.tiles_y = sr::cdiv(p.H, 16);
      return launch<2, 4, 3, false>(p, d->n, groups, stream, d);
    }
    return launch<2, PT, 3, false>(p, d->n, groups, stream, d);
  }
  // 32-cout groups: 16-row tiles (PT = 4) halve the weight refill per MFMA (measured +2.5..7 % on the RDB conv1-4
  // shapes, tools/conv_ablate.hip) as long as the launch still has >= 2 workgroups per CU and rows are not wasted.
  const long long wg4 = (long long)p.tiles_x * sr::cdiv(p.H, 16) * d->n * groups;
  if (p.H % 16 == 0 && wg4 >= 512 && !g_f32_rows8) {
    p.tiles_y = sr::cdiv(p.H, 16);
    return launch<1, 4, 3, false>(p, d->n, groups, stream, d);
  }
  if (g_f32_rows8 == 2) {
    p.tiles_y = sr::cdiv(p.H, 4);
    return launch<1, 1, 3, false>(p, d->n, groups, stream, d);
  }
  return launch<1, PT, 3, false>(p, d->n, groups, stream, d);
}

// 4x4 / stride 2 / pad 1 convolution (VGGStyleDiscriminator128 conv*_1, discriminator_arch.py:22-43; UNetDiscriminatorSN
// conv1-3) as four accumulating parity passes:  Y[y][x] = sum_{ry,rx} sum_{ty,tx} W[2ty+1-ry... see pack] X[2(y+qy)+ry][..]
// Each pass is a 2x2-tap stride-1 conv over the parity sub-image (ry,rx) of X: qy = ty + (ry ? -1 : 0), dy = 2*qy + 1 + ry.
extern "C" int sr_conv4x4s2_f32(const sr_conv3x3_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvParams p;
  int rc = fill_common(d, &p, "sr_conv4x4s2_f32");
  if (rc) return rc;
  SR_CHECK_ARG(!d->out_nchw && !d->upsample && !d->mask_src && !d->accumulate && d->in_h >= 2 && d->in_w >= 2,
               "sr_conv4x4s2_f32: unsupported option");
  p.H = p.oH = (d->in_h - 2) / 2 + 1;
  p.W = p.oW = (d->in_w - 2) / 2 + 1;
  const int gc = sr::conv_group_couts(d->cout);
  const int groups = ((d->cout + 31) / 32 * 32) / gc;
  constexpr int PT = 2;
  p.tiles_x = sr::cdiv(p.W, 32);
  p.tiles_y = sr::cdiv(p.H, 4 * PT);
  rc = check_sizes(p, d->n, "sr_conv4x4s2_f32");
  if (rc) return rc;
  const size_t pass_floats = (size_t)groups * p.cin_blocks * 4 * gc * 8;
  for (int pass = 0; pass < 4; ++pass) {
    const int ry = pass >> 1, rx = pass & 1;
    ConvParams q = p;
    q.w = d->wpacked + pass * pass_floats;
    q.src_mul = 2;
    q.src_oy = ry;
    q.src_ox = rx;
    q.vH = (d->in_h - ry + 1) / 2;
    q.vW = (d->in_w - rx + 1) / 2;
    q.tap_oy = ry ? -1 : 0;
    q.tap_ox = rx ? -1 : 0;
    q.accumulate = pass > 0;
    q.acc_first = 1;
    if (pass < 3) {  // bias, activation, scale and residuals belong to the completed sum
      q.bias = nullptr;
      q.slope = 1.f;
      q.alpha = 1.f;
      q.res1 = q.res2 = nullptr;
    }
    rc = launch_small<2>(q, d, groups, gc, stream);
    if (rc < 0) rc = gc == 64 ? launch<2, PT, 2, false>(q, d->n, groups, stream, d) : launch<1, PT, 2, false>(q, d->n, groups, stream, d);
    if (rc) return rc;
  }
  return SR_OK;
}

// Data gradient of the 4x4/s2 conv: dX[2a+py][2b+px] = sum over the two taps per axis of matching parity.
// d->in = dY (in_h x in_w), d->out = dX (out_h x out_w, required), d->wpacked from sr_conv4x4s2_pack_f32(mode 1),
// d->cout = channels of dX.
extern "C" int sr_conv4x4s2_dgrad_f32(const sr_conv3x3_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvParams p;
  int rc = fill_common(d, &p, "sr_conv4x4s2_dgrad_f32");
  if (rc) return rc;
  SR_CHECK_ARG(!d->out_nchw && !d->upsample && d->out_h > 0 && d->out_w > 0, "sr_conv4x4s2_dgrad_f32: out_h/out_w required");
  SR_CHECK_ARG((d->out_h - 2) / 2 + 1 == d->in_h && (d->out_w - 2) / 2 + 1 == d->in_w,
               "sr_conv4x4s2_dgrad_f32: dY %dx%d does not match dX %dx%d", d->in_h, d->in_w, d->out_h, d->out_w);
  p.oH = d->out_h;
  p.oW = d->out_w;
  p.vH = d->in_h;
  p.vW = d->in_w;
  const int gc = sr::conv_group_couts(d->cout);
  const int groups = ((d->cout + 31) / 32 * 32) / gc;
  constexpr int PT = 2;
  const size_t pass_floats = (size_t)groups * p.cin_blocks * 4 * gc * 8;
  for (int pass = 0; pass < 4; ++pass) {
    const int py = pass >> 1, px = pass & 1;
    ConvParams q = p;
    q.w = d->wpacked + pass * pass_floats;
    q.H = (d->out_h - py + 1) / 2;
    q.W = (d->out_w - px + 1) / 2;
    q.tiles_x = sr::cdiv(q.W, 32);
    q.tiles_y = sr::cdiv(q.H, 4 * PT);
    q.dst_mul = 2;
    q.dst_oy = py;
    q.dst_ox = px;
    q.tap_oy = py ? 0 : -1;
    q.tap_ox = px ? 0 : -1;
    rc = check_sizes(q, d->n, "sr_conv4x4s2_dgrad_f32");
    if (rc) return rc;
    rc = launch_small<2>(q, d, groups, gc, stream);
    if (rc < 0) rc = gc == 64 ? launch<2, PT, 2, false>(q, d->n, groups, stream, d) : launch<1, PT, 2, false>(q, d->n, groups, stream, d);
    if (rc) return rc;
  }
  return SR_OK;
}

extern "C" const char* sr_kernel_name(int id) {
  static const char* names[8] = {
      "conv_f32_kernelILi1ELi2ELi3ELb0E", "conv_f32_kernelILi1ELi2ELi3ELb1E", "conv_f32_kernelILi1ELi2ELi2ELb0E",
      "conv_f32_kernelILi1ELi2ELi2ELb1E", "conv_f32_kernelILi2ELi2ELi3ELb0E", "conv_f32_kernelILi2ELi2ELi3ELb1E",
      "conv_f32_kernelILi2ELi2ELi2ELb0E", "conv_f32_kernelILi2ELi2ELi2ELb1E"};
  static const char* wnames[5] = {"wgrad3x3_f32_kernelILi1ELi1ELi1E", "wgrad3x3_f32_kernelILi1ELi2ELi1E",
                                  "wgrad3x3_f32_kernelILi1ELi4ELi1E", "wgrad3x3_f32_kernelILi2ELi2ELi1E",
                                  "wgrad3x3_f32_kernelILi2ELi1ELi1E"};
  if (id >= 8 && id < 13) return wnames[id - 8];
  if (id == 13) return "conv_f32_kernelILi1ELi4ELi3ELb0E";
  if (id == 14) return "conv_fewcout_f32_kernelILi3E";
  if (id >= 16 && id < 32) {  // conv_bf16.hip: 16 + (COT-1)*2 + (PT==4) + 4*NCHW_OUT + 8*(NW==8)
    static const char* hnames[16] = {
        "conv_bf16_kernelILi1ELi2ELi4ELb0E", "conv_bf16_kernelILi1ELi4ELi4ELb0E", "conv_bf16_kernelILi2ELi2ELi4ELb0E",
        "conv_bf16_kernelILi2ELi4ELi4ELb0E", "conv_bf16_kernelILi1ELi2ELi4ELb1E", "conv_bf16_kernelILi1ELi4ELi4ELb1E",
        "conv_bf16_kernelILi2ELi2ELi4ELb1E", "conv_bf16_kernelILi2ELi4ELi4ELb1E", "conv_bf16_kernelILi1ELi2ELi8ELb0E",
        "conv_bf16_kernelILi1ELi4ELi8ELb0E", "conv_bf16_kernelILi2ELi2ELi8ELb0E", "conv_bf16_kernelILi2ELi4ELi8ELb0E",
        "conv_bf16_kernelILi1ELi2ELi8ELb1E", "conv_bf16_kernelILi1ELi4ELi8ELb1E", "conv_bf16_kernelILi2ELi2ELi8ELb1E",
        "conv_bf16_kernelILi2ELi4ELi8ELb1E"};
    return hnames[id - 16];
  }
  if (id >= 32 && id < 40) {  // wgrad_bf16.hip <CT, IT, KS, R, NSTG>
    static const char* gnames[8] = {"wgrad_bf16_kernelILi1ELi1ELi8ELi2ELi4E", "wgrad_bf16_kernelILi1ELi2ELi4ELi2ELi4E",
                                    "wgrad_bf16_kernelILi1ELi4ELi2ELi1ELi4E", "wgrad_bf16_kernelILi2ELi1ELi4ELi2ELi4E",
                                    "wgrad_bf16_kernelILi2ELi2ELi2ELi2ELi3E", "wgrad_bf16_kernelILi2ELi4ELi1ELi1ELi4E",
                                    "wgrad_bf16_kernelILi1ELi3ELi2ELi1ELi4E", "wgrad_bf16_kernelILi1ELi5ELi1ELi1ELi4E"};
    return gnames[id - 32];
  }
  if (id >= 44 && id < 48) {  // small-map variants: WT 16 / 8, KS 3 / 2 (PT 1 or 2 share an id)
    static const char* snames[4] = {"conv_f32_kernel<2,*,3,false,0,4,16>", "conv_f32_kernel<2,*,3,false,0,4,8>", "conv_f32_kernel<2,*,2,false,0,4,16>",
                                    "conv_f32_kernel<2,*,2,false,0,4,8>"};
    return snames[id - 44];
  }
  if (id == 40) return "wgrad_rdb_bf16_kernel";
  if (id == 48) return "wgrad_f32_rdb_kernel";
  if (id == 64) return "conv_stream_bf16_kernel";
  if (id == 60) return "rdb_fused_bf16_kernelILi0E";
  if (id == 61) return "rdb_fused_bf16_kernelILi1E";
  if (id == 62) return "rdb_fused_bf16_kernelILi2E";
  if (id == 50) return "conv_fewcout_bf16_kernel";
  if (id == 42) return "conv_bf16_kernelILi1ELi1ELi4ELb0E";
  if (id == 43) return "conv_bf16_kernelILi2ELi1ELi4ELb0E";
  if (id == 15) return "conv_f32_kernelILi1ELi1ELi3ELb0E";
  if (id == 41) return "conv_f32_kernelILi2ELi1ELi3ELb0E";
  return (id >= 0 && id < 8) ? names[id] : "";
}
