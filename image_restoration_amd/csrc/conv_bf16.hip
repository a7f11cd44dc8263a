// bf16 3x3 convolution on v_mfma_f32_32x32x16_bf16 (gfx950) — inference path of the generator.
//
// The reference has no reduced precision at all; BASELINE configs 3-4 name bf16, so this is a build
// extension whose parity is declared against the fp32 oracle with its own tolerance (tests/test_bf16_gpu.py).
//
// Layout CB16: __bf16 feat[N][C/16][H][W][16] — one pixel of one 16-channel block is again 32 bytes, so the LDS
// images, the LDS-DMA staging and every address of the fp32 kernel (conv_f32.hip) carry over byte for byte:
//   X tile [TH+2][34][32 B], W image [9][32*COT][32 B], chunk = one 16-channel block.
// One MFMA (K = 16) consumes a whole chunk of one tap: lane (pixel j, half h) reads 16 B = channels 8h..8h+7 as the B
// operand, lane (cout i, half h) the matching 16 B of the weight image as the A operand; accumulators are fp32 and
// have the fp32 kernel's layout (pixel on the lane, 4 consecutive couts per register quad), so the epilogue is the
// same with an 8-byte bf16x4 store.  Per chunk a wave issues 9*(COT+PT) ds_read_b128 for 9*COT*PT MFMAs of 32
// cycles: with COT = 2, PT = 4 that is 0.75 reads per MFMA (LDS limit: 2).  The kernel is fed by L2->LDS traffic:
// 37.6 KB per 9.4 MFLOP chunk of a 16x32x64 tile = 251 FLOP/B.
//
// LDS bank swizzle.  A lane reads 16 B of a 32-byte pixel, so the 32 lanes of one half (same h) stride 32 B and a
// ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}) would hit every 16-byte bank slot twice: PMC of the
// unswizzled kernel showed SQ_LDS_BANK_CONFLICT = exactly half of SQ_LDS_IDX_ACTIVE, and with 0.6 reads per MFMA the LDS
// array (7.8 us per conv1-4 launch per CU) was busier than the matrix pipe (6.7 us).  Both LDS images therefore store the
// two 16-byte halves of pixel / cout number c in swapped order when bit 3 of c is set: lanes 8 apart then fall on
// different slots and every group covers all 64 banks.  The swap is applied by the LDS-DMA source address (the lane that
// fills unit q fetches the half that belongs there), so global reads stay whole 32-byte pixels.
#include <cstring>

#include "sr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {


struct ConvParamsH {
  const void* zero;  // sr::zero_line()
  const char* in;     // CB16 bf16
  const char* w;      // packed bf16 image [group][cin/16][tap][32*COT][16]
  const float* bias;  // fp32 [cout pad 32] or null
  char* out;          // CB16 bf16, or NCHW fp32 when NCHW_OUT
  const char* res1;
  const char* res2;
  const char* mask;  // CB16 forward activation whose sign gates the output (LeakyReLU backward), first mask_cbn blocks
  long long in_nb, out_nb, res1_nb, res2_nb, mask_nb;  // image strides in BYTES
  int cin_blocks;   // Cin / 16
  int cout_blocks;  // valid 16-channel blocks of the destination (ceil(cout/16))
  int cout;
  int in_h, in_w, H, W, tiles_x, tiles_y;
  int cogs;         // cout groups (32*COT couts each)
  int s2_cpb;       // S2 kernels: 16-channel blocks per parity class of the unshuffled operand
  int s2_side;      // S2 kernels: 0 = parity of the input chunk decides the live taps, 1 = parity of the cout tile
  int src_shift;    // 1: nearest x2 upsample on the fly
  int mask_cbn;
  int out_u2;       // 1: the CB16 destination is written pixel-unshuffled (sr_conv3x3_desc.out_unshuffle2)
  int res1_mode;    // bit 0: res1 is read pixel-unshuffled (res1_u2); bit 1: sign-keeping rounding of act + res1 (res1_keep_sign)
  float slope, alpha, beta1, beta2, mask_slope;
  long long* dbg;  // development: per-workgroup phase clocks
};

__device__ __forceinline__ void glds16h(const void* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// S2: the weights carry a 4x4/s2 convolution embedded in the 3x3 grid of a pixel-unshuffled operand: a parity class
// (ry, rx) of channels has non-zero weights on 2x2 of the 9 taps only — rows {1-ry, 2-ry}, columns {1-rx, 2-rx} of the
// forward image (input-channel parity), rows {ry, ry+1}, columns {rx, rx+1} of the data-gradient image (output-channel
// parity) — and the loop issues those 4 taps (4/9 of the LDS reads and MFMAs; the skipped products are exact zeros).
// 16-byte store that writes through to memory (sc1): how a tile that another workgroup of the SAME launch will read is published
// (MI355X_MICROARCH.md, inter-workgroup visibility: write-through stores + every wave's vmcnt(0) + barrier + flag; no L2 write-back).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(void* ptr, u32x4_t v, bool write_through) {
  if (write_through)
    // + two wait states: a 16-byte store reads its data registers after it has issued, and behind an asm the compiler does not
    // know that the next vector instruction must not overwrite them yet (the VMEM store-data hazard)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(ptr), "v"(v) : "memory");
  else
    *(__attribute__((address_space(1))) u32x4_t*)ptr = v;  // global, also where the pointer's origin is opaque (conv_params)
}

// CB16 epilogue of a tile: bias, LeakyReLU, scale, residual scale-adds, LeakyReLU-backward mask in fp32, then bf16 stores.
// A lane holds 4 consecutive couts per register quad and its partner lane (same pixel, other half h) the next 4, so 8-byte accesses
// would touch half of every 32-byte pixel: v_permlane32_swap exchanges quads between the two lane halves so that every lane
// loads / stores 16 B = 8 consecutive channels and a wave instruction covers 1 KB of contiguous memory (full 128-byte lines).
// Two passes, so that no load waits behind a store: the vector-memory counter retires in order, and the first form of this epilogue
// (bias / residual load -> use -> store, block by block) made every load wait for the previous block's stores — write-through
// ones in the chain kernel: 6-16 thousand cycles per tile.  Pass 1 loads the bias values once, then finishes every value in place
// in the accumulators (its residual / mask loads only queue behind other loads); pass 2 converts and stores.  Per element the
// operations and their order are unchanged (results are bit-identical).
template <int COT, int PT, bool WT_OUT, bool BIAS_LDS = false>
__device__ __forceinline__ void epilogue_cb16(const ConvParamsH& p, f32x16 (&acc)[COT][PT], const int cog, const int n, const int x,
                                              const int y_first, const int h,
                                              const __attribute__((address_space(3))) float* lbias = nullptr) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto swap_halves = [](u32x4& t) {  // t[0..1].upper-lanes <-> t[2..3].lower-lanes
    auto r0 = __builtin_amdgcn_permlane32_swap(t[0], t[2], false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(t[1], t[3], false, false);
    t = u32x4{r0[0], r1[0], r0[1], r1[1]};
  };
  auto bf2f = [](unsigned w, int hi) { return __builtin_bit_cast(float, hi ? (w & 0xffff0000u) : (w << 16)); };
  auto f2bf2 = [](float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 t = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, t);
  };
  const long long HW = (long long)p.H * p.W;
  f32x4 bias[COT][4];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if constexpr (BIAS_LDS)  // the caller keeps the packed bias in the LDS (no vector-memory load behind its LDS-DMA queue)
        bias[c][g] = *(const __attribute__((address_space(3))) f32x4*)(lbias + (cog * COT + c) * 32 + g * 8 + h * 4);
      else
        bias[c][g] = p.bias ? *(const f32x4*)(p.bias + (cog * COT + c) * 32 + g * 8 + h * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  // pass 1
#pragma unroll
  for (int r = 0; r < PT; ++r) {
    const int y = y_first + r;
    if (y >= p.H || x >= p.W) continue;
    const long long pixoff = (long long)y * p.W + x;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {  // the two 16-channel blocks of this 32-cout tile
        const int cb = (cog * COT + c) * 2 + m;
        if (cb >= p.cout_blocks) continue;
        const long long off = ((cb * HW + pixoff) * 16 + h * 8) * 2;  // bytes: this lane's 8 channels of the pixel
        f32x4 v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int g = 2 * m + q;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[q][e] = acc[c][r][g * 4 + e];
          if (p.bias) v[q] += bias[c][g];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[q][e] = v[q][e] > 0.f ? v[q][e] : v[q][e] * p.slope;
          v[q] *= p.alpha;
        }
        auto add_res = [&](const char* res, long long nb, float beta) {
          u32x4 rr = *(const __attribute__((address_space(1))) u32x4*)(res + (long long)n * nb + off);  // channels 8h..8h+7
          swap_halves(rr);  // -> rr[0..1] = quad 2m, rr[2..3] = quad 2m+1 of this lane
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[q][e] += beta * bf2f(rr[q * 2 + (e >> 1)], e & 1);
        };
        if (p.res1 && p.res1_mode) {
          // a skip connection that only exists pixel-unshuffled, added so that the stored sum still tells the activation's sign
          // (sr_conv3x3_desc.res1_u2 / res1_keep_sign): the value is finished here, already rounded to bf16 (pass 2 converts exactly)
          long long roff = off;
          if (p.res1_mode & 1)
            roff = (((long long)((((y & 1) << 1) | (x & 1)) * p.cout_blocks + cb) * (HW >> 2) + (long long)(y >> 1) * (p.W >> 1) + (x >> 1)) * 16 + h * 8) * 2;
          u32x4 rr = *(const __attribute__((address_space(1))) u32x4*)(p.res1 + (long long)n * p.res1_nb + roff);
          swap_halves(rr);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned w16 = (e & 1) ? (rr[q * 2 + (e >> 1)] >> 16) : (rr[q * 2 + (e >> 1)] & 0xffffu);
              const float act = v[q][e];
              float sum = act + p.beta1 * __builtin_bit_cast(float, w16 << 16);
              if (p.res1_mode & 2) {
                const __bf16 rb = (__bf16)sum;
                unsigned sb = __builtin_bit_cast(unsigned short, rb);
                if (act > 0.f && sb == w16) sb = (w16 & 0x7fffu) == 0 ? 1u : ((w16 & 0x8000u) ? w16 - 1 : w16 + 1);
                sum = __builtin_bit_cast(float, sb << 16);
              }
              v[q][e] = sum;
            }
        } else if (p.res1) {
          add_res(p.res1, p.res1_nb, p.beta1);
        }
        if (p.res2) add_res(p.res2, p.res2_nb, p.beta2);
        if (p.mask && cb < p.mask_cbn) {
          u32x4 mm = *(const __attribute__((address_space(1))) u32x4*)(p.mask + (long long)n * p.mask_nb + off);
          swap_halves(mm);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (!(bf2f(mm[q * 2 + (e >> 1)], e & 1) > 0.f)) v[q][e] *= p.mask_slope;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[c][r][(2 * m + q) * 4 + e] = v[q][e];
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // pass 2
#pragma unroll
  for (int r = 0; r < PT; ++r) {
    const int y = y_first + r;
    if (y >= p.H || x >= p.W) continue;
    const long long pixoff = (long long)y * p.W + x;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int cb = (cog * COT + c) * 2 + m;
        if (cb >= p.cout_blocks) continue;
        long long off = ((cb * HW + pixoff) * 16 + h * 8) * 2;
        if (p.out_u2)  // plane (2 ry + rx) CB + cb of the half-resolution tensor (sr_conv3x3_desc.out_unshuffle2)
          off = (((long long)((((y & 1) << 1) | (x & 1)) * p.cout_blocks + cb) * (HW >> 2) + (long long)(y >> 1) * (p.W >> 1) + (x >> 1)) * 16 + h * 8) * 2;
        const int b = m * 8;
        u32x4 o = {f2bf2(acc[c][r][b + 0], acc[c][r][b + 1]), f2bf2(acc[c][r][b + 2], acc[c][r][b + 3]),
                   f2bf2(acc[c][r][b + 4], acc[c][r][b + 5]), f2bf2(acc[c][r][b + 6], acc[c][r][b + 7])};
        swap_halves(o);
        store16(p.out + (long long)n * p.out_nb + off, o, WT_OUT);
      }
    }
  }
}

// One output tile (TH rows x 32 columns x 32*COT couts) of one conv: the whole body of conv_bf16_kernel, also run once per work
// item by the persistent chain kernel below.  WT_OUT: publish the tile with write-through stores.
template <int COT, int PT, int NW, bool NCHW_OUT, bool S2, bool WT_OUT>
__device__ __forceinline__ void conv_tile_h(const ConvParamsH p, const int cog, const int tx, const int ty, const int n, char* smem) {
  constexpr int TH = NW * PT, XROW = 34, XPIX = (TH + 2) * XROW;
  constexpr int XBYTES = ((XPIX * 32 + 1023) / 1024) * 1024;
  constexpr int NXU = XBYTES / 1024, NWU = 9 * COT;
  constexpr int WBYTES = NWU * 1024, STAGE = XBYTES + WBYTES;
  constexpr int NXR = (NXU + NW - 1) / NW, NWR = (NWU + NW - 1) / NW;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int x0 = tx * 32, y0 = ty * TH;
  const long long plane_b = (long long)p.in_h * p.in_w * 32;  // bytes of one 16-channel block plane
  const char* in_n = p.in + (long long)n * p.in_nb;
  const char* wg = p.w + (size_t)cog * p.cin_blocks * WBYTES;

  int xoff[NXR];  // byte offset inside a block plane; -1 = zero padding
#pragma unroll
  for (int r = 0; r < NXR; ++r) {
    const int u = r * NW + wave;
    const int q = u * 64 + lane;
    const int pix = q >> 1, half = q & 1;
    const int row = pix / XROW, col = pix - row * XROW;
    const int gy = y0 - 1 + row, gx = x0 - 1 + col;
    const bool valid = (pix < XPIX) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const int sy = gy >> p.src_shift, sx = gx >> p.src_shift;
    xoff[r] = valid ? ((sy * p.in_w + sx) * 32 + (half ^ ((col >> 3) & 1)) * 16) : -1;  // bank swizzle (header)
  }

  // LDS-DMA through buffer descriptors (see conv_f32.hip): base in scalar registers, one 32-bit per-lane byte offset computed once
  // per tile, the chunk as the scalar soffset; padding lanes carry an offset beyond num_records and read as zeros.
  const __amdgpu_buffer_rsrc_t x_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)in_n, 0, (unsigned)((long long)p.cin_blocks * plane_b), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, (unsigned)((long long)p.cin_blocks * WBYTES), 0x00020000);
  unsigned xvo[NXR];
#pragma unroll
  for (int r = 0; r < NXR; ++r) xvo[r] = xoff[r] >= 0 ? (unsigned)xoff[r] : 0xfffffff0u;
  const unsigned wvo = (lane ^ ((lane >> 4) & 1)) * 16;  // unit (cout i, half) <- half ^ bit3(i)
  auto stage = [&](int buf, int cb) {
    char* xs = smem + buf * STAGE;
    char* ws = xs + XBYTES;
    const unsigned xso = (unsigned)cb * (unsigned)plane_b, wso = (unsigned)cb * (unsigned)WBYTES;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int u = r * NW + wave;
      if (u < NXU)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (__attribute__((address_space(3))) void*)(xs + u * 1024), 16, xvo[r], xso, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
      const int u = r * NW + wave;
      if (u < NWU)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (__attribute__((address_space(3))) void*)(ws + u * 1024), 16, wvo,
                                                 wso + (unsigned)u * 1024u, 0, 0);
    }
  };

  f32x16 acc[COT][PT];
#pragma unroll
  for (int a = 0; a < COT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // byte offset of this lane's B operand at column offset dxa (0..3): pixel column j + dxa, swizzled half
  const int xrow0 = ((wave * PT) * XROW + j) * 32;
  auto xl = [&](int dxa) { return xrow0 + dxa * 32 + ((h ^ (((j + dxa) >> 3) & 1)) * 16); };
  const int xlane0 = xl(0), xlane1 = xl(1), xlane2 = xl(2);
  const int wlane = j * 32 + ((h ^ ((j >> 3) & 1)) * 16);

  auto compute = [&](int buf, int dy0, int dx0) {
    if constexpr (S2) {
      const char* xb = smem + buf * STAGE + dy0 * XROW * 32;
      const char* ws = smem + buf * STAGE + XBYTES + wlane + (dy0 * 3 + dx0) * COT * 1024;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        bf16x8 bx[PT + 1], a[2][COT];
        const int dxa = dx0 + dx;  // 0..2
        const char* xs = xb + (dxa == 0 ? xlane0 : dxa == 1 ? xlane1 : xlane2);
#pragma unroll
        for (int r = 0; r < PT + 1; ++r) bx[r] = *(const bf16x8*)(xs + r * XROW * 32);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int c = 0; c < COT; ++c) a[dy][c] = *(const bf16x8*)(ws + ((dy * 3 + dx) * COT + c) * 1024);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int c = 0; c < COT; ++c)
#pragma unroll
            for (int r = 0; r < PT; ++r)
              acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[dy][c], bx[r + dy], acc[c][r], 0, 0, 0);
      }
      return;
    }
    const char* xb = smem + buf * STAGE;
    const char* ws = smem + buf * STAGE + XBYTES + wlane;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      // the PT+2 tile rows of this column offset serve all three dy: 3*(PT+2) + 9*COT reads per chunk, not 9*(PT+COT)
      bf16x8 bx[PT + 2], a[3][COT];
      const char* xs = xb + (dx == 0 ? xlane0 : dx == 1 ? xlane1 : xlane2);
#pragma unroll
      for (int r = 0; r < PT + 2; ++r) bx[r] = *(const bf16x8*)(xs + r * XROW * 32);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int c = 0; c < COT; ++c) a[dy][c] = *(const bf16x8*)(ws + ((dy * 3 + dx) * COT + c) * 1024);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
          for (int r = 0; r < PT; ++r)
            acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[dy][c], bx[r + dy], acc[c][r], 0, 0, 0);
    }
  };
  // S2: first live tap row / column of a parity class
  auto s2_taps = [&](int chunk, int& dy0, int& dx0) {
    const int par = (p.s2_side ? cog * (COT * 2) : chunk) / p.s2_cpb;
    const int ry = par >> 1, rx = par & 1;
    dy0 = p.s2_side ? ry : 1 - ry;
    dx0 = p.s2_side ? rx : 1 - rx;
  };

  const int nchunk = p.cin_blocks;
  long long tk[6] = {0, 0, 0, 0, 0, 0};
  if (p.dbg) tk[0] = __builtin_readcyclecounter();
  stage(0, 0);
  __syncthreads();
  if (p.dbg) tk[1] = __builtin_readcyclecounter();
  for (int c = 0; c < nchunk; ++c) {
    long long t0 = 0, t1 = 0;
    if (p.dbg) t0 = __builtin_readcyclecounter();
    if (c + 1 < nchunk) stage((c + 1) & 1, c + 1);
    int dy0 = 0, dx0 = 0;
    if constexpr (S2) s2_taps(c, dy0, dx0);
    compute(c & 1, dy0, dx0);
    if (p.dbg) t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (p.dbg) {
      tk[2] += t1 - t0;
      tk[3] += __builtin_readcyclecounter() - t1;
    }
  }
  if (p.dbg) tk[4] = __builtin_readcyclecounter();

  const int x = x0 + j;
  if constexpr (!NCHW_OUT) {
    epilogue_cb16<COT, PT, WT_OUT>(p, acc, cog, n, x, y0 + wave * PT, h);
  } else {  // fp32 NCHW destination (conv_last): 4-byte stores per channel plane
    const long long HW = (long long)p.H * p.W;
    f32x4 bias[COT][4];
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co0 = (cog * COT + c) * 32 + g * 8 + h * 4;
        bias[c][g] = (p.bias && (co0 >> 4) < p.cout_blocks) ? *(const f32x4*)(p.bias + co0) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int r = 0; r < PT; ++r) {
      const int y = y0 + wave * PT + r;
      if (y >= p.H || x >= p.W) continue;
      const long long pixoff = (long long)y * p.W + x;
#pragma unroll
      for (int c = 0; c < COT; ++c) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co0 = (cog * COT + c) * 32 + g * 8 + h * 4;  // first of this lane's 4 couts
          if ((co0 >> 4) >= p.cout_blocks) continue;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[c][r][g * 4 + e];
          if (p.bias) v += bias[c][g];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
          v *= p.alpha;
          float* on = (float*)(p.out + (long long)n * p.out_nb) + pixoff;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co0 + e < p.cout) on[(co0 + e) * HW] = v[e];
        }
      }
    }
  }
  if (p.dbg && lane == 0) {
    tk[5] = __builtin_readcyclecounter();
    long long* o = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
    o[0] = tk[0]; o[1] = tk[1] - tk[0]; o[2] = tk[2]; o[3] = tk[3]; o[4] = tk[4] - tk[0]; o[5] = tk[5] - tk[4];
    o[6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
  }
}

template <int COT, int PT, int NW, bool NCHW_OUT, bool S2 = false>
__global__ __launch_bounds__(NW * 64) void conv_bf16_kernel(const ConvParamsH p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so each XCD gets a contiguous run of the
  // (tile, cout group) sequence with the cout group fastest: the cout groups of one pixel tile run side by side on one
  // XCD and share the tile's input through that XCD's L2 (a deep layer with 4-8 cout groups otherwise re-reads its whole
  // input from HBM once per group), and neighbouring tiles share halo rows and weights.
  int t;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int cog = t % p.cogs;
  t /= p.cogs;
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  conv_tile_h<COT, PT, NW, NCHW_OUT, S2, false>(p, cog, tx, ty, n, smem);
}

// ------------------------------------------------------------------------------------------------ few couts
// conv_last (Cout = 3, rrdbnet_arch.py:118) and the U-Net discriminator's logit conv (Cout = 1): on the 32-cout tile 29-31 of 32
// MFMA rows multiply zero weights (203 us per batch of 16 512x512 images).  Here one v_mfma_f32_4x4x4_16B_bf16 serves 64 pixels x
// 4 couts x 4 channels: lane = pixel (two tile rows of 32 pixels per wave instruction), its B operand 8 bytes of its own pixel,
// its A operand the 8 bytes of cout (lane & 3) — the twin of conv_fewcout_f32_kernel.  The weights come out of the ordinary
// 32-cout image (rows 0..3 of each tap), so nothing is packed differently.  HBM-bound: 32 B read per pixel and chunk, 4 B written
// per pixel and cout.
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void conv_fewcout_bf16_kernel(const ConvParamsH p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TH = 16, XROW = 34, XPIX = (TH + 2) * XROW;
  constexpr int XBYTES = ((XPIX * 32 + 1023) / 1024) * 1024, NXU = XBYTES / 1024;
  constexpr int WBYTES = 2048, STAGE = XBYTES + WBYTES;  // 9 taps x 4 couts x 32 B = 1152 B
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
  const long long plane_b = (long long)p.in_h * p.in_w * 32;
  const char* in_n = p.in + (long long)n * p.in_nb;

  int xoff[NXR];
#pragma unroll
  for (int r = 0; r < NXR; ++r) {
    const int q = (r * 4 + wave) * 64 + lane;
    const int pix = q >> 1, half = q & 1;
    const int row = pix / XROW, col = pix - row * XROW;
    const int gy = y0 - 1 + row, gx = x0 - 1 + col;
    const bool valid = (pix < XPIX) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    // LDS bank swizzle (as conv_tile_h): halves of staged pixel column c swapped when bit 3 of c is set; the readers undo it
    xoff[r] = valid ? (((gy >> p.src_shift) * p.in_w + (gx >> p.src_shift)) * 32 + (half ^ ((col >> 3) & 1)) * 16) : -1;
  }
  // weight piece q of the chunk: tap = q / 8, cout = (q % 8) / 2, half = q % 2  ->  packed image [tap][32 couts][32 B]
  const int wq = wave * 64 + lane;
  const int woff = (wave < 2 && wq < 72) ? ((wq >> 3) * 32 + ((wq & 7) >> 1)) * 32 + (wq & 1) * 16 : -1;

  auto stage = [&](int buf, int cb) {
    char* xs = smem + buf * STAGE;
    const char* plane = in_n + (size_t)cb * plane_b;
#pragma unroll
    for (int r = 0; r < NXR; ++r) {
      const int u = r * 4 + wave;
      if (u < NXU) glds16h(xoff[r] >= 0 ? (const void*)(plane + xoff[r]) : p.zero, xs + u * 1024);
    }
    if (wave < 2) {  // 72 pieces of 16 B
      const char* wchunk = p.w + (size_t)cb * (9 * 32 * 32);
      glds16h(woff >= 0 ? (const void*)(wchunk + woff) : p.zero, xs + XBYTES + wave * 1024);
    }
  };

  f32x4 acc[2];
  acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int prow = lane >> 5, pcol = lane & 31;
  const int xlane = ((wave * 4 + prow) * XROW + pcol) * 32;  // this lane's pixel, row group 0, tap (0, 0)
  const int wlane = (lane & 3) * 32;                         // this lane's cout row of the weight chunk

  auto compute = [&](int buf) {
    const char* xs = smem + buf * STAGE + xlane;
    const char* ws = smem + buf * STAGE + XBYTES + wlane;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int tap = dy * 3 + dx;
        const bf16x8 w0 = *(const bf16x8*)(ws + tap * 128), w1 = *(const bf16x8*)(ws + tap * 128 + 16);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const char* xp = xs + ((2 * g + dy) * XROW + dx) * 32;
          const int sw = (((pcol + dx) >> 3) & 1) * 16;
          const bf16x8 a0 = *(const bf16x8*)(xp + sw), a1 = *(const bf16x8*)(xp + (sw ^ 16));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 W0 = __builtin_bit_cast(s16x8, w0), W1 = __builtin_bit_cast(s16x8, w1);
          const s16x8 A0 = __builtin_bit_cast(s16x8, a0), A1 = __builtin_bit_cast(s16x8, a1);
          acc[g] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(s16x4{W0[0], W0[1], W0[2], W0[3]}, s16x4{A0[0], A0[1], A0[2], A0[3]}, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(s16x4{W0[4], W0[5], W0[6], W0[7]}, s16x4{A0[4], A0[5], A0[6], A0[7]}, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(s16x4{W1[0], W1[1], W1[2], W1[3]}, s16x4{A1[0], A1[1], A1[2], A1[3]}, acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(s16x4{W1[4], W1[5], W1[6], W1[7]}, s16x4{A1[4], A1[5], A1[6], A1[7]}, acc[g], 0, 0, 0);
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
  // epilogue: bias, LeakyReLU, scale; NCHW fp32 store (lane = pixel: 32 consecutive x per row and channel)
  const long long HW = (long long)p.H * p.W;
  const int x = x0 + pcol;
  float bias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bias[e] = (p.bias && e < p.cout) ? p.bias[e] : 0.f;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int y = y0 + wave * 4 + 2 * g + prow;
    if (y >= p.H || x >= p.W) continue;
    float* o = (float*)(p.out + (long long)n * p.out_nb) + (long long)y * p.W + x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e >= p.cout) break;
      float v = acc[g][e] + bias[e];
      v = v > 0.f ? v : v * p.slope;
      o[e * HW] = v * p.alpha;
    }
  }
}

// ------------------------------------------------------------------------------------------------ 32x32 ring tile
// The chain kernel's tile: 32 rows x 32 columns x 32*COT couts on ONE 8-wave workgroup per CU (two waves per SIMD, 4 rows a wave).
// What differs from conv_tile_h, and why (measured on the 16-row two-workgroups-per-CU form, tools/chain_phase.py and the ISA):
//  * LDS ring of THREE stages for the 32-cout convs (two for the 64-cout conv, whose chunk holds twice the MFMA work): the LDS-DMA
//    of chunk c+2 is in flight while chunk c is multiplied, so the per-chunk barrier meets landed data (the two-stage form spent
//    15 % of its loop waiting there, both workgroups of a CU at once).  The wait is counted: every wave issues exactly R pieces per
//    chunk (padding pieces copy the zero line to a spare unit), so `s_waitcnt vmcnt(R * later chunks in flight)` + a raw s_barrier
//    replaces __syncthreads()' vmcnt(0).
//  * one weight image per 32 rows instead of per 16: 20 % fewer L2->LDS bytes per MFMA.
//  * operand reads are software-pipelined by tap column: the reads of column dx+1 are issued before the MFMAs of column dx
//    (hipcc left alone emits read -> wait -> MFMA per instruction and relies on four waves per SIMD to cover the LDS latency).
constexpr int T32_XU = 37;                            // X units of 1 KiB: 34 x 34 pixels x 32 B
constexpr int T32_LDS = 3 * (T32_XU + 9) * 1024 + 1024;  // three 32-cout stages (>= two 64-cout stages) + the spare unit

template <int COT, bool WT_OUT>
__device__ __forceinline__ void conv_tile32_h(const ConvParamsH p, const int tx, const int ty, const int n, char* smem,
                                              long long* tk = nullptr) {
  constexpr int NW = 8, PT = 4, TH = 32, XROW = 34, XPIX = XROW * (TH + 2);
  constexpr int NXU = T32_XU, NWU = 9 * COT, UNITS = NXU + NWU;
  constexpr int XBYTES = NXU * 1024, WBYTES = NWU * 1024, STAGE = UNITS * 1024;
  constexpr int NS = COT == 1 ? 3 : 2, D = NS - 1;  // ring depth, chunks in flight ahead of the one being multiplied
  constexpr int R = (UNITS + NW - 1) / NW;           // LDS-DMA pieces per wave and chunk (uniform: counted waits)
  constexpr int RX = (NXU + NW - 1) / NW;            // rounds that can hold an X piece
  static_assert(NS * STAGE + 1024 <= T32_LDS, "ring does not fit");
  char* const spare = smem + T32_LDS - 1024;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int x0 = tx * 32, y0 = ty * TH;
  const long long plane_b = (long long)p.in_h * p.in_w * 32;
  const char* in_n = p.in + (long long)n * p.in_nb;
  const int nchunk = p.cin_blocks;

  int xoff[RX];
#pragma unroll
  for (int r = 0; r < RX; ++r) {
    const int q = (r * NW + wave) * 64 + lane;
    const int pix = q >> 1, half = q & 1;
    const int row = pix / XROW, col = pix - row * XROW;
    const int gy = y0 - 1 + row, gx = x0 - 1 + col;
    const bool valid = (pix < XPIX) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    xoff[r] = valid ? ((gy * p.in_w + gx) * 32 + (half ^ ((col >> 3) & 1)) * 16) : -1;  // bank swizzle (file header)
  }
  const int wsw = (lane ^ ((lane >> 4) & 1)) * 16;
  // loop invariants in registers: the counted-wait asm below is a memory clobber, behind which the compiler would otherwise
  // re-load these from the kernel arguments (an s_load + lgkmcnt(0) per LDS-DMA piece)
  const void* const zero = p.zero;
  const char* const wbase = p.w + wsw;

  auto stage = [&](int buf, int cb) {  // exactly R pieces per wave, whatever cb
    char* base = smem + buf * STAGE;
    const bool live = cb < nchunk;
    const char* plane = in_n + (size_t)cb * plane_b;
    const char* wsrc = wbase + (size_t)cb * WBYTES;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int u = r * NW + wave;
      if (live && u < NXU) {
        const int xo = r < RX ? xoff[r < RX ? r : 0] : -1;
        glds16h(xo >= 0 ? (const void*)(plane + xo) : zero, base + u * 1024);
      } else if (live && u < UNITS) {
        glds16h(wsrc + (u - NXU) * 1024, base + u * 1024);
      } else {
        glds16h(zero, spare);
      }
    }
  };

  f32x16 acc[COT][PT];
#pragma unroll
  for (int a = 0; a < COT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int xrow0 = ((wave * PT) * XROW + j) * 32;
  int xl[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) xl[dx] = xrow0 + dx * 32 + ((h ^ (((j + dx) >> 3) & 1)) * 16);
  const int wlane = j * 32 + ((h ^ ((j >> 3) & 1)) * 16);

  struct Ops {
    bf16x8 bx[PT + 2];
    bf16x8 a[3][COT];
  };
  auto load = [&](Ops& o, const char* xb, const char* ws, int dx) {
    const char* xs = xb + xl[dx];
#pragma unroll
    for (int r = 0; r < PT + 2; ++r) o.bx[r] = *(const bf16x8*)(xs + r * XROW * 32);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int c = 0; c < COT; ++c) o.a[dy][c] = *(const bf16x8*)(ws + ((dy * 3 + dx) * COT + c) * 1024);
  };
  auto mfma = [&](const Ops& o) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int r = 0; r < PT; ++r)
          acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.a[dy][c], o.bx[r + dy], acc[c][r], 0, 0, 0);
  };
  auto compute = [&](int buf) {
    const char* xb = smem + buf * STAGE;
    const char* ws = xb + XBYTES + wlane;
    Ops o0, o1;
    load(o0, xb, ws, 0);
    __builtin_amdgcn_sched_barrier(0);
    load(o1, xb, ws, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma(o0);
    __builtin_amdgcn_sched_barrier(0);
    load(o0, xb, ws, 2);
    __builtin_amdgcn_sched_barrier(0);
    mfma(o1);
    __builtin_amdgcn_sched_barrier(0);
    mfma(o0);
    __builtin_amdgcn_sched_barrier(0);
  };

  if (tk) tk[0] = __builtin_readcyclecounter();
#pragma unroll
  for (int c = 0; c < D; ++c) stage(c, c);
  int buf = 0, nxt = D % NS;
  if (tk) tk[1] = __builtin_readcyclecounter();
  for (int c = 0; c < nchunk; ++c) {
    // this wave's pieces of chunk c have landed once at most (D-1)*R younger ones are outstanding; the barrier extends that to every
    // wave's pieces and also says that everybody is done multiplying chunk c-1, whose stage the refill below overwrites
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * R) : "memory");
    __builtin_amdgcn_s_barrier();
    if (tk && c == 0) tk[2] = __builtin_readcyclecounter();
    stage(nxt, c + D);
    compute(buf);
    buf = buf + 1 == NS ? 0 : buf + 1;
    nxt = nxt + 1 == NS ? 0 : nxt + 1;
  }

  if (tk) tk[3] = __builtin_readcyclecounter();
  epilogue_cb16<COT, PT, WT_OUT>(p, acc, 0, n, x0 + j, y0 + wave * PT, h);
}

// ------------------------------------------------------------------------------------------------ persistent conv chain
// The five convs of a residual dense block (rrdbnet_arch.py:32-39) — any chain in which conv k reads what convs < k wrote over the
// same pixel grid — as ONE launch.  Work items (conv k, tile t) are claimed from a global counter in k-major order, so every
// dependency of a claimed item (conv k-1 on t and its 8 neighbour tiles: the 1-pixel halo) was claimed earlier, i.e. is running on
// a resident workgroup or finished: progress never depends on how many workgroups are resident (no co-residency assumption, no
// grid barrier), and conv k+1 starts on the first tiles while conv k still runs on the last ones — ramp-up, tail and the kernel
// boundary are paid once per chain instead of once per conv.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): the producer stores its tile write-through (sc1), every wave drains
// its stores (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, one lane stores done[tile] = epoch + convs finished with an
// agent-scope relaxed store; the consumer's nine lanes poll the neighbours' words with agent-scope relaxed loads, then ONE
// agent-scope acquire (buffer_inv sc1) + s_waitcnt vmcnt(0) + barrier, then plain loads / LDS-DMA.  Every spin is bounded: a
// consumer that waits too long raises the chain's abort word and every workgroup leaves (the host entry point reports it).
#define SR_CHAIN_MAX 5
struct ChainParams {
  ConvParamsH lv[SR_CHAIN_MAX];
  int cot[SR_CHAIN_MAX];  // cout tile (1: 32 couts, 2: 64) per conv
  int nconv, ntiles, tiles_x, tiles_y;
  int* head;   // work counter of this launch (zero before the launch)
  int* done;   // [ntiles] epoch + number of convs finished on the tile (monotone over the launches that share it)
  int* abort;  // raised on a timed-out wait
  int epoch;
  long long* dbg;  // development: per-item phase clocks (tools/chain_phase.py)
};

__device__ __forceinline__ int flag_load(const int* f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool TALL>
__global__ __launch_bounds__(512, TALL ? 2 : 4) void conv_chain_bf16_kernel(const ChainParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_item;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nitems = P.nconv * P.ntiles;
  for (;;) {
    long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    if (P.dbg) c0 = __builtin_readcyclecounter();
    if (wave == 0) {
      int item = 0;
      if (lane == 0) item = atomicAdd(P.head, 1);
      item = __builtin_amdgcn_readfirstlane(item);
      if (P.dbg) c1 = __builtin_readcyclecounter();
      if (item < nitems && item >= P.ntiles) {  // conv k > 0: wait for conv k-1 on the 3x3 tile neighbourhood
        const int k = item / P.ntiles;
        int t = item - k * P.ntiles;
        const int tx = t % P.tiles_x;
        t /= P.tiles_x;
        const int ty = t % P.tiles_y, n = t / P.tiles_y;
        bool gave_up = false;
        if (lane < 9) {
          const int ny = ty + lane / 3 - 1, nx = tx + lane % 3 - 1;
          if (ny >= 0 && ny < P.tiles_y && nx >= 0 && nx < P.tiles_x) {
            const int* f = P.done + (n * P.tiles_y + ny) * P.tiles_x + nx;
            const int want = P.epoch + k;
            int spins = 0;
            while (flag_load(f) < want) {
              __builtin_amdgcn_s_sleep(2);
              if ((++spins & 255) == 0 && (spins > (1 << 22) || flag_load(P.abort))) {
                __hip_atomic_store(P.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = true;
                break;
              }
            }
          }
        }
        if (__builtin_amdgcn_ballot_w64(gave_up)) item = nitems;  // any lane that gave up takes the whole workgroup out
        if (P.dbg) c2 = __builtin_readcyclecounter();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (P.dbg) c3 = __builtin_readcyclecounter();
      if (lane == 0) s_item = item;
    }
    __syncthreads();
    const int item = s_item;
    if (item >= nitems) break;
    const int k = item / P.ntiles;
    int t = item - k * P.ntiles;
    const int tile = t;
    const int tx = t % P.tiles_x;
    t /= P.tiles_x;
    const int ty = t % P.tiles_y, n = t / P.tiles_y;
    long long tk[4] = {0, 0, 0, 0};
    if constexpr (TALL) {
      if (P.cot[k] == 2)
        conv_tile32_h<2, true>(P.lv[k], tx, ty, n, smem, P.dbg ? tk : nullptr);
      else
        conv_tile32_h<1, true>(P.lv[k], tx, ty, n, smem, P.dbg ? tk : nullptr);
    } else {
      if (P.cot[k] == 2)
        conv_tile_h<2, 2, 8, false, false, true>(P.lv[k], 0, tx, ty, n, smem);
      else
        conv_tile_h<1, 2, 8, false, false, true>(P.lv[k], 0, tx, ty, n, smem);
    }
    if (P.dbg) c4 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have landed
    __syncthreads();                                   // ... and every other wave's; s_item may be rewritten
    if (tid == 0) __hip_atomic_store(P.done + tile, P.epoch + k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (P.dbg && tid == 0) {
      long long* o = P.dbg + (size_t)item * 16;
      o[0] = c0; o[1] = c1 - c0; o[2] = c2 - c1; o[3] = c3 - c2; o[4] = c4 - c3; o[5] = __builtin_readcyclecounter() - c4;
      o[6] = blockIdx.x; o[7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
      o[8] = tk[0] - c3; o[9] = tk[1] - tk[0]; o[10] = tk[2] - tk[1]; o[11] = tk[3] - tk[2]; o[12] = c4 - tk[3];
    }
  }
}

// ------------------------------------------------------------------------------------------------ fused dense block
// The dense block re-associated so that every input is staged ONCE.  conv k reads the concat [x | x1 .. x(k-1)]: launched conv by
// conv (or item by item in the chain kernel above) x crosses L2 -> LDS five times, x1 four times ... — 640 channel-reads for 192
// distinct channels, and at bf16 rates that fill traffic, not the MFMA, sets the pace (conv1-4: 288 FLOP per staged byte).  Here a
// workgroup owns one 16x32-pixel tile for the whole block and keeps the partial sums of ALL unfinished convs in registers:
//   input group I_s (x for s = 0, else x_s):  acc_k += W_k[:, I_s] * I_s   for every conv k > s
// Once conv s+1 has seen I_s it is complete: bias / LeakyReLU / write-through store / publish, and x_(s+1) becomes an input.
// Per accumulator the MFMA sequence (chunk order, then tap column, then tap row) is the one of conv_tile_h, so results are
// bit-identical to the conv-by-conv path.  Registers: 6 accumulator groups (conv1-4: 32 couts, conv5: 2 x 32) x 2 tile rows x 16
// fp32 per lane, at most 160 live; 8 waves per workgroup, one workgroup per CU.
// LDS (154 KB): six 20 KB tile buffers (x: 0-3, x1: 4-5, x2: 0-1, x3: 2-3, x4: 4-5 — the next input's tile lands while the current
// ones are multiplied) + a 32-piece ring of 1 KB weight pieces that streams the 468 pieces of the block in consumption order + the
// biases + a landing pad for the neighbour flags.
// Every step (a chunk x tap-column range x accumulator groups) begins with a COUNTED s_waitcnt vmcnt + raw s_barrier; what may stay
// in flight at each wait is computed at compile time from the static issue sequence (make_sched; vector-memory operations retire in
// issue order), and every wave issues the same number of LDS-DMA instructions per step (pieces are dealt out in groups of eight).
// Hand-off between tiles, with nothing drained: a conv's tile leaves with write-through stores and is published a few steps later,
// at a step whose counted wait covers those stores; wave 0 fetches its nine neighbours' progress words by LDS-DMA (agent scope) a
// few steps before the dependent tile is issued and inspects them behind a counted wait; the dependent tile itself is loaded at agent
// scope (sc1: no cache invalidate).  The phases are ordered so that every hand-off (store -> publish -> flag -> halo fetch, ~4
// memory round trips) runs under MFMAs that do not depend on it: after conv k's own phase come partial sums of later convs over
// inputs that are already in the LDS (conv5's are deferred as far as its accumulation order allows).
// A tile's workgroup waits for its 8 neighbours at every hand-off.  Tiles are numbered image-major, row-major over the whole batch and
// workgroup b of a grid of G <= CUs takes tiles b, b + G, ... in increasing order, so the tiles in flight are a window of that order
// (see the kernel): the host only asks for G >= 6 tiles_x + 2 and falls back to the chain kernel otherwise.
#define SR_FZ_NS fz
#define SR_FZ_PT 2
#define SR_FZ_KERNEL rdb_fused_bf16_kernel
#include "fused_block.inc"
#undef SR_FZ_NS
#undef SR_FZ_PT
#undef SR_FZ_KERNEL
// ... and on 8-row tiles (one row per wave), for launches whose 16-row tiles would leave most of the chip idle: a batch of 32 x 32
// training patches is 64 tiles of 16 x 32 on 256 CUs, and a tile's five convs are one dependency chain
// (their tile buffers are 11 KB instead of 20: the weight ring grows to 64 pieces, which lets the partial-sum phases advance a whole
// chunk per step — 34 steps per tile instead of 62 — with hand-off lags of one or two of those longer steps)
#define SR_FZ_NS fz8
#define SR_FZ_PT 1
#define SR_FZ_KERNEL rdb_fused8_bf16_kernel
#undef SR_FZ_RING
#undef SR_FZ_PERDX
#undef SR_FZ_CLAIMLEAD
#undef SR_FZ_PUBLAG
#undef SR_FZ_TILELAG
#undef SR_FZ_FLAGLEAD
#define SR_FZ_RING 64
#define SR_FZ_PERDX 0
#define SR_FZ_CLAIMLEAD 2
#define SR_FZ_PUBLAG {1, 1, 1, 1}
#define SR_FZ_TILELAG {2, 2, 2, 2}
#define SR_FZ_FLAGLEAD {1, 1, 1, 1}
#include "fused_block.inc"
#undef SR_FZ_NS
#undef SR_FZ_PT
#undef SR_FZ_KERNEL
// ... and the 16-row tile on FOUR waves of four rows each (one wave per SIMD, 512 registers): a weight fragment read from the LDS
// feeds four MFMAs instead of two and six pixel rows serve four output rows instead of four serving two — 0.44 of the LDS read bytes
// per tile.  The chip holds this kernel's clock down under load (1.5 GHz; in cycles the tile is 1.4 x its MFMAs), and LDS
// read bytes are what the clock pays for.  Schedule, ring and lags of the 16-row instance.
#define SR_FZ_NS fz4
#define SR_FZ_PT 4
#undef SR_FZ_NW
#define SR_FZ_NW 4
#define SR_FZ_KERNEL rdb_fused4_bf16_kernel
#undef SR_FZ_RING
#undef SR_FZ_PERDX
#undef SR_FZ_CLAIMLEAD
#undef SR_FZ_PUBLAG
#undef SR_FZ_TILELAG
#undef SR_FZ_FLAGLEAD
#include "fused_block.inc"
#undef SR_FZ_NS
#undef SR_FZ_PT
#undef SR_FZ_NW
#undef SR_FZ_KERNEL

// ------------------------------------------------------------------------------------------------ streaming conv (Cin <= 64)
// The large-image layers with few input channels — conv_hr / the upsampling convs of the generator's head, the U-Net
// discriminator's 512x512 and 256x256 levels and their data gradients — are 1-4 chunks deep: on the per-tile kernel above a tile is
// a prologue (first chunk's latency), 1-4 short chunk steps and an epilogue, and every tile re-fetches the whole weight image
// (half of its L2 -> LDS traffic); measured 0.39 of HBM and 0.37 of the MFMA peak on 64 -> 64 channels at 512x512.  Here one
// workgroup per CU walks a strip of tiles with ONE continuous pipeline: the weights (<= 72 KB) stay in the LDS for the whole launch,
// tile chunks stream through a ring of NS = 4-6 buffers that runs across tile boundaries (D = NS-1 chunks in flight: the next
// tiles' data is on its way while this tile is multiplied and stored), every step is a counted s_waitcnt vmcnt + raw s_barrier, and
// the one optional extra operand of the epilogue (a residual source or the LeakyReLU-backward mask) is fetched one tile ahead,
// in issue order behind that tile's chunks, so it never waits behind the ring's younger LDS-DMA.
namespace st {
constexpr int NW = 8, PT = 2, TH = NW * PT, XROW = 34, XPIX = (TH + 2) * XROW, XU = 20, XBUF = XU * 1024;
constexpr int R = 3;  // LDS-DMA pieces per wave and chunk (24 >= 20: the extra ones copy nothing into a spare piece)

template <int COT, int NC, bool AUX>
struct Cfg {
  static constexpr int WBYTES = NC * 9 * COT * 1024;
  static constexpr int NS_FIT = (160 * 1024 - WBYTES - 1024 - 512) / XBUF;
  static constexpr int NS_MAX = NC == 1 ? 4 : 6;  // (one-chunk tiles: every ring slot is a tile with its own loads and stores in flight)
  static constexpr int NS = NS_FIT > NS_MAX ? NS_MAX : NS_FIT;
  static constexpr int D = NS - 1;
  static constexpr int LDS_RING = WBYTES, LDS_SPARE = WBYTES + NS * XBUF, LDS_BIAS = LDS_SPARE + 1024, LDS_BYTES = LDS_BIAS + 256;
  static constexpr int A = AUX ? 4 * COT : 0;  // loads of the extra epilogue operand per tile
  static constexpr int S = 4 * COT;            // stores per tile
  static_assert(NS >= 3 && D <= 5, "ring length");
  // what may stay in flight when step (tile t, chunk c) starts: the operations issued after that chunk's LDS-DMA.  The issue
  // sequence of a wave: prologue = chunks 0..D-1 (the extra operand of tile 0 right behind chunk 0); step g = chunk g+D, then, on a
  // tile's first chunk, the previous tile's stores and the next tile's extra operand.
  static constexpr int K(int t, int c) {
    int seq = 0, pos[64] = {};
    for (int g = 0; g < D; ++g) {
      seq += R;
      pos[g] = seq;
      if (g == 0) seq += A;
    }
    int k = 0;
    for (int g = 0; g <= t * NC + c; ++g) {
      k = seq - pos[g];
      seq += R;
      pos[g + D] = seq;
      if (g % NC == 0) seq += (g > 0 ? S : 0) + A;
    }
    return k > 63 ? 63 : k;  // the counter's range (waiting for more than necessary is safe)
  }
};

struct Cursor {  // a position in the workgroup's strip of tiles
  int ti, n, ty, tx;
  __device__ __forceinline__ void next(const ConvParamsH& p) {
    ++ti;
    if (++tx == p.tiles_x) {
      tx = 0;
      if (++ty == p.tiles_y) {
        ty = 0;
        ++n;
      }
    }
  }
};

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
}  // namespace st

template <int COT, int NC, bool AUX>
__global__ __launch_bounds__(512, 2) void conv_stream_bf16_kernel(const ConvParamsH p, const int per, const int aux_kind) {
  using namespace st;
  typedef Cfg<COT, NC, AUX> C;
  typedef __attribute__((address_space(3))) void* lds_void_p;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int NT = p.tiles_x * p.tiles_y * p.cogs;  // cogs carries the image count here
  const int t0 = blockIdx.x * per, t1 = t0 + per < NT ? t0 + per : NT;
  if (t0 >= t1) return;
  const unsigned plane_b = (unsigned)(p.in_h * p.in_w * 32), oplane_b = (unsigned)(p.H * p.W * 32);
  // the weight image and the bias become resident
  {
    const __amdgpu_buffer_rsrc_t w_rs = rsrc(p.w, (unsigned)C::WBYTES);
    const unsigned wvo = (lane ^ ((lane >> 4) & 1)) * 16;
    for (int u = wave; u < NC * 9 * COT; u += NW)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_void_p)(smem + u * 1024), 16, wvo, (unsigned)u * 1024u, 0, 0);
    if (tid < 64) ((float*)(smem + C::LDS_BIAS))[tid] = (p.bias && tid < 32 * COT) ? p.bias[tid] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const __amdgpu_buffer_rsrc_t null_rs = rsrc(p.in, 0u);
  // issue side: the cursor runs D chunks ahead of the compute side
  Cursor ic = {t0, 0, 0, 0};
  {
    int t = t0;
    ic.tx = t % p.tiles_x;
    t /= p.tiles_x;
    ic.ty = t % p.tiles_y;
    ic.n = t / p.tiles_y;
  }
  Cursor cc = ic;
  int ichunk = 0, islot = 0;
  unsigned xvo[R];
  auto lane_offsets = [&](const Cursor& c) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int u = r * NW + wave;
      const int q = u * 64 + lane;
      const int pix = q >> 1, half = q & 1;
      const int row = pix / XROW, col = pix - row * XROW;
      const int gy = c.ty * TH - 1 + row, gx = c.tx * 32 - 1 + col;
      const bool valid = u < XU && pix < XPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      xvo[r] = valid ? (unsigned)((((gy >> p.src_shift) * p.in_w + (gx >> p.src_shift)) * 32 + (half ^ ((col >> 3) & 1)) * 16)) : 0xfffffff0u;
    }
  };
  auto issue_chunk = [&]() {  // the chunk under the issue cursor (nothing real behind the strip's end: same instruction count)
    const bool live = ic.ti < t1;
    if (ichunk == 0) lane_offsets(ic);
    const __amdgpu_buffer_rsrc_t x_rs = live ? rsrc(p.in + (long long)ic.n * p.in_nb, (unsigned)NC * plane_b) : null_rs;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int u = r * NW + wave;
      char* dst = u < XU ? smem + C::LDS_RING + islot * XBUF + u * 1024 : smem + C::LDS_SPARE;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_void_p)dst, 16, xvo[r], (unsigned)ichunk * plane_b, 0, 0);
    }
    islot = islot + 1 == C::NS ? 0 : islot + 1;
    if (++ichunk == NC) {
      ichunk = 0;
      ic.next(p);
    }
  };
  // the extra epilogue operand of a tile (residual source or mask): 4 * COT loads, double-buffered over tiles
  u32x4s aux[COT][PT][2], aux_next[COT][PT][2];  // this tile's / the next tile's (registers: no run-time indexing)
  auto out_lane_offset = [&](const Cursor& c) {
    const int x = c.tx * 32 + j;
    return x < p.W ? (unsigned)(((c.ty * TH + wave * PT) * p.W + x) * 32 + h * 16) : 0xfffffff0u;
  };
  auto fetch_aux = [&](const Cursor& c, u32x4s (&dst)[COT][PT][2]) {
    if constexpr (AUX) {
      const char* base = aux_kind == 0 ? p.res1 : aux_kind == 1 ? p.res2 : p.mask;
      const long long nb = aux_kind == 0 ? p.res1_nb : aux_kind == 1 ? p.res2_nb : p.mask_nb;
      const bool live = c.ti < t1;
      const __amdgpu_buffer_rsrc_t a_rs = live ? rsrc(base + (long long)c.n * nb, (unsigned)(2 * COT) * oplane_b) : null_rs;
      const unsigned vo = out_lane_offset(c);
#pragma unroll
      for (int cg = 0; cg < COT; ++cg)
#pragma unroll
        for (int r = 0; r < PT; ++r)
#pragma unroll
          for (int m = 0; m < 2; ++m)
            dst[cg][r][m] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, vo == 0xfffffff0u ? vo : vo + (unsigned)(r * p.W * 32),
                                                                      (unsigned)(cg * 2 + m) * oplane_b, 0);
    }
  };

  // prologue
#pragma unroll
  for (int g = 0; g < C::D; ++g) {
    issue_chunk();
    if (g == 0) {
      __builtin_amdgcn_sched_barrier(0);
      fetch_aux(cc, aux_next);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  const int xrow0 = ((wave * PT) * XROW + j) * 32;
  int xl[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) xl[dx] = xrow0 + dx * 32 + ((h ^ (((j + dx) >> 3) & 1)) * 16);
  const int wlane = j * 32 + ((h ^ ((j >> 3) & 1)) * 16);
  int cslot = 0;

  // One third of a tile's epilogue (store groups [3 part, 3 part + 3) of 4 COT): the operations and their order are epilogue_cb16's
  // (bias, LeakyReLU, scale, residual scale-add, mask); the extra operand comes out of the registers it was fetched into.
  auto epilogue_part = [&](f32x16 (&ac)[COT][PT], const Cursor& c, const int part) {
    typedef const __attribute__((address_space(3))) f32x4* lds_f4_p;
    const unsigned vo = out_lane_offset(c);
    const bool has_bias = p.bias != nullptr;
    const float beta = aux_kind == 0 ? p.beta1 : p.beta2;
#pragma unroll
    for (int grp = 3 * part; grp < 3 * part + 3 && grp < 4 * COT; ++grp) {
      const int cg = grp >> 2, r = (grp >> 1) & 1, m = grp & 1;
      u32x4s ax = {0u, 0u, 0u, 0u};
      if constexpr (AUX) {
        ax = aux[cg][r][m];
        auto r0 = __builtin_amdgcn_permlane32_swap(ax[0], ax[2], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(ax[1], ax[3], false, false);
        ax = u32x4s{r0[0], r1[0], r0[1], r1[1]};
      }
      f32x4 bias[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) bias[q] = *(lds_f4_p)(smem + C::LDS_BIAS + (cg * 32 + (2 * m + q) * 8 + h * 4) * 4);
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float tt = ac[cg][r][m * 8 + i];
        if (has_bias) tt += bias[i >> 2][i & 3];
        tt = tt > 0.f ? tt : tt * p.slope;
        tt *= p.alpha;
        if constexpr (AUX) {
          const unsigned wa = ax[i >> 1];
          const float av = __builtin_bit_cast(float, (i & 1) ? (wa & 0xffff0000u) : (wa << 16));
          if (aux_kind < 2)
            tt += beta * av;
          else if (!(av > 0.f))
            tt *= p.mask_slope;
        }
        v[i] = tt;
      }
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      auto pk = [](float lo, float hi) {
        bf16x2 q = {(__bf16)lo, (__bf16)hi};
        return __builtin_bit_cast(unsigned, q);
      };
      u32x4s o = {pk(v[0], v[1]), pk(v[2], v[3]), pk(v[4], v[5]), pk(v[6], v[7])};
      auto s0 = __builtin_amdgcn_permlane32_swap(o[0], o[2], false, false);
      auto s1 = __builtin_amdgcn_permlane32_swap(o[1], o[3], false, false);
      o = u32x4s{s0[0], s1[0], s0[1], s1[1]};
      // exec-masked where the column lies beyond the image: the instruction is issued either way (the waits count it)
      if (vo != 0xfffffff0u) {
        // (only the one-chunk instance without an extra operand — the U-Net's conv0 — offers it: the deeper ones have no registers to spare)
        if (NC == 1 && !AUX && p.out_u2) {  // pixel-unshuffled destination: plane (2 ry + rx) CB + cb at half resolution
          const int x = c.tx * 32 + j, y = c.ty * TH + wave * PT + r;
          const size_t uo = (size_t)((((y & 1) << 1) | (x & 1)) * p.cout_blocks + cg * 2 + m) * (oplane_b >> 2) +
                            (size_t)((y >> 1) * (p.W >> 1) + (x >> 1)) * 32 + h * 16;
          store16(p.out + (long long)c.n * p.out_nb + uo, o, false);
        } else {
          store16(p.out + (long long)c.n * p.out_nb + (size_t)(cg * 2 + m) * oplane_b + vo + (unsigned)(r * p.W * 32), o, false);
        }
      }
    }
  };

  // A tile: NC chunk steps.  The epilogue of the PREVIOUS tile rides on this tile's first chunk — a third of it behind the MFMAs of
  // each tap column, vector work in the matrix pipe's shadow — instead of holding every wave of the workgroup between two tiles.
  Cursor pc = cc;
  bool have_prev = false;
  auto tile_body = [&](f32x16 (&acc)[COT][PT], f32x16 (&accp)[COT][PT], const int trel) {
#pragma unroll
    for (int a = 0; a < COT; ++a)
#pragma unroll
      for (int b = 0; b < PT; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      // steady from the tile whose chunks were all issued by regular steps (t * NC + c >= D): seven tile cases cover D <= 5
#define SR_ST_WAIT(T)                                                                                          \
  switch (c) {                                                                                                 \
    case 0: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::K(T, 0)) : "memory"); break;                           \
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::K(T, NC > 1 ? 1 : 0)) : "memory"); break;             \
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::K(T, NC > 2 ? 2 : 0)) : "memory"); break;             \
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::K(T, NC > 3 ? 3 : 0)) : "memory"); break;            \
  }
      switch (trel < 6 ? trel : 6) {
        case 0: SR_ST_WAIT(0) break;
        case 1: SR_ST_WAIT(1) break;
        case 2: SR_ST_WAIT(2) break;
        case 3: SR_ST_WAIT(3) break;
        case 4: SR_ST_WAIT(4) break;
        case 5: SR_ST_WAIT(5) break;
        default: SR_ST_WAIT(6) break;
      }
#undef SR_ST_WAIT
      __builtin_amdgcn_s_barrier();
      issue_chunk();
      const char* xb = smem + C::LDS_RING + cslot * XBUF;
      const char* ws = smem + c * (9 * COT * 1024) + wlane;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        bf16x8 bx[PT + 2], a[3][COT];
        const char* xs = xb + xl[dx];
#pragma unroll
        for (int r = 0; r < PT + 2; ++r) bx[r] = *(const bf16x8*)(xs + r * XROW * 32);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int cg = 0; cg < COT; ++cg) a[dy][cg] = *(const bf16x8*)(ws + ((dy * 3 + dx) * COT + cg) * 1024);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int cg = 0; cg < COT; ++cg)
#pragma unroll
            for (int r = 0; r < PT; ++r)
              acc[cg][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[dy][cg], bx[r + dy], acc[cg][r], 0, 0, 0);
        if (c == 0 && have_prev) epilogue_part(accp, pc, dx);
      }
      if (c == 0) {  // this tile's extra operand moves up, the next tile's is fetched (behind the stores above in the issue order)
        if constexpr (AUX) {
#pragma unroll
          for (int cg = 0; cg < COT; ++cg)
#pragma unroll
            for (int r = 0; r < PT; ++r)
#pragma unroll
              for (int m = 0; m < 2; ++m) aux[cg][r][m] = aux_next[cg][r][m];
        }
        Cursor nx = cc;
        nx.next(p);
        __builtin_amdgcn_sched_barrier(0);
        fetch_aux(nx, aux_next);
        __builtin_amdgcn_sched_barrier(0);
      }
      cslot = cslot + 1 == C::NS ? 0 : cslot + 1;
    }
    pc = cc;
    have_prev = true;
    cc.next(p);
  };
  f32x16 acc0[COT][PT], acc1[COT][PT];
  int t = t0;
  for (; t + 1 < t1; t += 2) {
    tile_body(acc0, acc1, t - t0);
    tile_body(acc1, acc0, t + 1 - t0);
  }
  if (t < t1) {
    tile_body(acc0, acc1, t - t0);
#pragma unroll
    for (int part = 0; part < 3; ++part) epilogue_part(acc0, pc, part);
  } else {
#pragma unroll
    for (int part = 0; part < 3; ++part) epilogue_part(acc1, pc, part);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's read-ahead behind the strip lands before the LDS is released
}

template <int COT, int PT, int NW>
constexpr int conv_bf16_lds() {
  return 2 * ((((NW * PT + 2) * 34 * 32 + 1023) / 1024) * 1024 + 9 * COT * 1024);
}

template <int COT, int PT, int NW, bool NCHW_OUT, bool S2 = false>
int launch_h(ConvParamsH p, int n, int groups, hipStream_t stream, const sr_conv3x3_desc* d) {
  constexpr int lds = conv_bf16_lds<COT, PT, NW>();
  static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
  auto kern = conv_bf16_kernel<COT, PT, NW, NCHW_OUT, S2>;
  if (int rc = sr::ensure_dynamic_lds((const void*)kern, lds)) return rc;  // once per (kernel, device)
  p.tiles_x = sr::cdiv(p.W, 32);
  p.tiles_y = sr::cdiv(p.H, NW * PT);
  const bool prof = sr::prof_on();
  if (prof) {
    sr_launch_record r = {};
    r.kernel_id = PT == 1 ? 42 + (COT - 1) : 16 + (COT - 1) * 2 + (PT == 4 ? 1 : 0) + (NCHW_OUT ? 4 : 0) + (NW == 8 ? 8 : 0);
    r.cin = d->cin_real > 0 ? d->cin_real : d->cin_pad;
    r.cout = d->cout;
    r.n = n;
    r.h = p.H;
    r.w = p.W;
    const double px = (double)n * p.H * p.W;
    r.flops = 2.0 * (S2 ? 4 : 9) * r.cin * r.cout * px;  // S2: the 16 taps of the 4x4 kernel = 4 per unshuffled channel
    r.bytes = 2.0 * ((double)n * p.in_h * p.in_w * r.cin + px * r.cout * (NCHW_OUT ? 2 : 1) + (d->res1 ? px * r.cout : 0) +
                     (d->res2 ? px * r.cout : 0));
    sr::prof_begin(stream, r);
  }
  p.cogs = groups;
  hipLaunchKernelGGL(kern, dim3(p.tiles_x * p.tiles_y * n * groups), dim3(NW * 64), lds, stream, p);
  if (prof) sr::prof_end(stream);
  SR_CHECK_LAUNCH("conv_bf16 launch");
  return SR_OK;
}

long long* g_phase_clocks = nullptr;

}  // namespace

// Development aid (tools/bf16_phase.py; not part of the ABI in include/sr_hip.h): when set, every wave of the next
// launches writes its prologue / compute / barrier-wait / epilogue cycle counts to buf[(workgroup*NW + wave)*8 ..].
extern "C" void sr_dev_conv_bf16_phase_clocks(void* buf) { g_phase_clocks = (long long*)buf; }

static int fill_params_h(const sr_conv3x3_desc* d, ConvParamsH& p, const char* who) {
  SR_CHECK_ARG(d && d->in && d->wpacked && d->out, "%s: null argument", who);
  SR_CHECK_ARG(d->cin_pad > 0 && d->cin_pad % 16 == 0, "%s: cin_pad=%d must be a multiple of 16", who, d->cin_pad);
  SR_CHECK_ARG(d->cout > 0 && d->n > 0 && d->in_h > 0 && d->in_w > 0, "%s: bad shape", who);
  SR_CHECK_ARG(!d->accumulate && d->res_cbn == 0 && d->mask_cb0 == 0 && !(d->mask_src && d->out_nchw),
               "%s: accumulate / res_cbn / mask_cb0 are fp32-path options", who);
  SR_CHECK_ARG(((uintptr_t)d->in | (uintptr_t)d->wpacked | (uintptr_t)d->out | (uintptr_t)d->res1 | (uintptr_t)d->res2 |
                (uintptr_t)d->bpacked | (uintptr_t)d->mask_src) % 16 == 0,
               "%s: pointers must be 16-byte aligned", who);
  SR_CHECK_ARG(d->s2_channels == 0 || (d->s2_channels > 0 && d->s2_channels % 64 == 0 && !d->upsample &&
                                       (d->s2_side ? d->cout : d->cin_pad) == 4 * d->s2_channels),
               "%s: s2_channels=%d must be a multiple of 64 and a quarter of the %s channels", who, d->s2_channels,
               d->s2_side ? "output" : "input");
  p = ConvParamsH{};
  p.zero = sr::zero_line();
  p.in = (const char*)d->in;
  p.w = (const char*)d->wpacked;
  p.bias = d->bpacked;
  p.out = (char*)d->out;
  p.res1 = (const char*)d->res1;
  p.res2 = (const char*)d->res2;
  p.in_nb = d->in_img_stride * 2;
  p.out_nb = d->out_img_stride * (d->out_nchw ? 4 : 2);
  p.res1_nb = d->res1_img_stride * 2;
  p.res2_nb = d->res2_img_stride * 2;
  p.mask = (const char*)d->mask_src;
  p.mask_nb = d->mask_img_stride * 2;
  p.mask_cbn = d->mask_cbn;
  p.mask_slope = d->mask_slope;
  p.cin_blocks = d->cin_pad / 16;
  p.cout_blocks = (d->cout + 15) / 16;
  p.cout = d->cout;
  p.in_h = d->in_h;
  p.in_w = d->in_w;
  p.H = d->upsample ? 2 * d->in_h : d->in_h;
  p.W = d->upsample ? 2 * d->in_w : d->in_w;
  p.src_shift = d->upsample ? 1 : 0;
  p.slope = d->act_slope;
  p.alpha = d->alpha;
  p.beta1 = d->beta1;
  p.beta2 = d->beta2;
  p.dbg = nullptr;
  p.out_u2 = d->out_unshuffle2 ? 1 : 0;
  p.res1_mode = (d->res1_u2 ? 1 : 0) | (d->res1_keep_sign ? 2 : 0);
  SR_CHECK_ARG(!p.res1_mode || (d->res1 && !d->out_nchw && !d->out_unshuffle2 && d->cout % 16 == 0 && p.H % 2 == 0 && p.W % 2 == 0 &&
                                (!d->res1_keep_sign || (d->beta1 == 1.f && d->alpha > 0.f && !d->res2 && !d->mask_src))),
               "%s: res1_u2 / res1_keep_sign need res1, a plain CB16 destination of even size with cout %% 16 == 0 (keep_sign: beta1 = 1, "
               "alpha > 0, no res2 / mask)", who);
  SR_CHECK_ARG(!p.out_u2 || (!d->out_nchw && d->cout % 16 == 0 && p.H % 2 == 0 && p.W % 2 == 0),
               "%s: out_unshuffle2 needs a CB16 destination, cout %% 16 == 0 and an even output size (%dx%d, cout %d)", who, p.H, p.W, d->cout);
  SR_CHECK_ARG((long long)p.H * p.W * 32 * (long long)(p.cout_blocks > p.cin_blocks ? p.cout_blocks : p.cin_blocks) < (1ll << 31),
               "%s: image too large for 32-bit plane offsets", who);
  return SR_OK;
}

// Ints of the `sync` block of sr_conv3x3_chain_bf16 for images of h x w pixels: [0] abort word, [1 .. 1+SR_CHAIN_EPOCHS) the work
// counters of the calls that share the block, then one progress word per 16x32 tile.
#define SR_CHAIN_EPOCHS 256
extern "C" size_t sr_conv3x3_chain_sync_ints(int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  return 1 + SR_CHAIN_EPOCHS + (size_t)n * sr::cdiv(h, 8) * sr::cdiv(w, 32);  // (progress words: one per 8-row tile, the smallest)
}

// Default: 16-row tiles, two workgroups per CU (mode 2).  Same box, BASELINE config 2 in bf16 (batch 16, 20 steps): conv by conv
// 1502 img/s, mode 2 1541, mode 1 (32-row ring tiles, one workgroup per CU) 1458; a dense block alone 159 -> 147 us at batch 16,
// 310 -> 272 us at batch 32, 771 -> 664 us on four 544x544 tiler cells.  With two workgroups per CU one's hand-off, prologue and
// epilogue overlap the other's MFMA loop; the single 32-row workgroup has nothing to overlap them with.
// 8-row tiles with a whole chunk per step (34 steps per tile instead of 62) where 16-row tiles would give fewer than half of the CUs a
// tile.  1 (default): forward blocks only.  A block alone: 42.7 -> 31.6 us at 32 x 32x32, 41.7 -> 31.5 us on one 64x64 image; the
// reference recipe's step 21.2 -> 20.0 ms.  2 = also the transposed block of the backward pass: 21.9 ms — that block shares the chip
// with the weight-gradient lane, which loses more CUs to 128 small tiles than the block gains.  (sr_dev_set_fused_rows8)
static int g_fused_rows8 = 1;
// the lean 16-row launches on four waves of four rows (rdb_fused4_bf16_kernel) instead of eight of two.  (sr_dev_set_fused_wave4)
static int g_fused_wave4 = 0;
static int g_chain_enabled = 3;  // 0 off, 1 = 32-row ring tiles (one workgroup per CU), 2 = 16-row tiles (two workgroups per CU),
                                 // 3 = fused dense block (rdb_fused_bf16_kernel) where eligible, else as 2
static long long* g_chain_clocks = nullptr;
// Development aid (tools/chain_phase.py; not part of the ABI): per work item, wave 0 writes claim / wait / acquire / tile / drain clocks.
extern "C" void sr_dev_chain_phase_clocks(void* buf) { g_chain_clocks = (long long*)buf; }
namespace sr {
int chain_mode() { return g_chain_enabled; }
}
extern "C" int sr_set_conv_chain(int enabled) {
  g_chain_enabled = enabled;
  return SR_OK;
}

static long long* g_fused_clocks = nullptr;
// Development aid (tools/fused_phase.py; not part of the ABI): 32 cycle stamps per workgroup of the fused dense-block kernel.
extern "C" void sr_dev_fused_phase_clocks(void* buf) { g_fused_clocks = (long long*)buf; }
// Development hook (tests/test_watchdog_gpu.py; not part of the ABI): launch the fused dense block with at most `cap` workgroups even
// when that is fewer than the window of tile rows its hand-offs need — the launch then cannot make progress, every bounded wait spins
// out, the abort word is raised and the give-up path (ctl word -> return false -> drain) runs: what a GPU that withholds CUs from this
// process would cause, made deterministic.  0 = off.
static int g_fused_grid_cap = 0;
extern "C" void sr_dev_fused_grid_cap(int cap) { g_fused_grid_cap = cap; }

// The dense block as rdb_fused_bf16_kernel when the five descriptors are one (64 + 4 x 32 channel) block over a single concat buffer and
// every image's tiles fit the chip together; *launched says whether it ran.
static int g_mids_scratch = 0;  // development / tests: 1 = every forward dense block as if its caller had said "x1..x4 are scratch", -1 = nobody's word counts
extern "C" void sr_dev_set_chain_mids_scratch(int on) { g_mids_scratch = on; }
static int try_fused_dense_block(const sr_conv3x3_desc* d, int32_t* sync, int call_index, hipStream_t stream, bool* launched, bool mids_scratch) {
  *launched = false;
  const int n = d[0].n, h = d[0].in_h, w = d[0].in_w;
  const long long hw = (long long)h * w;
  if (h % 16 != 0 || hw * 32 * 12 >= (1ll << 31)) return SR_OK;
  for (int k = 0; k < 5; ++k) {
    const sr_conv3x3_desc& c = d[k];
    if (c.in != d[0].in || c.in_img_stride != d[0].in_img_stride || c.n != n || c.in_h != h || c.in_w != w || c.upsample || c.out_nchw ||
        c.s2_channels != 0 || c.out_unshuffle2 || c.res1_u2 || c.res1_keep_sign || c.cin_pad != 64 + 32 * k || c.cout != (k < 4 ? 32 : 64) || c.accumulate)
      return SR_OK;
    if (k < 4 && ((const __bf16*)c.out != (const __bf16*)d[0].in + (64 + 32 * k) * hw || c.out_img_stride != d[0].in_img_stride ||
                  c.res1 || c.res2))
      return SR_OK;
  }
  if (d[0].in_img_stride < 192 * hw) return SR_OK;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SR_OK;
  static int cu_count[16] = {0};
  if (dev < 0 || dev >= 16) return SR_OK;
  if (cu_count[dev] == 0) {
    if (int rc = sr::ensure_dynamic_lds((const void*)rdb_fused_bf16_kernel<0>, fz::LDS_BYTES)) return rc;
    if (int rc = sr::ensure_dynamic_lds((const void*)rdb_fused_bf16_kernel<1>, fz::LDS_BYTES)) return rc;
    if (int rc = sr::ensure_dynamic_lds((const void*)rdb_fused_bf16_kernel<2>, fz::LDS_BYTES)) return rc;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)rdb_fused_bf16_kernel<1>, 512, fz::LDS_BYTES) != hipSuccess ||
        per_cu < 1) {
      cu_count[dev] = -1;
    } else {
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = -1;
      cu_count[dev] = cus;
    }
  }
  if (cu_count[dev] < 1) return SR_OK;
  const int conc = sr::launch_concurrency();
  const int avail = cu_count[dev] / (conc > 1 ? conc : 1);
  // 8-row tiles where 16-row tiles would give fewer than half of the CUs a tile (a tile's five convs are one dependency chain)
  // (g_fused_rows8: 1 = forward blocks only — the transposed block of a backward pass shares the chip with the weight-gradient lane,
  // which loses more CUs to 128 small tiles than the block gains; 2 = both)
  const bool transposed_sig = !d[4].bpacked && d[0].mask_src != nullptr;
  const bool rows8 = g_fused_rows8 >= 3 || (g_fused_rows8 && (g_fused_rows8 > 1 || !transposed_sig) && (long long)sr::cdiv(w, 32) * (h / 16) * n * 2 <= avail);  // (3: every launch, development)
  if (rows8) {
    static bool lds8[16] = {false};
    if (!lds8[dev]) {
      if (int rc = sr::ensure_dynamic_lds((const void*)rdb_fused8_bf16_kernel<0>, fz8::LDS_BYTES)) return rc;
      if (int rc = sr::ensure_dynamic_lds((const void*)rdb_fused8_bf16_kernel<1>, fz8::LDS_BYTES)) return rc;
      if (int rc = sr::ensure_dynamic_lds((const void*)rdb_fused8_bf16_kernel<2>, fz8::LDS_BYTES)) return rc;
      lds8[dev] = true;
    }
  }
  const int tiles_x = sr::cdiv(w, 32), tiles_y = h / (rows8 ? 8 : 16);
  const int T = tiles_x * tiles_y;
  // the tiles in flight are a window of the row-major tile order (see the kernel): it must hold the rows a block's hand-offs can block
  const long long NT = (long long)T * n;
  int grid = (int)(NT < avail ? NT : avail);
  if (grid < NT && grid < 6 * tiles_x + 2) return SR_OK;
  if (g_fused_grid_cap > 0 && grid > g_fused_grid_cap) grid = g_fused_grid_cap;
  if (NT >= (1ll << 30)) return SR_OK;
  fz::FusedParams P = {};
  const char* lo = (const char*)d[0].wpacked;
  for (int k = 1; k < 5; ++k)
    if ((const char*)d[k].wpacked < lo) lo = (const char*)d[k].wpacked;
  unsigned long long span = 0;
  for (int k = 0; k < 5; ++k) {
    if (int rc = fill_params_h(&d[k], P.lv[k], "sr_conv3x3_chain_bf16")) return rc;
    P.lv[k].tiles_x = tiles_x;
    P.lv[k].tiles_y = tiles_y;
    P.lv[k].cogs = 1;
    const unsigned long long dl = (unsigned long long)((const char*)d[k].wpacked - lo);
    const unsigned long long end = dl + (unsigned long long)(4 + 2 * k) * 9 * (k < 4 ? 1 : 2) * 1024;
    if (end >= (1ull << 31)) return SR_OK;
    P.wdelta[k] = (unsigned)dl;
    if (end > span) span = end;
  }
  P.wbase = lo;
  P.wspan = (unsigned)span;
  P.n = n;
  P.head = sync + 1 + call_index;
  P.tiles_x = tiles_x;
  P.tiles_y = tiles_y;
  P.abort = sync;
  P.done = sync + 1 + SR_CHAIN_EPOCHS;
  P.epoch = call_index * 8;
  P.dbg = g_fused_clocks;
  // the forward block's epilogues in their lean form: conv1-4 = LeakyReLU(conv + bias), conv5 = alpha (conv + bias) + beta1 res1 [+ beta2 res2]
  bool lean = d[4].bpacked != nullptr;
  for (int k = 0; k < 4 && lean; ++k)
    lean = d[k].bpacked && d[k].alpha == 1.f && !d[k].mask_src && d[k].act_slope == d[0].act_slope && d[k].act_slope >= 0.f && d[k].act_slope <= 1.f;
  // ... and the transposed block's: conv1-4 = lrelu'(mask) * conv, conv5 = conv + beta1 res1 [+ beta2 res2]
  const bool tail_ok = d[4].res1 && !d[4].mask_src && d[4].res1_img_stride >= 64 * hw && (!d[4].res2 || d[4].res2_img_stride >= 64 * hw) &&
                       d[4].out_img_stride >= 64 * hw && d[4].act_slope == 1.f;
  bool back = tail_ok && !d[4].bpacked && d[4].alpha == 1.f;
  for (int k = 0; k < 4 && back; ++k)
    back = !d[k].bpacked && d[k].alpha == 1.f && d[k].act_slope == 1.f && d[k].mask_src && d[k].mask_cbn >= 2 &&
           d[k].mask_img_stride == d[0].mask_img_stride && d[k].mask_img_stride >= 32 * hw && d[k].mask_slope == d[0].mask_slope;
  lean = lean && tail_ok;
  if (g_chain_enabled == 4) lean = back = false;  // development: the generic epilogues
  if (lean || back) {
    P.slope = d[0].act_slope;
    P.alpha5 = d[4].alpha;
    P.beta1 = d[4].beta1;
    P.beta2 = d[4].beta2;
    P.out5 = P.lv[4].out;
    P.out5_nb = P.lv[4].out_nb;
    P.res1 = P.lv[4].res1;
    P.res1_nb = P.lv[4].res1_nb;
    P.res2 = P.lv[4].res2;
    P.res2_nb = P.lv[4].res2_nb;
    for (int k = 0; k < 4; ++k) P.mask[k] = P.lv[k].mask;
    P.mask_nb = P.lv[0].mask_nb;
    P.mask_slope = d[0].mask_slope;
    P.mids_scratch = (lean && g_mids_scratch >= 0 && (mids_scratch || g_mids_scratch > 0)) ? 1 : 0;
  }
  const bool prof = sr::prof_on();
  if (prof) {  // one record for the whole block: the FLOPs of its five convs, the bytes a block must move at least
    sr_launch_record r = {};
    r.kernel_id = 60 + (lean ? 1 : back ? 2 : 0);
    r.cin = 192;
    r.cout = 192;
    r.n = n;
    r.h = h;
    r.w = w;
    const double px = (double)n * hw;
    r.flops = 2.0 * 9 * px * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64);
    // x read, x1..x4 written, the output written, the residual sources (and the four masks of the transposed block) read
    r.bytes = 2.0 * px * (64 + 128 + 64 + (d[4].res1 ? 64 : 0) + (d[4].res2 ? 64 : 0) + (d[0].mask_src ? 128 : 0));
    sr::prof_begin(stream, r);
  }
  if (rows8) {
    static_assert(sizeof(fz8::FusedParams) == sizeof(fz::FusedParams), "one parameter block for both tile heights");
    fz8::FusedParams P8;
    std::memcpy(&P8, &P, sizeof(P));
    if (lean)
      hipLaunchKernelGGL(rdb_fused8_bf16_kernel<1>, dim3((unsigned)grid), dim3(512), fz8::LDS_BYTES, stream, P8);
    else if (back)
      hipLaunchKernelGGL(rdb_fused8_bf16_kernel<2>, dim3((unsigned)grid), dim3(512), fz8::LDS_BYTES, stream, P8);
    else
      hipLaunchKernelGGL(rdb_fused8_bf16_kernel<0>, dim3((unsigned)grid), dim3(512), fz8::LDS_BYTES, stream, P8);
  } else if (g_fused_wave4 && (lean || back)) {
    static_assert(sizeof(fz4::FusedParams) == sizeof(fz::FusedParams), "one parameter block for every instance");
    static bool lds4[16] = {false};
    if (!lds4[dev]) {
      if (int rc = sr::ensure_dynamic_lds((const void*)rdb_fused4_bf16_kernel<1>, fz4::LDS_BYTES)) return rc;
      if (int rc = sr::ensure_dynamic_lds((const void*)rdb_fused4_bf16_kernel<2>, fz4::LDS_BYTES)) return rc;
      lds4[dev] = true;
    }
    fz4::FusedParams P4;
    std::memcpy(&P4, &P, sizeof(P));
    if (lean)
      hipLaunchKernelGGL(rdb_fused4_bf16_kernel<1>, dim3((unsigned)grid), dim3(fz4::NW * 64), fz4::LDS_BYTES, stream, P4);
    else
      hipLaunchKernelGGL(rdb_fused4_bf16_kernel<2>, dim3((unsigned)grid), dim3(fz4::NW * 64), fz4::LDS_BYTES, stream, P4);
  } else if (lean)
    hipLaunchKernelGGL(rdb_fused_bf16_kernel<1>, dim3((unsigned)grid), dim3(512), fz::LDS_BYTES, stream, P);
  else if (back)
    hipLaunchKernelGGL(rdb_fused_bf16_kernel<2>, dim3((unsigned)grid), dim3(512), fz::LDS_BYTES, stream, P);
  else
    hipLaunchKernelGGL(rdb_fused_bf16_kernel<0>, dim3((unsigned)grid), dim3(512), fz::LDS_BYTES, stream, P);
  if (prof) sr::prof_end(stream);
  SR_CHECK_LAUNCH("rdb_fused_bf16 launch");
  *launched = true;
  return SR_OK;
}

extern "C" void sr_dev_set_fused_rows8(int on) { g_fused_rows8 = on; }
extern "C" void sr_dev_set_fused_wave4(int on) { g_fused_wave4 = on; }

extern "C" int sr_conv3x3_chain_bf16(const sr_conv3x3_desc* d, int nconv, int32_t* sync, int call_index, void* stream_) {
  return sr::conv3x3_chain_bf16(d, nconv, sync, call_index, (hipStream_t)stream_, false);
}
// mids_scratch: the caller reads nothing but the LAST conv's output afterwards (the inference forward: the concat buffer behind a
// dense block is scratch) — the fused kernel then stores of x1..x4 only what neighbouring tiles read; every other path stores all
int sr::conv3x3_chain_bf16(const sr_conv3x3_desc* d, int nconv, int32_t* sync, int call_index, hipStream_t stream, bool mids_scratch) {
  SR_CHECK_ARG(d && nconv >= 1, "sr_conv3x3_chain_bf16: bad argument");
  if (g_chain_enabled >= 3 && sync && nconv == 5 && call_index >= 0 && call_index < SR_CHAIN_EPOCHS) {  // (also when profiling: one record)
    bool launched = false;
    if (int rc = try_fused_dense_block(d, sync, call_index, stream, &launched, mids_scratch)) return rc;
    if (launched) return SR_OK;
  }
  // eligibility of the one-launch form: one tile grid (same n, H, W, no upsampling), CB16 outputs, <= 64 couts, 16-row tiles,
  // enough tiles to fill the chip; anything else runs conv by conv (same results)
  const bool tall = g_chain_enabled == 1;
  const int rows = tall ? 32 : 16;
  bool one_launch = g_chain_enabled && sync && nconv >= 2 && nconv <= SR_CHAIN_MAX && call_index >= 0 && call_index < SR_CHAIN_EPOCHS &&
                    !sr::prof_on();
  for (int k = 0; k < nconv && one_launch; ++k) {
    const sr_conv3x3_desc& c = d[k];
    one_launch = c.n == d[0].n && c.in_h == d[0].in_h && c.in_w == d[0].in_w && !c.upsample && !c.out_nchw && c.s2_channels == 0 && !c.out_unshuffle2 && !c.res1_u2 && !c.res1_keep_sign &&
                 c.cout <= 64 && c.cin_pad <= 256 && c.in_h % rows == 0;
  }
  const int conc = sr::launch_concurrency();
  if (one_launch) one_launch = (long long)sr::cdiv(d[0].in_w, 32) * (d[0].in_h / rows) * d[0].n * conc >= (tall ? 192 : 384);
  if (!one_launch) {
    for (int k = 0; k < nconv; ++k)
      if (int rc = sr_conv3x3_bf16(&d[k], (void*)stream)) return rc;
    return SR_OK;
  }
  ChainParams P = {};
  for (int k = 0; k < nconv; ++k) {
    if (int rc = fill_params_h(&d[k], P.lv[k], "sr_conv3x3_chain_bf16")) return rc;
    P.cot[k] = ((d[k].cout + 31) / 32 * 32) % 64 == 0 ? 2 : 1;
    P.lv[k].tiles_x = sr::cdiv(P.lv[k].W, 32);
    P.lv[k].tiles_y = P.lv[k].H / rows;
    P.lv[k].cogs = 1;
  }
  P.nconv = nconv;
  P.tiles_x = P.lv[0].tiles_x;
  P.tiles_y = P.lv[0].tiles_y;
  P.ntiles = P.tiles_x * P.tiles_y * d[0].n;
  P.abort = sync;
  P.head = sync + 1 + call_index;
  P.done = sync + 1 + SR_CHAIN_EPOCHS;
  P.epoch = call_index * 8;
  P.dbg = g_chain_clocks;
  const int lds = tall ? T32_LDS : conv_bf16_lds<2, 2, 8>();
  auto kern = tall ? conv_chain_bf16_kernel<true> : conv_chain_bf16_kernel<false>;
  if (int rc = sr::ensure_dynamic_lds((const void*)kern, lds)) return rc;
  // Grid = what the chip holds at once (one 32-row or two 16-row workgroups per CU by LDS), shared between concurrently launching
  // image groups.  Not a correctness condition (work items are claimed, see the kernel): more would only queue, fewer would idle CUs.
  static int slots[2][16] = {{0}, {0}};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (slots[tall][dev] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 512, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    const int want = tall ? 1 : 2;
    slots[tall][dev] = (per_cu > want ? want : per_cu) * cus;
  }
  long long grid = slots[tall][dev] / (conc > 1 ? conc : 1);
  if (grid > P.ntiles) grid = P.ntiles;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, stream, P);
  SR_CHECK_LAUNCH("conv_chain_bf16 launch");
  return SR_OK;
}

// d->in / out / res*: CB16 bf16 tensors (out: NCHW fp32 when out_nchw); *_img_stride in ELEMENTS of the tensor's dtype;
// cin_pad multiple of 16; wpacked from sr_conv3x3_pack_bf16 (passed through the float* field); bpacked fp32.
static int g_stream_enabled = 1;
// Development switch (not part of the ABI): 0 = large few-channel convs on the per-tile kernel instead of conv_stream_bf16_kernel.
extern "C" void sr_dev_set_conv_stream(int on) { g_stream_enabled = on; }

extern "C" int sr_conv3x3_bf16(const sr_conv3x3_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvParamsH p;
  if (int rc = fill_params_h(d, p, "sr_conv3x3_bf16")) return rc;
  p.dbg = g_phase_clocks;
  const int cp = (d->cout + 31) / 32 * 32;
  const int gc = (cp % 64 == 0) ? 64 : 32;
  const int groups = cp / gc;
  const bool w8 = p.H % 32 == 0;  // whole 32-row tiles for the 8-wave x 4-row workgroup
  if (d->out_nchw) {
    if (p.out_nb == 0) p.out_nb = (long long)d->cout * p.H * p.W * 4;
    if (d->cout <= 4 && !d->res1 && !d->res2 && !d->mask_src) {  // conv_last / logit conv: the 4-cout kernel (16-row tiles)
      constexpr int lds = 2 * (((18 * 34 * 32 + 1023) / 1024) * 1024 + 2048);
      auto kern = conv_fewcout_bf16_kernel;
      if (int rc = sr::ensure_dynamic_lds((const void*)kern, lds)) return rc;
      p.tiles_x = sr::cdiv(p.W, 32);
      p.tiles_y = sr::cdiv(p.H, 16);
      const bool prof = sr::prof_on();
      if (prof) {
        sr_launch_record r = {};
        r.kernel_id = 50;
        r.cin = d->cin_real > 0 ? d->cin_real : d->cin_pad;
        r.cout = d->cout;
        r.n = d->n;
        r.h = p.H;
        r.w = p.W;
        const double px = (double)d->n * p.H * p.W;
        r.flops = 2.0 * 9 * r.cin * r.cout * px;
        r.bytes = (double)d->n * p.in_h * p.in_w * r.cin * 2.0 + px * r.cout * 4.0;
        sr::prof_begin(stream, r);
      }
      hipLaunchKernelGGL(kern, dim3(p.tiles_x * p.tiles_y * d->n), dim3(256), lds, stream, p);
      if (prof) sr::prof_end(stream);
      SR_CHECK_LAUNCH("conv_fewcout_bf16 launch");
      return SR_OK;
    }
    if (gc == 64) return w8 ? launch_h<2, 4, 8, true>(p, d->n, groups, stream, d) : launch_h<2, 2, 4, true>(p, d->n, groups, stream, d);
    return w8 ? launch_h<1, 4, 8, true>(p, d->n, groups, stream, d) : launch_h<1, 2, 4, true>(p, d->n, groups, stream, d);
  }
  // Few input channels on a large pixel grid: the streaming kernel (resident weights, one pipeline across a strip of tiles)
  if (g_stream_enabled && p.cin_blocks <= 4 && d->cout == 64 && d->s2_channels == 0 && p.H % 16 == 0) {
    const int naux = (d->res1 ? 1 : 0) + (d->res2 ? 1 : 0) + (d->mask_src ? 1 : 0);
    int dev = 0, cus = 0;
    if (naux <= 1 && !(naux == 1 && p.cin_blocks == 2) /* that instance spills */ && (!d->mask_src || d->mask_cbn >= 4) &&
        (!p.out_u2 || (naux == 0 && p.cin_blocks == 1)) && !p.res1_mode && hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) {
      const int conc0 = sr::launch_concurrency();
      const int grid = cus / (conc0 > 1 ? conc0 : 1);
      p.tiles_x = sr::cdiv(p.W, 32);
      p.tiles_y = p.H / 16;
      const long long NT = (long long)p.tiles_x * p.tiles_y * d->n;
      if (grid >= 1 && NT >= 24ll * grid && NT < (1ll << 30)) {  // (shorter strips: measured slower than the per-tile kernel)
        p.cogs = d->n;
        const int per = (int)((NT + grid - 1) / grid);
        const int aux_kind = d->res1 ? 0 : d->res2 ? 1 : 2;
        const bool prof = sr::prof_on();
        if (prof) {
          sr_launch_record r = {};
          r.kernel_id = 64;
          r.cin = d->cin_real > 0 ? d->cin_real : d->cin_pad;
          r.cout = d->cout;
          r.n = d->n;
          r.h = p.H;
          r.w = p.W;
          const double px = (double)d->n * p.H * p.W;
          r.flops = 2.0 * 9 * r.cin * r.cout * px;
          r.bytes = 2.0 * ((double)d->n * p.in_h * p.in_w * r.cin + px * r.cout * (1 + naux));
          sr::prof_begin(stream, r);
        }
        int rc = SR_OK;
        auto go = [&](auto kern, int lds) {
          rc = sr::ensure_dynamic_lds((const void*)kern, lds);
          if (rc == SR_OK) hipLaunchKernelGGL(kern, dim3((unsigned)sr::cdiv((int)NT, per)), dim3(512), lds, stream, p, per, aux_kind);
        };
        switch (p.cin_blocks * 2 + (naux ? 1 : 0)) {
          case 2: go(conv_stream_bf16_kernel<2, 1, false>, st::Cfg<2, 1, false>::LDS_BYTES); break;
          case 3: go(conv_stream_bf16_kernel<2, 1, true>, st::Cfg<2, 1, true>::LDS_BYTES); break;
          case 4: go(conv_stream_bf16_kernel<2, 2, false>, st::Cfg<2, 2, false>::LDS_BYTES); break;
          case 5:  // excluded above: at 2 chunks + an extra operand hipcc spills 176 VGPRs (tests/test_codeobj_host.py), so it is not built
            sr::set_error("sr_conv3x3_bf16: internal dispatch error (streaming conv, 2 chunks + extra operand)");
            return SR_EINVAL;
          case 6: go(conv_stream_bf16_kernel<2, 3, false>, st::Cfg<2, 3, false>::LDS_BYTES); break;
          case 7: go(conv_stream_bf16_kernel<2, 3, true>, st::Cfg<2, 3, true>::LDS_BYTES); break;
          case 8: go(conv_stream_bf16_kernel<2, 4, false>, st::Cfg<2, 4, false>::LDS_BYTES); break;
          default: go(conv_stream_bf16_kernel<2, 4, true>, st::Cfg<2, 4, true>::LDS_BYTES); break;
        }
        if (prof) sr::prof_end(stream);
        if (rc) return rc;
        SR_CHECK_LAUNCH("conv_stream_bf16 launch");
        return SR_OK;
      }
    }
  }
  // Tile shape by launch size and depth.  Launches of >= 256 tiles with Cin <= 256 (the dense blocks, the head) use 16-row
  // tiles on 8 waves: 57 KB of LDS, so two workgroups share a CU and one's prologue / epilogue (no loads in flight for the
  // MFMA loop) overlaps the other's loop — 6-25 % faster than one 32-row workgroup per CU.  Deep layers (Cin > 256: the
  // discriminator's inner levels) are MFMA-heavier per byte and keep the 32-row tile (less halo and weight refill per
  // MFMA).  Smaller inputs (single plate crops, 32x32 training patches) fall to 4-wave tiles of 16, 8 and 4 rows, which
  // cost more halo / weight refill per MFMA but keep CUs from idling while a few workgroups walk K serially.
  const int conc = sr::launch_concurrency();  // image groups launching side by side (grouped forward)
  auto wgs = [&](int rows) { return (long long)sr::cdiv(p.W, 32) * sr::cdiv(p.H, rows) * d->n * groups * conc; };
  enum Tile { R16_W8, R32_W8, R16_W4, R8_W4, R4_W4 } tile;
  if (p.H % 16 == 0 && p.cin_blocks <= 16 && wgs(16) >= 256)
    tile = R16_W8;
  else if (w8 && wgs(32) >= 256)
    tile = R32_W8;
  else if (p.H % 16 == 0 && wgs(16) >= 256)
    tile = R16_W4;
  else if (wgs(8) >= 256 || p.H <= 4)
    tile = R8_W4;
  else
    tile = R4_W4;
  if (gc == 64) {
    if (d->s2_channels > 0 && (tile == R16_W8 || tile == R32_W8 || tile == R8_W4 || tile == R16_W4)) {  // the strided convs of the discriminators
      p.s2_cpb = d->s2_channels / 16;
      p.s2_side = d->s2_side;
      // (also on the 4-wave tiles: at patch size these layers are 8x8 and 4x4 maps of 2048 channels, and 5 of the 9 taps of every
      // channel block are the zeros of the 3x3 embedding)
      switch (tile) {
        case R16_W8: return launch_h<2, 2, 8, false, true>(p, d->n, groups, stream, d);
        case R32_W8: return launch_h<2, 4, 8, false, true>(p, d->n, groups, stream, d);
        case R16_W4: return launch_h<2, 4, 4, false, true>(p, d->n, groups, stream, d);
        default: return launch_h<2, 2, 4, false, true>(p, d->n, groups, stream, d);
      }
    }
    switch (tile) {
      case R16_W8: return launch_h<2, 2, 8, false>(p, d->n, groups, stream, d);
      case R32_W8: return launch_h<2, 4, 8, false>(p, d->n, groups, stream, d);
      case R16_W4: return launch_h<2, 4, 4, false>(p, d->n, groups, stream, d);
      case R8_W4: return launch_h<2, 2, 4, false>(p, d->n, groups, stream, d);
      default: return launch_h<2, 1, 4, false>(p, d->n, groups, stream, d);
    }
  }
  switch (tile) {
    case R16_W8: return launch_h<1, 2, 8, false>(p, d->n, groups, stream, d);
    case R32_W8: return launch_h<1, 4, 8, false>(p, d->n, groups, stream, d);
    case R16_W4: return launch_h<1, 4, 4, false>(p, d->n, groups, stream, d);
    case R8_W4: return launch_h<1, 2, 4, false>(p, d->n, groups, stream, d);
    default: return launch_h<1, 1, 4, false>(p, d->n, groups, stream, d);
  }
}
