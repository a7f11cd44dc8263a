// Weight gradient of the fused 3x3 convolution on fp32 MFMA (gfx950).
//
// Autograd counterpart of nn.Conv2d's weight/bias gradient on the RRDBNet path
// (rrdbnet_arch.py:21-25,94-101 through ESRGANModel.optimize_parameters' backward calls,
// esrgan_model.py:47,68,72):   dW[co][ci][tap] = sum_p dY[co][p] * X[ci][p + tap],   db[co] = sum_p dY[co][p].
//
// GEMM view: D[cout][cin] (per tap) += A[cout][k] * B[k][cin] with k = PIXELS, on
// v_mfma_f32_32x32x2_f32.  One wave owns one (32-cout tile, 32-cin tile) pair for all 9 taps
// (9 x 16 accumulator registers) and walks pixels; a workgroup is 4 waves = P tile pairs x KS
// row-splits (P*KS = 4) that share the staged rows.  A workgroup walks DOWN a 32-pixel-wide
// column strip: per step it consumes R = KS output rows; input rows live in an LDS ring of
// 2R+2 rows ([34 px][32*IT ch], pixel-major so a lane's operand is one conflict-free
// ds_read_b32), dY rows in a 2R-row ring ([32 px][32*CT]).  Rows arrive by LDS-DMA
// (global_load_lds_dwordx4, per-lane source address out of the CB8 planes; halo / ragged /
// channel padding = the zero line; the head's nearest x2 upsample = src >> 1).
// Partial sums leave the workgroup once (deterministic slab, no atomics); a second kernel sums
// the slab over workgroups, applies the scale and scatters to OIHW.
#include <algorithm>

#include "sr_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {


struct WgradParams {
  const void* zero;  // sr::zero_line()
  const float* x;   // forward source activation, CB8
  const float* dy;  // gradient wrt the conv's (pre-activation) output, CB8
  float* slab;      // [splits][pairs_in_launch][9][1024]
  float* bslab;     // [splits][CT][32] or null
  long long x_ns, dy_ns;
  int x_h, x_w;     // source spatial size
  int H, W;         // output spatial size
  int cin_blocks;   // valid channel blocks of x
  int cout_blocks;  // valid channel blocks of dy
  int cin_tile0;    // first 32-channel cin tile of this launch
  int cout_tile0;   // first 32-channel cout tile of this launch
  int gi;           // cin groups per cout-group row: group blockIdx.y = (row blockIdx.y / gi, column blockIdx.y % gi)
  int strips;       // column strips per image = ceil(W/32)
  int rows_per_wg;  // multiple of R
  int row_splits;   // ceil(H / rows_per_wg)
  // source map (same convention as the forward kernel, conv_f32.hip): tap (ty,tx) of output pixel (y,x) reads
  // virtual pixel (y+tap_oy+ty, x+tap_ox+tx), valid in [0,vH)x[0,vW), real = ((v*src_mul)>>src_shift)+src_o.
  int vH, vW, src_mul, src_shift, src_oy, src_ox, tap_oy, tap_ox;
};

__device__ __forceinline__ void glds16w(const float* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// R = output rows per step.  R == KS: the KS waves of a pair split the step's rows;
// R == 1 (with KS > 1): they split the 16 pixel-pair k-steps of the single row.
// LDS-DMA through a buffer descriptor (kept in a __device__ function: called from the kernel body directly, the host pass of
// hipcc drops the kernel's stub)
__device__ __forceinline__ void blds16f(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, char* lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

// The body of one workgroup: tile group `by` of the launch, column strip / row range `bx` of `nbx`.  A device function so that
// the single-conv kernel and the dense-block kernel (all weight gradients of a residual dense block in ONE launch, below) share it;
// the parameter block travels by value (conv_f32.hip: by reference the compiler re-reads fields from the kernel-argument segment).
template <int CT, int IT, int R, int KT>
__device__ __forceinline__ void wgrad_f32_body(const WgradParams p, char* smem, const int bx, const int by, const int nbx) {
  constexpr int P = CT * IT, KS = 4 / P;
  static_assert(R == KS || R == 1, "row split or k-step split");
  constexpr bool ROWSPLIT = (R == KS);
  constexpr int SN = ROWSPLIT ? 16 : 16 / KS;  // k-steps per wave per row
  constexpr int XCH = 32 * IT, YCH = 32 * CT;
  constexpr int XPIX = 32 + KT - 1;                            // pixels of one X row (with halo)
  constexpr int XPIECES = XPIX * (XCH / 4);                    // 16-byte pieces of one X row
  constexpr int XUNITS = (XPIECES + 63) / 64;                  // wave-instructions per X row
  constexpr int XROWB = XUNITS * 1024;                         // bytes per X ring slot
  constexpr int YUNITS = (32 * (YCH / 4)) / 64;                // = 4*CT
  constexpr int YROWB = YUNITS * 1024;
  constexpr int NXR = 2 * R + KT - 1, NYR = 2 * R;             // ring depths (rows)
  constexpr int XRING = NXR * XROWB;
  constexpr int UNITS_PER_STEP = R * (XUNITS + YUNITS);
  constexpr int UPW = (UNITS_PER_STEP + 3) / 4;                // units per wave per step

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int ct = wave / (IT * KS), it = (wave / KS) % IT, ks = wave % KS;
  // one launch covers a grid of same-shaped (CT x IT) tile groups: blockIdx.y walks them
  const int grow = by / p.gi, gcol = by - grow * p.gi;
  const int cout_tile0 = p.cout_tile0 + grow * CT, cin_tile0 = p.cin_tile0 + gcol * IT;

  int t = bx;
  const int rs = t % p.row_splits;
  t /= p.row_splits;
  const int strip = t % p.strips;
  const int n = t / p.strips;
  const int x0 = strip * 32;
  const int y_begin = rs * p.rows_per_wg;
  const int y_end = min(y_begin + p.rows_per_wg, p.H);

  const float* xn = p.x + (long long)n * p.x_ns;
  const float* dyn = p.dy + (long long)n * p.dy_ns;
  const long long xplane = (long long)p.x_h * p.x_w * 8, yplane = (long long)p.H * p.W * 8;
  char* xring = smem;
  char* yring = smem + XRING;

  // Stage the rows a step needs: X tap-rows u in [ux, ux+R) (virtual row u + tap_oy, ring slot u % NXR) and
  // dY rows [yy, yy+R).  Units (1 KiB wave-instructions) are dealt round-robin to the 4 waves.  Which unit of which row a wave's
  // uu-th load moves never changes, so its per-lane byte offset (channel block, column, half; beyond the buffer for lanes outside
  // the image) is computed once, and a step only adds the row as the scalar offset of a buffer-descriptor LDS-DMA
  // (conv_f32.hip); a descriptor with num_records = 0 delivers the zero rows.
  const __amdgpu_buffer_rsrc_t x_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (unsigned)((long long)p.cin_blocks * xplane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t y_rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)dyn, 0, (unsigned)((long long)p.cout_blocks * yplane * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t null_rs = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, 0, 0x00020000);
  unsigned lane_off[UPW];
#pragma unroll
  for (int uu = 0; uu < UPW; ++uu) {
    const int u = uu * 4 + wave;
    const int v = u % (XUNITS + YUNITS);
    unsigned off = 0xfffffff0u;
    if (v < XUNITS) {
      const int q = v * 64 + lane;
      const int pix = q / (XCH / 4), c4 = q % (XCH / 4);
      const int cb = cin_tile0 * 4 + (c4 >> 1);
      const int gx = x0 + p.tap_ox + pix;
      if (q < XPIECES && gx >= 0 && gx < p.vW && cb < p.cin_blocks) {
        const int sx = ((gx * p.src_mul) >> p.src_shift) + p.src_ox;
        off = (unsigned)((cb * xplane + (long long)sx * 8 + (c4 & 1) * 4) * 4);
      }
    } else {
      const int q = (v - XUNITS) * 64 + lane;
      const int pix = q / (YCH / 4), c4 = q % (YCH / 4);
      const int cb = cout_tile0 * 4 + (c4 >> 1);
      const int gx = x0 + pix;
      if (gx < p.W && cb < p.cout_blocks) off = (unsigned)((cb * yplane + (long long)gx * 8 + (c4 & 1) * 4) * 4);
    }
    lane_off[uu] = off;
  }
  auto stage = [&](int ux, int yy, bool with_dy) {
#pragma unroll
    for (int uu = 0; uu < UPW; ++uu) {
      const int u = uu * 4 + wave;
      if (u >= UNITS_PER_STEP) break;
      const int r = u / (XUNITS + YUNITS), v = u % (XUNITS + YUNITS);
      if (v >= XUNITS && !with_dy) continue;
      if (v < XUNITS) {
        const int ur = ux + r;
        const int vy = ur + p.tap_oy;  // virtual source row; outside [0, vH) = zero row
        const bool ok = vy >= 0 && vy < p.vH;
        const int sy = ((vy * p.src_mul) >> p.src_shift) + p.src_oy;
        blds16f(ok ? x_rs : null_rs, lane_off[uu], ok ? (unsigned)sy * (unsigned)p.x_w * 32u : 0u, xring + (ur % NXR) * XROWB + v * 1024);
      } else {
        const int y = yy + r;
        const bool ok = y < p.H;
        blds16f(ok ? y_rs : null_rs, lane_off[uu], ok ? (unsigned)y * (unsigned)p.W * 32u : 0u, yring + (y % NYR) * YROWB + (v - XUNITS) * 1024);
      }
    }
  };

  f32x16 acc[KT * KT];
#pragma unroll
  for (int a = 0; a < KT * KT; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float bsum = 0.f;

  const int a_lane = (h * YCH + ct * 32 + j) * 4;  // byte offset of A operand: pixel h of the pair, this lane's cout
  const int b_lane = (h * XCH + it * 32 + j) * 4;  // byte offset of B operand: pixel h (+dx), this lane's cin

  // Invariant at the top of step y (tap-rows u = y + ty): X rows u in [y, y+KT-1+R) and dY rows [y, y+R) are in LDS.
  // During step y the rows of step y+R arrive: u in [y+KT-1+R, y+KT-1+2R), dY rows [y+R, y+2R).
  // Live tap-rows [y, y+KT-1+2R) are NXR consecutive rows = distinct ring slots; dY rows [y, y+2R) likewise.
  stage(y_begin, y_begin, true);
  for (int u0 = y_begin + R; u0 < y_begin + KT - 1 + R; u0 += R) stage(u0, 0, false);
  __syncthreads();
  for (int y = y_begin; y < y_end; y += R) {
    stage(y + KT - 1 + R, y + R, true);
    const int row = ROWSPLIT ? y + ks : y;
    const int s0 = ROWSPLIT ? 0 : ks * SN;
    if (row < y_end) {
      const char* ya = yring + (row % NYR) * YROWB + a_lane + s0 * 2 * YCH * 4;
      const char* xb[KT];
#pragma unroll
      for (int dy = 0; dy < KT; ++dy) xb[dy] = xring + ((row + dy) % NXR) * XROWB + b_lane + s0 * 2 * XCH * 4;
      // Software pipeline over the pixel-pair k-steps: the operands of step s+1 (1 dY value, KT*(KT-1) new X values —
      // the dx = 0 column of step s+1 is the dx = 2 column of step s) are loaded BEFORE the KT*KT MFMAs of step s,
      // so their LDS latency hides behind 576 MFMA cycles instead of stalling every few MFMAs (2 waves per SIMD only).
      float a_cur, b_cur[KT][KT];
      a_cur = *(const float*)ya;
#pragma unroll
      for (int dy = 0; dy < KT; ++dy)
#pragma unroll
        for (int dx = 0; dx < KT; ++dx) b_cur[dy][dx] = *(const float*)(xb[dy] + dx * XCH * 4);
#pragma unroll
      for (int s = 0; s < SN; ++s) {
        float a_nxt = 0.f, b_nxt[KT][KT];
        if (s + 1 < SN) {
          a_nxt = *(const float*)(ya + (s + 1) * 2 * YCH * 4);
#pragma unroll
          for (int dy = 0; dy < KT; ++dy) {
            if (KT == 3) {
              b_nxt[dy][0] = b_cur[dy][2];
#pragma unroll
              for (int dx = 1; dx < KT; ++dx) b_nxt[dy][dx] = *(const float*)(xb[dy] + ((s + 1) * 2 + dx) * XCH * 4);
            } else {
#pragma unroll
              for (int dx = 0; dx < KT; ++dx) b_nxt[dy][dx] = *(const float*)(xb[dy] + ((s + 1) * 2 + dx) * XCH * 4);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above the MFMA block (hipcc otherwise sinks the loads)
        bsum += a_cur;
#pragma unroll
        for (int dy = 0; dy < KT; ++dy)
#pragma unroll
          for (int dx = 0; dx < KT; ++dx)
            acc[dy * KT + dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[dy][dx], acc[dy * KT + dx], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < SN) {
          a_cur = a_nxt;
#pragma unroll
          for (int dy = 0; dy < KT; ++dy)
#pragma unroll
            for (int dx = 0; dx < KT; ++dx) b_cur[dy][dx] = b_nxt[dy][dx];
        }
      }
    }
    __syncthreads();
  }

  // write this wave's partial tile:  slab[split][pair][tap][g][lane][4]
  const int pair = ct * IT + it;
  // slab[group][split][pair][tap][1024]; bias partials per cout-group ROW: bslab[row][split][CT][32] (column 0 writes)
  const long long nsplit = (long long)nbx * KS;
  const long long split = (long long)bx * KS + ks;
  float* dst = p.slab + (((by * nsplit + split) * P + pair) * (KT * KT)) * 1024 + lane * 4;
#pragma unroll
  for (int tap = 0; tap < KT * KT; ++tap)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[tap][g * 4 + e];
      *(f32x4*)(dst + tap * 1024 + g * 256) = v;
    }
  if (p.bslab && it == 0 && gcol == 0) {
    bsum += __shfl_xor(bsum, 32);
    if (h == 0) p.bslab[((grow * nsplit + split) * CT + ct) * 32 + j] = bsum;
  }
}

template <int CT, int IT, int R, int KT>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  wgrad_f32_body<CT, IT, R, KT>(p, smem, blockIdx.x, blockIdx.y, gridDim.x);
}

// Every weight gradient of a residual dense block (rrdbnet_arch.py:21-25: five convs = up to kMaxSub same-shaped tile-group sets of
// four shapes) as ONE launch: blockIdx.y walks the tile groups of all sets, blockIdx.x the column strips / row ranges (the same
// for every set).  The sets are what run_groups<3> would launch one by one — each followed by its own two reduction launches and
// each cut into >= 512 workgroups so that it fills the chip ALONE.  Together they fill it with a quarter of the row splits, i.e. a
// quarter of the partial tiles written to and read back from the slab (which, not the MFMAs, is what a weight gradient of a 32x32
// patch costs), and one table-driven reduction serves all sets.
constexpr int kMaxSub = 12;
struct WgradMulti {
  WgradParams sub[kMaxSub];
  int y0[kMaxSub + 1];   // first blockIdx.y of each set
  int variant[kMaxSub];  // 0: <2,2>  1: <2,1>  2: <1,4>  3: <1,2>  4: <1,1>
  int nsub;
};
__global__ __launch_bounds__(256) void wgrad_f32_rdb_kernel(const WgradMulti m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int s = 0;
  while (s + 1 < m.nsub && (int)blockIdx.y >= m.y0[s + 1]) ++s;
  const int by = blockIdx.y - m.y0[s];
  switch (m.variant[s]) {
    case 0: wgrad_f32_body<2, 2, 1, 3>(m.sub[s], smem, blockIdx.x, by, gridDim.x); break;
    case 1: wgrad_f32_body<2, 1, 1, 3>(m.sub[s], smem, blockIdx.x, by, gridDim.x); break;
    case 2: wgrad_f32_body<1, 4, 1, 3>(m.sub[s], smem, blockIdx.x, by, gridDim.x); break;
    case 3: wgrad_f32_body<1, 2, 1, 3>(m.sub[s], smem, blockIdx.x, by, gridDim.x); break;
    default: wgrad_f32_body<1, 1, 1, 3>(m.sub[s], smem, blockIdx.x, by, gridDim.x); break;
  }
}

// Slab reduction, two deterministic stages.
// Stage 1: grid (E4/256, SCH): block (x, sc) sums the splits of chunk sc for 256 float4 columns
//          (coalesced 4 KiB rows, 8 independent loads in flight per lane) into part[sc][E].
// Stage 2: one thread per element sums the SCH partials in fixed order, applies the scale and scatters to
//          OIHW.  Element (pair, tap, g, lane, e): cout = ct*32 + 8g + 4(lane>>5) + e, cin position = it*32 + (lane&31).
__device__ __forceinline__ void wgrad_reduce1_body(const float4* __restrict__ slab, float4* __restrict__ part, int e4, int splits,
                                                   int chunk, const float* __restrict__ bslab, float* __restrict__ bpart, int nb,
                                                   long long stride4, int bstride, int gi, int grp, const int sch) {
  // grp = tile group of a multi-group launch: its slab / part blocks follow each other; bias per group row
  slab += (long long)grp * splits * stride4;
  part += (long long)grp * sch * e4;
  if (bslab) {
    bslab += (long long)(grp / gi) * splits * bstride;
    bpart += (long long)(grp / gi) * sch * nb;
  }
  const int s0 = blockIdx.y * chunk, s1 = min(s0 + chunk, splits);
  if (blockIdx.x == (unsigned)(e4 + 255) / 256) {  // the extra block column reduces the bias partials of this chunk
    if (bslab && grp % gi == 0 && (int)threadIdx.x < nb) {
      float b = 0.f;
      int s = s0;
      for (; s + 8 <= s1; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = bslab[(long long)(s + u) * bstride + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 8; ++u) b += v[u];
      }
      for (; s < s1; ++s) b += bslab[(long long)s * bstride + threadIdx.x];
      bpart[blockIdx.y * nb + threadIdx.x] = b;
    }
    return;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= e4) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* p = slab + (long long)s0 * stride4 + i;
  int s = s0;
  for (; s + 8 <= s1; s += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long long)u * stride4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc.x += v[u].x;
      acc.y += v[u].y;
      acc.z += v[u].z;
      acc.w += v[u].w;
    }
    p += (long long)8 * stride4;
  }
  for (; s < s1; ++s) {
    const float4 v = *p;
    acc.x += v.x;
    acc.y += v.y;
    acc.z += v.z;
    acc.w += v.w;
    p += stride4;
  }
  part[(long long)blockIdx.y * e4 + i] = acc;
}

__global__ __launch_bounds__(256) void wgrad_reduce1_kernel(const float4* __restrict__ slab, float4* __restrict__ part,
                                                            int e4, int splits, int chunk,
                                                            const float* __restrict__ bslab, float* __restrict__ bpart,
                                                            int nb, long long stride4, int bstride, int gi) {
  wgrad_reduce1_body(slab, part, e4, splits, chunk, bslab, bpart, nb, stride4, bstride, gi, blockIdx.z, gridDim.y);
}

// Several independent reductions (the rows of a dense-block weight-gradient launch) in one launch: blockIdx.z = row.
constexpr int kMaxReduceRows = 8;
struct Reduce1Row {
  const float4* slab;
  float4* part;
  const float* bslab;
  float* bpart;
  long long stride4;
  int e4, nb, bstride;
};
struct Reduce1Rows {
  Reduce1Row row[kMaxReduceRows];
  int splits, chunk;
};
__global__ __launch_bounds__(256) void wgrad_reduce1_rows_kernel(const Reduce1Rows p) {
  const Reduce1Row& r = p.row[blockIdx.z];
  if ((int)blockIdx.x > (r.e4 + 255) / 256) return;  // the grid is sized for the widest row (+ its bias column)
  if ((int)blockIdx.x == (r.e4 + 255) / 256) {        // this row's bias column: same code path as the last column below
    if (r.bslab && (int)threadIdx.x < r.nb) {
      const int s0 = blockIdx.y * p.chunk, s1 = min(s0 + p.chunk, p.splits);
      float b = 0.f;
      for (int s = s0; s < s1; ++s) b += r.bslab[(long long)s * r.bstride + threadIdx.x];
      r.bpart[blockIdx.y * r.nb + threadIdx.x] = b;
    }
    return;
  }
  wgrad_reduce1_body(r.slab, r.part, r.e4, p.splits, p.chunk, nullptr, nullptr, r.nb, r.stride4, r.bstride, 1, 0, gridDim.y);
}

struct ReduceParams {
  const float* part;   // [sch][P*9*1024]
  const float* bpart;  // [sch][CT*32]
  float* dw;           // [cout][cin][3][3]
  float* db;           // [cout] or null
  int sch, splits, P, IT, CT;
  int gi;                      // cin groups per cout-group row of a multi-group launch (blockIdx.y = group)
  int ntap, ks;                // taps per pair (ks*ks) and tap-grid width
  int kdim, t_mul, dy_off, dx_off;  // kernel position of tap (ty,tx): (ty*t_mul+dy_off, tx*t_mul+dx_off) in a kdim x kdim kernel
  int cin_tile0, cout_tile0;
  int cout, cin, first_seg, seg;  // reference channel counts and concat segmentation
  int seg_pad;                    // concat segments start on multiples of this many channels (8: CB8, 16: CB16)
  float scale;
  int accumulate;
};

__device__ __forceinline__ void wgrad_reduce2_body(const ReduceParams& p, int grp) {
  // One thread per element (coalesced partial reads, many independent threads).  A form with one thread per (cout quad, cin position)
  // and a tap loop — float4 reads, 36-byte runs of the OIHW gradient per thread — was measured slower (round 4: 8.8 -> 27.9 us per
  // single-set launch, 34.5 -> 44.5 us per dense-block table): P x 256 threads with 9 x sch dependent loads each are latency-bound.
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int per_split = p.P * p.ntap * 1024;
  const int grow = grp / p.gi, gcol = grp - grow * p.gi;
  const int cout_tile0 = p.cout_tile0 + grow * p.CT, cin_tile0 = p.cin_tile0 + gcol * p.IT;
  if (idx < per_split) {
    float s = 0.f;
    const float* part = p.part + (long long)grp * p.sch * per_split;
    for (int k = 0; k < p.sch; ++k) s += part[(long long)k * per_split + idx];
    const int e = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) & 3;
    const int tap = (idx >> 10) % p.ntap, pair = idx / (p.ntap * 1024);
    const int ct = pair / p.IT, it = pair % p.IT;
    const int co = (cout_tile0 + ct) * 32 + 8 * g + 4 * (lane >> 5) + e;
    const int pos = (cin_tile0 + it) * 32 + (lane & 31);
    // invert the concat position map of sr_conv3x3_pack_f32
    int ci = -1;
    const int fsp = (p.first_seg + p.seg_pad - 1) / p.seg_pad * p.seg_pad;
    if (pos < fsp) {
      if (pos < p.first_seg) ci = pos;
    } else if (p.seg > 0) {
      const int sp = (p.seg + p.seg_pad - 1) / p.seg_pad * p.seg_pad;
      const int r = pos - fsp, sgi = r / sp, o = r % sp;
      if (o < p.seg) ci = p.first_seg + sgi * p.seg + o;
    }
    if (co < p.cout && ci >= 0 && ci < p.cin) {
      const int ky = (tap / p.ks) * p.t_mul + p.dy_off, kx = (tap % p.ks) * p.t_mul + p.dx_off;
      float* o = p.dw + (((long long)co * p.cin + ci) * p.kdim + ky) * p.kdim + kx;
      *o = p.accumulate ? *o + s * p.scale : s * p.scale;
    }
  }
  if (p.db && p.bpart && gcol == 0 && idx < p.CT * 32) {
    float s = 0.f;
    const float* bpart = p.bpart + (long long)grow * p.sch * p.CT * 32;
    for (int k = 0; k < p.sch; ++k) s += bpart[k * p.CT * 32 + idx];
    const int co = cout_tile0 * 32 + idx;
    if (co < p.cout) p.db[co] = p.accumulate ? p.db[co] + s * p.scale : s * p.scale;
  }
}

__global__ void wgrad_reduce2_kernel(const ReduceParams p) { wgrad_reduce2_body(p, blockIdx.y); }

struct Reduce2Rows {
  ReduceParams row[kMaxReduceRows];
};
__global__ void wgrad_reduce2_rows_kernel(const Reduce2Rows p) { wgrad_reduce2_body(p.row[blockIdx.y], 0); }

// Table-driven form for sets that differ in everything (split count, tile shape, group count): blockIdx.z / blockIdx.y walks the
// tile groups of all sets (z0 = first group of a set).  Per element the same sums in the same order as the one-set kernels.
struct TableRow1 {
  const float4* slab;
  float4* part;
  const float* bslab;
  float* bpart;
  long long stride4;
  int e4, nb, bstride, splits, chunk, sch, gi, z0;
};
struct ReduceTable1 {
  TableRow1 row[kMaxSub];
  int nrows;
};
__global__ __launch_bounds__(256) void wgrad_reduce1_table_kernel(const ReduceTable1 t) {
  int r = 0;
  while (r + 1 < t.nrows && (int)blockIdx.z >= t.row[r + 1].z0) ++r;
  const TableRow1 q = t.row[r];
  if ((int)blockIdx.y >= q.sch || (int)blockIdx.x > (q.e4 + 255) / 256) return;  // the grid is sized for the widest / deepest set
  wgrad_reduce1_body(q.slab, q.part, q.e4, q.splits, q.chunk, q.bslab, q.bpart, q.nb, q.stride4, q.bstride, q.gi, blockIdx.z - q.z0, q.sch);
}
struct ReduceTable2 {
  ReduceParams row[kMaxSub];
  int z0[kMaxSub + 1];
  int nrows;
};
__global__ void wgrad_reduce2_table_kernel(const ReduceTable2 t) {
  int r = 0;
  while (r + 1 < t.nrows && (int)blockIdx.y >= t.z0[r + 1]) ++r;
  wgrad_reduce2_body(t.row[r], blockIdx.y - t.z0[r]);
}

template <int CT, int IT, int R, int KS>
constexpr int wgrad_lds_bytes() {
  constexpr int XUNITS = ((32 + KS - 1) * (32 * IT / 4) + 63) / 64, YUNITS = 4 * CT;
  return (2 * R + KS - 1) * XUNITS * 1024 + 2 * R * YUNITS * 1024;
}

struct TapMap {
  int kdim, t_mul, dy_off, dx_off;
};

int g_wgrad_target_wgs = 512;  // development switch (sr_dev_set_wgrad_f32_target)

// Collects what run_groups<3> would launch (sr::rdb_wgrad_f32 below) instead of launching it.
struct RdbCollector {
  bool dry = true;          // sizes only
  int rows = 0;             // rows per workgroup of every set (0: the whole image height)
  char* arena = nullptr;    // slab + partial buffers of all sets
  size_t arena_bytes = 0, used = 0;
  WgradMulti m = {};
  sr::WgradReduce red[kMaxSub];
  int sch[kMaxSub], chunk[kMaxSub];
  int lds = 0, total_groups = 0;
  long long nbx = 0, strips_total = 0;
  char* take(size_t bytes) {
    char* p = arena ? arena + used : nullptr;
    used += sr::align_up(bytes, 256);
    return p;
  }
};
thread_local RdbCollector* g_collect = nullptr;

template <int CT, int IT>
constexpr int rdb_variant() {
  return CT == 2 ? (IT == 2 ? 0 : 1) : (IT == 4 ? 2 : (IT == 2 ? 3 : 4));
}
template <int CT, int IT, int R, int KT>
int collect_group(const sr_conv3x3_wgrad_desc* d, WgradParams p, int cout_tile0, int cin_tile0, int grows, int gi, bool want_bias,
                  const TapMap& tm) {
  RdbCollector& c = *g_collect;
  constexpr int P = CT * IT, KS = 4 / P, NT = KT * KT;
  if (KT != 3 || R != 1 || c.m.nsub >= kMaxSub || !((CT == 2 && IT <= 2) || (CT == 1 && (IT == 4 || IT <= 2))))
    return 1;  // (positive: not an error) more tile-group sets than one launch carries — the caller runs the convs one by one
  const int groups = grows * gi, i = c.m.nsub;
  p.cin_tile0 = cin_tile0;
  p.cout_tile0 = cout_tile0;
  p.gi = gi;
  const int rows = c.rows > 0 ? std::min(c.rows, p.H) : p.H;
  p.rows_per_wg = rows;
  p.row_splits = sr::cdiv(p.H, rows);
  c.strips_total = (long long)d->n * p.strips;
  const long long nwg = c.strips_total * p.row_splits, splits = nwg * KS;
  if (c.nbx && c.nbx != nwg) {
    sr::set_error("rdb_wgrad_f32: sets of one launch must share the strip grid");
    return SR_EINVAL;
  }
  c.nbx = nwg;
  int sch = (int)std::min<long long>(64, (splits + 15) / 16);
  const int chunk = (int)((splits + sch - 1) / sch);
  sch = (int)((splits + chunk - 1) / chunk);
  p.slab = (float*)c.take((size_t)groups * splits * P * NT * 1024 * sizeof(float));
  p.bslab = want_bias ? (float*)c.take((size_t)grows * splits * CT * 32 * sizeof(float)) : nullptr;
  sr::WgradReduce& rr = c.red[i];
  rr = sr::WgradReduce{};
  rr.slab = p.slab;
  rr.bslab = p.bslab;
  rr.part = (float*)c.take((size_t)groups * sch * P * NT * 1024 * sizeof(float));
  rr.bpart = (float*)c.take((size_t)grows * sch * CT * 32 * sizeof(float));
  rr.splits = splits;
  rr.groups = groups;
  rr.gi = gi;
  rr.P = P;
  rr.IT = IT;
  rr.CT = CT;
  rr.ntap = NT;
  rr.ks = KT;
  rr.kdim = tm.kdim;
  rr.t_mul = tm.t_mul;
  rr.dy_off = tm.dy_off;
  rr.dx_off = tm.dx_off;
  rr.cin_tile0 = cin_tile0;
  rr.cout_tile0 = cout_tile0;
  rr.cout = d->cout;
  rr.cin = d->cin;
  rr.first_seg = d->first_seg;
  rr.seg = d->seg;
  rr.seg_pad = 8;
  rr.scale = d->scale;
  rr.accumulate = d->accumulate;
  rr.dw = d->dweight;
  rr.db = want_bias ? d->dbias : nullptr;
  c.sch[i] = sch;
  c.chunk[i] = chunk;
  c.m.sub[i] = p;
  c.m.variant[i] = rdb_variant<CT, IT>();
  c.m.y0[i] = c.total_groups;
  c.total_groups += groups;
  c.m.y0[i + 1] = c.total_groups;
  c.m.nsub = i + 1;
  c.lds = std::max(c.lds, wgrad_lds_bytes<CT, IT, R, KT>());
  return SR_OK;
}

template <int CT, int IT, int R, int KT>
int launch_group(const sr_conv3x3_wgrad_desc* d, WgradParams p, int cout_tile0, int cin_tile0, int grows, int gi, float* slab,
                 float* bslab, bool want_bias, const TapMap& tm, hipStream_t stream) {
  // One launch covers grows x gi same-shaped tile groups (grid.y), starting at (cout_tile0, cin_tile0), and ONE slab
  // reduction: wide convs (the 512-channel discriminator layers: 64 groups x 4 parity passes) used to cost a launch
  // and two reduce launches per group.
  constexpr int P = CT * IT, KS = 4 / P, NT = KT * KT;
  constexpr int lds = wgrad_lds_bytes<CT, IT, R, KT>();
  if (g_collect) return collect_group<CT, IT, R, KT>(d, p, cout_tile0, cin_tile0, grows, gi, want_bias, tm);
  const int groups = grows * gi;
  p.cin_tile0 = cin_tile0;
  p.cout_tile0 = cout_tile0;
  p.gi = gi;
  // rows per workgroup: aim at >= 512 workgroups (2 per CU), multiple of R, at least 4R rows to amortise the prologue
  const long long strips_total = (long long)d->n * p.strips;
  int rows = p.H;
  while (rows > 4 * R && strips_total * sr::cdiv(p.H, rows) * groups < g_wgrad_target_wgs) rows = (rows + 1) / 2;
  rows = (rows + R - 1) / R * R;
  p.rows_per_wg = rows;
  p.row_splits = sr::cdiv(p.H, rows);
  const long long nwg = strips_total * p.row_splits;
  const long long splits = nwg * KS;
  if ((size_t)groups * splits * P * NT * 1024 * sizeof(float) > d->slab_bytes ||
      (size_t)grows * splits * CT * 32 * sizeof(float) > d->slab_bytes / 63) {
    sr::set_error("sr_conv3x3_wgrad_f32: slab %zu B too small (need %zu B)", d->slab_bytes,
                  (size_t)groups * splits * P * NT * 1024 * sizeof(float));
    return SR_ENOSPACE;
  }
  p.slab = slab;
  p.bslab = want_bias ? bslab : nullptr;
  auto kern = wgrad_f32_kernel<CT, IT, R, KT>;
  if (int rc = sr::ensure_dynamic_lds((const void*)kern, lds)) return rc;  // once per (kernel, device)
  const bool prof = sr::prof_on();
  if (prof) {
    sr_launch_record r = {};
    r.kernel_id = 8 + (CT == 2 ? (IT == 2 ? 3 : 4) : (IT == 4 ? 2 : (IT == 2 ? 1 : 0)));
    r.cin = 32 * IT;
    r.cout = 32 * CT;
    r.n = d->n;
    r.h = p.H;
    r.w = p.W;
    const double px = (double)d->n * p.H * p.W;
    const int cin_eff = min(32 * IT, d->cin_pad - 32 * cin_tile0), cout_eff = min(32 * CT, d->cout - 32 * cout_tile0);
    r.flops = 2.0 * NT * cin_eff * cout_eff * px * groups;
    r.bytes = 4.0 * px * (cin_eff + cout_eff) * groups;
    sr::prof_begin(stream, r);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg, (unsigned)groups), dim3(256), lds, stream, p);
  if (prof) sr::prof_end(stream);
  SR_CHECK_LAUNCH("wgrad3x3_f32 launch");
  // stage 1 partials live behind the bias slab
  sr::WgradReduce rr = {};
  rr.slab = slab;
  rr.bslab = want_bias ? bslab : nullptr;
  rr.part = (float*)((char*)bslab + d->slab_bytes / 63 / 256 * 256);
  rr.bpart = rr.part + (size_t)64 * 4 * 9 * 1024;  // behind the weight partials
  rr.splits = splits;
  rr.groups = groups;
  rr.gi = gi;
  rr.P = P;
  rr.IT = IT;
  rr.CT = CT;
  rr.ntap = NT;
  rr.ks = KT;
  rr.kdim = tm.kdim;
  rr.t_mul = tm.t_mul;
  rr.dy_off = tm.dy_off;
  rr.dx_off = tm.dx_off;
  rr.cin_tile0 = cin_tile0;
  rr.cout_tile0 = cout_tile0;
  rr.cout = d->cout;
  rr.cin = d->cin;
  rr.first_seg = d->first_seg;
  rr.seg = d->seg;
  rr.seg_pad = 8;
  rr.scale = d->scale;
  rr.accumulate = d->accumulate;
  rr.dw = d->dweight;
  rr.db = want_bias ? d->dbias : nullptr;
  return sr::wgrad_reduce(rr, stream);
}

}  // namespace

namespace {
ReduceParams reduce2_params(const sr::WgradReduce& r, int sch, int gi) {
  ReduceParams rp;
  rp.part = r.part;
  rp.bpart = r.bslab ? r.bpart : nullptr;
  rp.dw = r.dw;
  rp.db = r.bslab ? r.db : nullptr;
  rp.sch = sch;
  rp.splits = (int)r.splits;
  rp.P = r.P;
  rp.IT = r.IT;
  rp.CT = r.CT;
  rp.gi = gi;
  rp.ntap = r.ntap;
  rp.ks = r.ks;
  rp.kdim = r.kdim;
  rp.t_mul = r.t_mul;
  rp.dy_off = r.dy_off;
  rp.dx_off = r.dx_off;
  rp.cin_tile0 = r.cin_tile0;
  rp.cout_tile0 = r.cout_tile0;
  rp.cout = r.cout;
  rp.cin = r.cin;
  rp.first_seg = r.first_seg;
  rp.seg = r.seg;
  rp.seg_pad = r.seg_pad;
  rp.scale = r.scale;
  rp.accumulate = r.accumulate;
  return rp;
}
}  // namespace

namespace sr {
// Two-stage deterministic slab reduction + scatter to OIHW, shared by the fp32 and bf16 weight-gradient kernels
// (both leave fp32 partial tiles in the 32x32 MFMA accumulator layout).  part holds 64 * P*ntap*1024 floats.
int wgrad_reduce(const WgradReduce& r, hipStream_t stream) {
  const int e4 = r.P * r.ntap * 256;
  const int groups = r.groups > 0 ? r.groups : 1, gi = r.gi > 0 ? r.gi : 1;
  int sch = (int)((r.splits + 63) / 64);  // (finer chunks measured slower here: 72 -> 76 ms on the fp32 recipe step)
  // part holds 64 * 4*9*1024 floats and bpart 64*64: shrink the stage-1 fan-out of multi-group launches to fit
  int cap = (int)((size_t)64 * 4 * 9 * 1024 / ((size_t)groups * r.P * r.ntap * 1024));
  const int bcap = 4096 / ((groups / gi) * r.CT * 32);
  if (bcap < cap) cap = bcap;
  if (cap < 1) {
    set_error("wgrad_reduce: %d groups do not fit the partial buffers", groups);
    return SR_EINVAL;
  }
  if (cap > 64) cap = 64;
  if (sch > cap) sch = cap;
  const int chunk = (int)((r.splits + sch - 1) / sch);
  sch = (int)((r.splits + chunk - 1) / chunk);
  hipLaunchKernelGGL(wgrad_reduce1_kernel, dim3((e4 + 255) / 256 + 1, sch, groups), dim3(256), 0, stream, (const float4*)r.slab,
                     (float4*)r.part, e4, (int)r.splits, chunk, r.bslab, r.bpart, r.CT * 32,
                     r.split_stride ? r.split_stride / 4 : (long long)e4, r.bsplit_stride ? r.bsplit_stride : r.CT * 32, gi);
  SR_CHECK_LAUNCH("wgrad_reduce1 launch");
  const ReduceParams rp = reduce2_params(r, sch, gi);
  hipLaunchKernelGGL(wgrad_reduce2_kernel, dim3((r.P * r.ntap * 1024 + 255) / 256, groups), dim3(256), 0, stream, rp);
  SR_CHECK_LAUNCH("wgrad_reduce2 launch");
  return SR_OK;
}

// The same reduction for up to kMaxReduceRows independent single-group rows (equal splits) in two launches instead of
// 2 x nrows: rows[i].part / bpart are ignored, the rows share r[0].part / r[0].bpart, carved here.
int wgrad_reduce_rows(const WgradReduce* r, int nrows, hipStream_t stream) {
  if (nrows <= 0) return SR_OK;
  if (nrows > kMaxReduceRows) {
    set_error("wgrad_reduce_rows: %d rows > %d", nrows, kMaxReduceRows);
    return SR_EINVAL;
  }
  int sum_p = 0, max_e4 = 0, max_e = 0;
  for (int i = 0; i < nrows; ++i) {
    if (r[i].splits != r[0].splits || r[i].groups > 1) {
      set_error("wgrad_reduce_rows: rows must share the split count and be single-group");
      return SR_EINVAL;
    }
    sum_p += r[i].P * r[i].ntap;
    max_e4 = std::max(max_e4, r[i].P * r[i].ntap * 256);
    max_e = std::max(max_e, r[i].P * r[i].ntap * 1024);
  }
  int sch = (int)((r[0].splits + 15) / 16);  // (16 splits per stage-1 block: the dense block's 64-split launches were ONE chunk of eight dependent rounds of loads)
  int cap = (int)((size_t)64 * 4 * 9 / (size_t)sum_p);  // part holds 64 * 4*9*1024 floats, bpart 64*64
  cap = std::min(cap, 4096 / (nrows * 32));
  if (cap < 1) {
    set_error("wgrad_reduce_rows: %d rows do not fit the partial buffers", nrows);
    return SR_EINVAL;
  }
  sch = std::min(sch, std::min(cap, 64));
  const int chunk = (int)((r[0].splits + sch - 1) / sch);
  sch = (int)((r[0].splits + chunk - 1) / chunk);
  Reduce1Rows s1;
  Reduce2Rows s2;
  s1.splits = (int)r[0].splits;
  s1.chunk = chunk;
  float* part = r[0].part;
  float* bpart = r[0].bpart;
  for (int i = 0; i < kMaxReduceRows; ++i) {
    const WgradReduce& q = r[i < nrows ? i : 0];
    WgradReduce w = q;
    if (i < nrows) {
      w.part = part;
      w.bpart = bpart;
      part += (size_t)sch * q.P * q.ntap * 1024;
      bpart += (size_t)sch * q.CT * 32;
    }
    Reduce1Row& a = s1.row[i];
    a.slab = (const float4*)w.slab;
    a.part = (float4*)w.part;
    a.bslab = w.bslab;
    a.bpart = w.bpart;
    a.e4 = i < nrows ? w.P * w.ntap * 256 : 0;
    a.nb = w.CT * 32;
    a.stride4 = w.split_stride ? w.split_stride / 4 : (long long)(w.P * w.ntap * 256);
    a.bstride = w.bsplit_stride ? w.bsplit_stride : w.CT * 32;
    s2.row[i] = reduce2_params(w, sch, 1);
  }
  hipLaunchKernelGGL(wgrad_reduce1_rows_kernel, dim3((max_e4 + 255) / 256 + 1, sch, nrows), dim3(256), 0, stream, s1);
  SR_CHECK_LAUNCH("wgrad_reduce1_rows launch");
  hipLaunchKernelGGL(wgrad_reduce2_rows_kernel, dim3((max_e + 255) / 256, nrows), dim3(256), 0, stream, s2);
  SR_CHECK_LAUNCH("wgrad_reduce2_rows launch");
  return SR_OK;
}
}  // namespace sr

extern "C" size_t sr_conv3x3_wgrad_slab_bytes(int n, int h, int w) {
  // splits*P = 4 * workgroups for every launch shape; the row-range heuristic of launch_group stops
  // halving once it has >= 512 workgroups, so workgroups < max(strips, 1024).  1/64 of the slab is bias partials.
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  const long long strips = (long long)n * ((w + 31) / 32);
  const long long nwg = strips > 1024 ? strips : 1024;
  const size_t wbytes = (size_t)nwg * 4 * 9 * 1024 * sizeof(float);
  const size_t part_bytes = (size_t)64 * 4 * 9 * 1024 * sizeof(float) + 64 * 64 * sizeof(float) + 4096;
  return (wbytes + wbytes / 32 + 2 * part_bytes + 4096) / 256 * 256;
}

namespace {

// Walks the (cout tile, cin tile) grid in workgroup-sized groups (2x2, 1x4, 1x2, 2x1, 1x1) for one tap grid.
template <int KT>
int run_groups(const sr_conv3x3_wgrad_desc* d, const WgradParams& p, int cin_pad, const TapMap& tm, bool want_bias,
               hipStream_t stream) {
  const int cts = sr::cdiv(d->cout, 32), its = sr::cdiv(cin_pad, 32);
  const size_t part_bytes = (size_t)64 * 4 * 9 * 1024 * sizeof(float) + 64 * 64 * sizeof(float) + 4096;
  SR_CHECK_ARG(d->slab_bytes > 2 * part_bytes, "sr_conv_wgrad: slab too small");
  const size_t wslab_bytes = (d->slab_bytes - part_bytes) / 64 * 63 / 256 * 256;
  float* slab = (float*)d->slab;
  float* bslab = (float*)((char*)d->slab + wslab_bytes);
  sr_conv3x3_wgrad_desc dd = *d;
  dd.slab_bytes = wslab_bytes;
  // Same-shaped groups of the (cout tile, cin tile) grid go out as ONE launch each: all 2-row x 2-column groups, then
  // the odd cin column of those rows, then the odd cout row (4-, 2-, 1-column groups).  A launch is cut into row chunks
  // when its tiles would not fit the slab.
  const long long strips_total = (long long)d->n * p.strips;
  auto row_chunk = [&](int rows_left, int gi, int P_) {  // group rows per launch that fit slab + partial buffers
    // launch_group uses < 1024 workgroups in total while strips * groups <= 512, else strips workgroups per group
    const long long cap = (long long)(wslab_bytes / ((size_t)4 * 9 * 1024 * sizeof(float)));
    long long fit = cap / strips_total;
    if (fit < 1) fit = 1;
    long long rows = fit / gi;
    if (rows < 1) rows = 1;
    long long rows_part = 64 * 4 / ((long long)gi * P_);  // groups * P <= 256: at least one stage-1 chunk per group
    if (rows_part < 1) rows_part = 1;
    if (rows > rows_part) rows = rows_part;
    if (rows > 64) rows = 64;  // bias partials: rows * CT * 32 floats per chunk in a 4096-float buffer
    return (int)(rows < rows_left ? rows : rows_left);
  };
  const int rows2 = cts / 2;  // cout-group rows of two tiles
  for (int r0 = 0; r0 < rows2;) {
    const int gi = its / 2;
    const int nr = row_chunk(rows2 - r0, gi > 0 ? gi : 1, 4);
    const bool bias = want_bias && d->dbias != nullptr;
    if (gi > 0) {
      // groups of one launch that fit the slab; a row wider than that (many cin tiles x many strips) goes out in cin chunks
      const long long fit = (long long)(wslab_bytes / ((size_t)4 * 9 * 1024 * sizeof(float))) / strips_total;
      SR_CHECK_ARG(fit >= 1, "sr_conv_wgrad: slab too small for one tile group of %lld strips", strips_total);
      if (fit >= gi) {
        int rc = launch_group<2, 2, 1, KT>(&dd, p, 2 * r0, 0, nr, gi, slab, bslab, bias, tm, stream);
        if (rc) return rc;
      } else {
        for (int g0 = 0; g0 < gi; g0 += (int)fit) {  // nr == 1 here (row_chunk)
          const int gn = gi - g0 < (int)fit ? gi - g0 : (int)fit;
          int rc = launch_group<2, 2, 1, KT>(&dd, p, 2 * r0, 2 * g0, 1, gn, slab, bslab, bias && g0 == 0, tm, stream);
          if (rc) return rc;
        }
      }
    }
    if (its % 2) {
      int rc = launch_group<2, 1, 1, KT>(&dd, p, 2 * r0, its - 1, nr, 1, slab, bslab, bias && gi == 0, tm, stream);
      if (rc) return rc;
    }
    r0 += nr;
  }
  if (cts % 2) {
    const int c0 = cts - 1;
    int i0 = 0;
    const bool bias = want_bias && d->dbias != nullptr;
    if (its / 4 > 0) {
      int rc = launch_group<1, 4, 1, KT>(&dd, p, c0, 0, 1, its / 4, slab, bslab, bias, tm, stream);
      if (rc) return rc;
      i0 = its / 4 * 4;
    }
    if (its - i0 >= 2) {
      int rc = launch_group<1, 2, 1, KT>(&dd, p, c0, i0, 1, 1, slab, bslab, bias && i0 == 0, tm, stream);
      if (rc) return rc;
      i0 += 2;
    }
    if (its - i0 >= 1) {
      int rc = launch_group<1, 1, 1, KT>(&dd, p, c0, i0, 1, 1, slab, bslab, bias && i0 == 0, tm, stream);
      if (rc) return rc;
    }
  }
  return SR_OK;
}

int fill_wgrad(const sr_conv3x3_wgrad_desc* d, WgradParams* pp, const char* who) {
  SR_CHECK_ARG(d && d->x && d->dy && d->dweight && d->slab, "%s: null argument", who);
  SR_CHECK_ARG(d->cout > 0 && d->cin > 0 && d->n > 0 && d->in_h > 0 && d->in_w > 0, "%s: bad shape", who);
  SR_CHECK_ARG((long long)d->in_h * d->in_w * 4 * 4 * ((d->cout > d->cin_pad ? d->cout : d->cin_pad) + 7) < (1ll << 32),
               "%s: image too large for 32-bit buffer offsets", who);
  SR_CHECK_ARG(((uintptr_t)d->x | (uintptr_t)d->dy | (uintptr_t)d->slab) % 16 == 0, "%s: pointers must be 16-byte aligned",
               who);
  WgradParams& p = *pp;
  p = WgradParams{};
  p.zero = sr::zero_line();
  p.x = d->x;
  p.dy = d->dy;
  p.x_ns = d->x_img_stride;
  p.dy_ns = d->dy_img_stride;
  p.x_h = d->in_h;
  p.x_w = d->in_w;
  p.cout_blocks = (d->cout + 7) / 8;
  p.src_mul = 1;
  return SR_OK;
}

}  // namespace

namespace {
int g_rdb_wgrad_f32 = 1;          // development switch (sr_dev_set_rdb_wgrad_f32): 0 = one launch + one reduction per tile-group set
int g_rdb_wgrad_f32_target = 256; // workgroups of the dense-block launch the row split aims at: one per CU (whole 32x32 patches per
                                  // workgroup: 54.2 ms per recipe step against 55.9 at two row ranges per patch and 59.8 at four)

// Runs run_groups<3> for the five convs of a dense block with the collector installed: nothing is launched.
int rdb_collect(RdbCollector& c, const float* cat, const float* D, long long ns, int n, int h, int w, int nf, int gc,
                float* const* dparams, float scale5, int accumulate) {
  const int nfp = (nf + 7) / 8 * 8, gcp = (gc + 7) / 8 * 8;
  const long long hw = (long long)h * w;
  static float dummy;  // dry runs: fill_wgrad wants non-null pointers, nothing dereferences them
  struct Guard {
    RdbCollector* prev;
    ~Guard() { g_collect = prev; }
  } guard{g_collect};
  g_collect = &c;
  for (int k = 5; k >= 1; --k) {
    float* dw = dparams ? dparams[2 * (k - 1)] : &dummy;
    float* db = dparams ? dparams[2 * (k - 1) + 1] : &dummy;
    if (!dw) continue;  // frozen conv
    sr_conv3x3_wgrad_desc d = {};
    d.x = cat ? cat : &dummy;
    d.x_img_stride = ns;
    d.cin = nf + (k - 1) * gc;
    d.first_seg = nf;
    d.seg = gc;
    d.cin_pad = sr_conv3x3_cin_pad(d.cin, nf, gc);
    d.in_h = h;
    d.in_w = w;
    d.dy = D ? (k == 5 ? D : D + (long long)(nfp + (4 - k) * gcp) * hw) : &dummy;
    d.dy_img_stride = ns;
    d.cout = k == 5 ? nf : gc;
    d.n = n;
    d.scale = k == 5 ? scale5 : 1.f;
    d.dweight = dw;
    d.dbias = db;
    d.accumulate = accumulate;
    d.slab = &dummy;
    d.slab_bytes = (size_t)1 << 44;  // the collector's arena is what is checked
    WgradParams p;
    p = WgradParams{};
    p.zero = sr::zero_line();
    p.x = d.x;
    p.dy = d.dy;
    p.x_ns = d.x_img_stride;
    p.dy_ns = d.dy_img_stride;
    p.x_h = h;
    p.x_w = w;
    p.cout_blocks = (d.cout + 7) / 8;
    p.src_mul = 1;
    p.H = p.vH = h;
    p.W = p.vW = w;
    p.tap_oy = p.tap_ox = -1;
    p.cin_blocks = d.cin_pad / 8;
    p.strips = sr::cdiv(w, 32);
    const TapMap tm = {3, 1, 0, 0};
    if (int rc = run_groups<3>(&d, p, d.cin_pad, tm, true, nullptr)) return rc;
  }
  return SR_OK;
}

// Rows per workgroup: halve from the whole image while the launch has fewer workgroups than the target.
int rdb_rows(const RdbCollector& whole, int h) {
  int rows = h;
  while (rows > 4 && whole.strips_total * sr::cdiv(h, rows) * whole.total_groups < g_rdb_wgrad_f32_target) rows = (rows + 1) / 2;
  return rows;
}
}  // namespace

namespace sr {
bool rdb_wgrad_f32_enabled() { return g_rdb_wgrad_f32 != 0; }

size_t rdb_wgrad_slab_bytes_f32(int n, int h, int w, int nf, int gc) {
  if (n <= 0 || h <= 0 || w <= 0 || nf <= 0 || gc <= 0) return 0;
  RdbCollector whole;
  if (rdb_collect(whole, nullptr, nullptr, 0, n, h, w, nf, gc, nullptr, 1.f, 0)) return 0;
  RdbCollector c;
  c.rows = rdb_rows(whole, h);
  if (rdb_collect(c, nullptr, nullptr, 0, n, h, w, nf, gc, nullptr, 1.f, 0)) return 0;
  return c.used + 4096;
}

// The five weight gradients of one residual dense block (rrdbnet_arch.py:21-25) from its concat buffer `cat` = [x|x1..x4] and its
// gradient concat buffer D = [dY5|dY4|dY3|dY2|dY1] (both CB8, image stride ns) as ONE launch + ONE table-driven reduction.
// Returns 1 (nothing launched) when the block's widths need more than kMaxSub tile-group sets: the caller then issues
// sr_conv3x3_wgrad_f32 per conv.
// dparams[2k], [2k+1] = dweight / dbias of conv k+1 (null weight: skipped); conv5's gradient is scaled by scale5.  Per conv the
// same products as sr_conv3x3_wgrad_f32; the sums run over fewer, longer row ranges (deterministic, another rounding order).
int rdb_wgrad_f32(const float* cat, const float* D, long long ns, int n, int h, int w, int nf, int gc, float* const* dparams,
                  float scale5, int accumulate, void* slab, size_t slab_bytes, hipStream_t stream) {
  if (!cat || !D || !dparams || !slab || ((uintptr_t)cat | (uintptr_t)D | (uintptr_t)slab) % 16 != 0) {
    set_error("rdb_wgrad_f32: bad argument");
    return SR_EINVAL;
  }
  RdbCollector whole;
  if (int rc = rdb_collect(whole, cat, D, ns, n, h, w, nf, gc, dparams, scale5, accumulate)) return rc;  // 1: does not fit one launch
  if (whole.m.nsub == 0) return SR_OK;  // every conv frozen
  RdbCollector c;
  c.rows = rdb_rows(whole, h);
  c.dry = false;
  c.arena = (char*)slab;
  c.arena_bytes = slab_bytes;
  if (int rc = rdb_collect(c, cat, D, ns, n, h, w, nf, gc, dparams, scale5, accumulate)) return rc;
  if (c.used > slab_bytes) {
    set_error("rdb_wgrad_f32: slab %zu B too small (need %zu B)", slab_bytes, c.used);
    return SR_ENOSPACE;
  }
  auto kern = wgrad_f32_rdb_kernel;
  constexpr int lds_all = std::max({wgrad_lds_bytes<2, 2, 1, 3>(), wgrad_lds_bytes<2, 1, 1, 3>(), wgrad_lds_bytes<1, 4, 1, 3>(),
                                    wgrad_lds_bytes<1, 2, 1, 3>(), wgrad_lds_bytes<1, 1, 1, 3>()});
  if (int rc = ensure_dynamic_lds((const void*)kern, lds_all)) return rc;
  const bool prof = prof_on();
  if (prof) {
    sr_launch_record r = {};
    r.kernel_id = 48;
    r.cin = nf + 4 * gc;
    r.cout = nf;
    r.n = n;
    r.h = h;
    r.w = w;
    const double px = (double)n * h * w;
    double macs = (double)nf * (nf + 4 * gc);
    for (int k = 1; k <= 4; ++k) macs += (double)gc * (nf + (k - 1) * gc);
    r.flops = 2.0 * 9 * macs * px;
    r.bytes = 4.0 * px * 2 * (nf + 4 * gc);  // cat and D read once
    prof_begin(stream, r);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)c.nbx, (unsigned)c.total_groups), dim3(256), c.lds, stream, c.m);
  if (prof) prof_end(stream);
  SR_CHECK_LAUNCH("wgrad_f32_rdb launch");
  ReduceTable1 t1 = {};
  ReduceTable2 t2 = {};
  int max_cols = 0, max_sch = 0, max_e = 0;
  for (int i = 0; i < c.m.nsub; ++i) {
    const WgradReduce& q = c.red[i];
    TableRow1& a = t1.row[i];
    a.slab = (const float4*)q.slab;
    a.part = (float4*)q.part;
    a.bslab = q.bslab;
    a.bpart = q.bpart;
    a.e4 = q.P * q.ntap * 256;
    a.stride4 = a.e4;
    a.nb = q.CT * 32;
    a.bstride = q.CT * 32;
    a.splits = (int)q.splits;
    a.chunk = c.chunk[i];
    a.sch = c.sch[i];
    a.gi = q.gi;
    a.z0 = c.m.y0[i];
    t2.row[i] = reduce2_params(q, c.sch[i], q.gi);
    t2.z0[i] = c.m.y0[i];
    max_cols = std::max(max_cols, (a.e4 + 255) / 256 + 1);
    max_sch = std::max(max_sch, a.sch);
    max_e = std::max(max_e, q.P * q.ntap * 1024);
  }
  t1.nrows = t2.nrows = c.m.nsub;
  t2.z0[c.m.nsub] = c.total_groups;
  hipLaunchKernelGGL(wgrad_reduce1_table_kernel, dim3(max_cols, max_sch, c.total_groups), dim3(256), 0, stream, t1);
  SR_CHECK_LAUNCH("wgrad_reduce1_table launch");
  hipLaunchKernelGGL(wgrad_reduce2_table_kernel, dim3((max_e + 255) / 256, c.total_groups), dim3(256), 0, stream, t2);
  SR_CHECK_LAUNCH("wgrad_reduce2_table launch");
  return SR_OK;
}
}  // namespace sr

extern "C" void sr_dev_set_rdb_wgrad_f32(int on, int target_wgs) {
  g_rdb_wgrad_f32 = on;
  if (target_wgs > 0) g_rdb_wgrad_f32_target = target_wgs;
}

extern "C" int sr_conv3x3_wgrad_f32(const sr_conv3x3_wgrad_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  WgradParams p;
  int rc = fill_wgrad(d, &p, "sr_conv3x3_wgrad_f32");
  if (rc) return rc;
  const int cin_pad = sr_conv3x3_cin_pad(d->cin, d->first_seg, d->seg);
  SR_CHECK_ARG(cin_pad > 0 && cin_pad == d->cin_pad, "sr_conv3x3_wgrad_f32: cin_pad=%d does not match cin=%d/%d/%d",
               d->cin_pad, d->cin, d->first_seg, d->seg);
  p.H = p.vH = d->upsample ? 2 * d->in_h : d->in_h;
  p.W = p.vW = d->upsample ? 2 * d->in_w : d->in_w;
  p.src_shift = d->upsample ? 1 : 0;
  p.tap_oy = p.tap_ox = -1;
  p.cin_blocks = cin_pad / 8;
  p.strips = sr::cdiv(p.W, 32);
  const TapMap tm = {3, 1, 0, 0};
  return run_groups<3>(d, p, cin_pad, tm, true, stream);
}

// Weight gradient of the 4x4/s2 conv as four parity passes (2x2 taps over the parity sub-image of X), each
// scattering its 2x2 taps into the [cout][cin][4][4] gradient:  dy = 2*ty + 1 - ry.
extern "C" int sr_conv4x4s2_wgrad_f32(const sr_conv3x3_wgrad_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  WgradParams p;
  int rc = fill_wgrad(d, &p, "sr_conv4x4s2_wgrad_f32");
  if (rc) return rc;
  SR_CHECK_ARG(!d->upsample && d->seg == 0 && d->first_seg == d->cin && d->in_h >= 2 && d->in_w >= 2,
               "sr_conv4x4s2_wgrad_f32: unsupported option");
  const int cin_pad = (d->cin + 7) / 8 * 8;
  SR_CHECK_ARG(cin_pad == d->cin_pad, "sr_conv4x4s2_wgrad_f32: cin_pad mismatch");
  p.H = (d->in_h - 2) / 2 + 1;
  p.W = (d->in_w - 2) / 2 + 1;
  p.cin_blocks = cin_pad / 8;
  p.strips = sr::cdiv(p.W, 32);
  p.src_mul = 2;
  for (int pass = 0; pass < 4; ++pass) {
    const int ry = pass >> 1, rx = pass & 1;
    WgradParams q = p;
    q.src_oy = ry;
    q.src_ox = rx;
    q.vH = (d->in_h - ry + 1) / 2;
    q.vW = (d->in_w - rx + 1) / 2;
    q.tap_oy = ry ? -1 : 0;
    q.tap_ox = rx ? -1 : 0;
    const TapMap tm = {4, 2, 1 - ry, 1 - rx};
    rc = run_groups<2>(d, q, cin_pad, tm, pass == 0, stream);
    if (rc) return rc;
  }
  return SR_OK;
}

extern "C" void sr_dev_set_wgrad_f32_target(int wgs) { g_wgrad_target_wgs = wgs > 0 ? wgs : 512; }
