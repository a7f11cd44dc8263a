// Whole-discriminator host drivers: VGGStyleDiscriminator128 / 256 forward (discriminator_arch.py:52-72, 120-143) and its
// autograd backward as fixed sequences of launches on the caller's stream — no allocation, no synchronisation,
// graph-capturable — shaped like the generator's drivers (rrdbnet.hip).  Every launch is one of the single-op entry points
// of include/sr_hip.h with the descriptor the per-layer host path (hip_autograd.py / hip_autograd_bf16.py) builds, in the same
// order, so results are bit-identical to that path; what changes is who issues them (one C call instead of ~60 Python
// autograd applies) and what can be shared:
//
//   * the activations a forward keeps live in ONE caller-owned `saved` block, so the ESRGAN step
//     (esrgan_model.py:38-39,65-72: net_d(gt) twice, net_d(output) three times on unchanged weights) can run each DISTINCT
//     forward once and hand the same block to every backward that needs it;
//   * train-mode BatchNorm's output does not depend on its running statistics, but every forward of the reference moves
//     them.  The BatchNorm launches therefore write their update into zeroed delta vectors inside `saved`
//     (delta = momentum * batch statistic, exactly), and sr_vgg_apply_stats_* folds a forward's deltas into the running
//     buffers — running = (1 - momentum) * running + delta, the very expression of the fused update — once per forward the
//     reference would have run, in its order, and bumps num_batches_tracked.
//
// Layers (L = 0 .. 2 + 2 * stages - 1): conv0_0 (3x3, bias, LeakyReLU), conv0_1 (4x4 / s2) + BN + LeakyReLU, then per stage
// conv{i}_0 (3x3) + BN + LeakyReLU and conv{i}_1 (4x4 / s2) + BN + LeakyReLU, flatten, linear1 + LeakyReLU, linear2.
// Parameter order = state_dict order: conv0_0.{weight,bias}, conv0_1.weight, bn0_1.{weight,bias}, [conv{i}_0.weight,
// bn{i}_0.{weight,bias}, conv{i}_1.weight, bn{i}_1.{weight,bias}] ..., linear1.{weight,bias}, linear2.{weight,bias}.
#include <vector>

#include "sr_internal.h"

namespace {

constexpr int kMaxLayers = 12;  // 10 convs of the 128 network, 12 of the 256 one
constexpr float kSlope = 0.2f, kMomentum = 0.1f, kEps = 1e-5f;
constexpr int kHidden = 100;  // linear1's width (discriminator_arch.py:45)
int g_vgg_lane = 1;           // development switch (sr_dev_set_vgg_lane): 0 = the backward's weight gradients on the caller's stream

struct Layer {
  int k;             // 3 or 4
  int cin, cout;     // reference channel counts
  int in_s, out_s;   // spatial size of source and result
  bool bn, bias;
  int p_w, p_b, p_gamma, p_beta;  // indices into the parameter list (-1: none)
  int bn_index;                   // index among the BatchNorm layers (-1: none)
  // packed blob (bytes)
  size_t w_off, b_off, dg_off, w3_off;  // forward image, packed bias, data-gradient image, fp32 3x3 embedding of a 4x4 weight (bf16)
  // saved block (bytes)
  size_t u_off, z_off, a_off, mean_off, invstd_off, dm_off, dv_off;  // u: pixel-unshuffled source of a 4x4 conv (bf16)
};

struct Plan {
  int nl, nbn, nparams, cin0, nf, S, feat_ch, feat_s, nin1;
  bool bf16;
  Layer L[kMaxLayers];
  int p_l1w, p_l1b, p_l2w, p_l2b;
  size_t packed_bytes;
  // saved block
  size_t xin_off, feat_off, y1_off, logits_off, delta_off, delta_bytes, saved_bytes;
};

int r8(int v) { return (v + 7) / 8 * 8; }
int r16(int v) { return (v + 15) / 16 * 16; }

size_t act_bytes(const Plan& P, int n, int c, int s) {
  return P.bf16 ? (size_t)n * r16(c) * s * s * 2 : (size_t)n * r8(c) * s * s * 4;
}

bool make_plan(const sr_vgg_cfg* c, int n, bool bf16, Plan* P) {
  if (!c || c->num_in_ch <= 0 || c->num_feat <= 0 || (c->input_size != 128 && c->input_size != 256)) return false;
  if (bf16 && c->num_feat % 16 != 0) return false;
  const int nf = c->num_feat, stages = c->input_size == 128 ? 4 : 5;
  P->bf16 = bf16;
  P->cin0 = c->num_in_ch;
  P->nf = nf;
  P->S = c->input_size;
  int np = 0, nbn = 0, nl = 0;
  auto add = [&](int k, int cin, int cout, int in_s, bool bn, bool bias) {
    Layer& l = P->L[nl++];
    l.k = k;
    l.cin = cin;
    l.cout = cout;
    l.in_s = in_s;
    l.out_s = k == 4 ? in_s / 2 : in_s;
    l.bn = bn;
    l.bias = bias;
    l.p_w = np++;
    l.p_b = bias ? np++ : -1;
    l.p_gamma = bn ? np++ : -1;
    l.p_beta = bn ? np++ : -1;
    l.bn_index = bn ? nbn++ : -1;
  };
  add(3, c->num_in_ch, nf, P->S, false, true);  // conv0_0
  add(4, nf, nf, P->S, true, false);            // conv0_1 + bn0_1
  int s = P->S / 2, ci = nf;
  for (int i = 1; i <= stages; ++i) {
    const int co = i <= 3 ? nf << i : nf * 8;  // nf * {2, 4, 8, 8, (8)}
    add(3, ci, co, s, true, false);
    add(4, co, co, s, true, false);
    s /= 2;
    ci = co;
  }
  P->nl = nl;
  P->nbn = nbn;
  P->feat_ch = ci;
  P->feat_s = s;  // 4
  P->nin1 = ci * s * s;
  P->p_l1w = np++;
  P->p_l1b = np++;
  P->p_l2w = np++;
  P->p_l2b = np++;
  P->nparams = np;
  // packed blob
  size_t off = 0;
  for (int i = 0; i < nl; ++i) {
    Layer& l = P->L[i];
    l.w3_off = l.b_off = 0;
    if (!bf16) {
      const int cin_pad = r8(l.cin);
      l.w_off = off;
      off += sr::align_up((l.k == 3 ? sr_conv3x3_packed_weight_floats(l.cout, cin_pad) : sr_conv4x4s2_packed_weight_floats(l.cout, l.cin, 0)) * 4, 256);
      if (l.bias) {
        l.b_off = off;
        off += sr::align_up(sr_conv3x3_packed_bias_floats(l.cout) * 4, 256);
      }
      l.dg_off = off;
      off += sr::align_up((l.k == 3 ? sr_conv3x3_packed_weight_floats(cin_pad, r8(l.cout)) : sr_conv4x4s2_packed_weight_floats(l.cout, l.cin, 1)) * 4, 256);
    } else {
      const int cin3 = l.k == 4 ? 4 * l.cin : l.cin;
      if (l.k == 4) {
        l.w3_off = off;
        off += sr::align_up((size_t)l.cout * cin3 * 9 * 4, 256);
      }
      l.w_off = off;
      off += sr::align_up(sr_conv3x3_packed_weight_elems_bf16(l.cout, cin3, cin3, 0, 0) * 2, 256);
      if (l.bias) {
        l.b_off = off;
        off += sr::align_up(sr_conv3x3_packed_bias_floats(l.cout) * 4, 256);
      }
      l.dg_off = off;
      off += sr::align_up(sr_conv3x3_packed_weight_elems_bf16(l.cout, cin3, cin3, 0, 1) * 2, 256);
    }
  }
  P->packed_bytes = off;
  // saved block
  off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += sr::align_up(bytes, 256);
    return at;
  };
  if (n > 0) {
    P->xin_off = take(act_bytes(*P, n, P->cin0, P->S));
    size_t stat_floats = 0;
    for (int i = 0; i < nl; ++i) {
      Layer& l = P->L[i];
      l.u_off = (bf16 && l.k == 4) ? take(act_bytes(*P, n, 4 * l.cin, l.out_s)) : 0;
      l.z_off = l.bn ? take(act_bytes(*P, n, l.cout, l.out_s)) : 0;
      l.a_off = take(act_bytes(*P, n, l.cout, l.out_s));
      if (l.bn) {
        l.mean_off = take((size_t)l.cout * 4);
        l.invstd_off = take((size_t)l.cout * 4);
        stat_floats += 2 * sr::align_up((size_t)l.cout, 64);
      }
    }
    P->feat_off = take((size_t)n * P->nin1 * 4);
    P->y1_off = take((size_t)n * kHidden * 4);
    P->logits_off = take((size_t)n * 4);
    P->delta_bytes = stat_floats * 4;
    P->delta_off = take(P->delta_bytes);
    size_t d = P->delta_off;
    for (int i = 0; i < nl; ++i) {
      Layer& l = P->L[i];
      if (!l.bn) continue;
      l.dm_off = d;
      d += sr::align_up((size_t)l.cout, 64) * 4;
      l.dv_off = d;
      d += sr::align_up((size_t)l.cout, 64) * 4;
    }
  }
  P->saved_bytes = off;
  return true;
}

// ---- workspaces -----------------------------------------------------------------------------------------------------
struct Space {
  char *A, *Cb;       // gradient wrt the current activation (+ a second buffer for the bf16 way back through the unshuffle)
  char* Z[kMaxLayers];  // gradient wrt every conv's result: one buffer per layer, so that the weight gradients, which read them on
                        // the second lane (sr_internal.h WgradLane), never hold the data-gradient chain up
  char* red;          // reduction scratch of the BatchNorm launches
  size_t red_bytes;
  char* slab;         // weight-gradient slab
  size_t slab_bytes;
  float *dz2, *dy1, *dz1, *dfeat;  // linear head
  float* small;                    // gradients that reach their destination through the final add (see backward)
  size_t small_floats, bytes;
};

Space carve(const Plan& P, int n, char* base) {
  Space W;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += sr::align_up(bytes, 256);
    return p;
  };
  size_t big = 0;
  int maxc = 8;
  size_t slab = 0, small = 0;
  for (int i = 0; i < P.nl; ++i) {
    const Layer& l = P.L[i];
    big = std::max(big, act_bytes(P, n, l.cout, l.out_s));
    big = std::max(big, act_bytes(P, n, l.cin, l.in_s));
    maxc = std::max(maxc, l.cout);
    slab = std::max(slab, P.bf16 ? sr_conv3x3_wgrad_slab_bytes_bf16(n, l.out_s, l.out_s) : sr_conv3x3_wgrad_slab_bytes(n, l.out_s, l.out_s));
    if (l.bn) small += 2 * sr::align_up((size_t)l.cout, 64);
    if (P.bf16 && l.k == 4) small += sr::align_up((size_t)l.cout * l.cin * 16, 64) + sr::align_up((size_t)l.cout * l.cin * 36, 64);
  }
  small += sr::align_up((size_t)kHidden * P.nin1, 64) + 2 * sr::align_up((size_t)kHidden, 64) + 64;  // linear1.{weight,bias}, linear2.{weight,bias}
  W.A = take(big);
  for (int i = 0; i < P.nl; ++i) W.Z[i] = take(act_bytes(P, n, P.L[i].cout, P.L[i].out_s));
  W.Cb = P.bf16 ? take(big) : nullptr;
  W.red_bytes = sr_reduce_workspace_bytes(maxc);
  W.red = take(W.red_bytes);
  W.slab_bytes = slab;
  W.slab = take(slab);
  W.dz2 = (float*)take((size_t)n * 4);
  W.dy1 = (float*)take((size_t)n * kHidden * 4);
  W.dz1 = (float*)take((size_t)n * kHidden * 4);
  W.dfeat = (float*)take((size_t)n * P.nin1 * 4);
  W.small_floats = small;
  W.small = (float*)take(small * 4);
  W.bytes = off;
  return W;
}

// ---- small kernels --------------------------------------------------------------------------------------------------
struct StatRow {
  float* rm;
  float* rv;
  long long* nbt;
  const float* dm;
  const float* dv;
  int c;
};
struct StatTable {
  StatRow row[kMaxLayers];
  float momentum;
  int bump;  // added to num_batches_tracked
  const int32_t* abort_word;  // sr_abort_latch(): statistics of a forward behind a timed-out launch are not folded in
};
// running = (1 - momentum) * running + delta with delta = momentum * batch statistic: the fused update of the BatchNorm launches
// (train_ops.hip bn_finalize_kernel, bn_small.h) split at its `+`.
__global__ void vgg_apply_stats_kernel(const StatTable t) {
  if (t.abort_word && *t.abort_word != 0) return;
  const StatRow r = t.row[blockIdx.y];
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch < r.c) {
    r.rm[ch] = (1.f - t.momentum) * r.rm[ch] + r.dm[ch];
    r.rv[ch] = (1.f - t.momentum) * r.rv[ch] + r.dv[ch];
  }
  if (ch == 0 && r.nbt) *r.nbt += t.bump;
}

constexpr int kMaxAdd = 40;
struct AddRow {
  float* dst;
  const float* src;
  long long n;
};
struct AddTable {
  AddRow row[kMaxAdd];
  int accumulate;
};
__global__ void vgg_grad_add_kernel(const AddTable t) {
  const AddRow r = t.row[blockIdx.y];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < r.n; i += (long long)gridDim.x * blockDim.x)
    r.dst[i] = t.accumulate ? r.dst[i] + r.src[i] : r.src[i];
}

#define SR_TRY(call)        \
  do {                      \
    const int rc__ = (call); \
    if (rc__) return rc__;  \
  } while (0)

int apply_stats(const Plan& P, const char* saved, void* const* host_buffers, int bump, hipStream_t stream) {
  StatTable t = {};
  int maxc = 0;
  for (int i = 0; i < P.nl; ++i) {
    const Layer& l = P.L[i];
    if (!l.bn) continue;
    StatRow& r = t.row[l.bn_index];
    r.rm = (float*)host_buffers[3 * l.bn_index];
    r.rv = (float*)host_buffers[3 * l.bn_index + 1];
    r.nbt = (long long*)host_buffers[3 * l.bn_index + 2];
    r.dm = (const float*)(saved + l.dm_off);
    r.dv = (const float*)(saved + l.dv_off);
    r.c = l.cout;
    if (!r.rm || !r.rv) {
      sr::set_error("sr_vgg_apply_stats: null running buffer of BatchNorm %d", l.bn_index);
      return SR_EINVAL;
    }
    maxc = std::max(maxc, l.cout);
  }
  t.momentum = kMomentum;
  t.bump = bump;
  t.abort_word = sr::abort_latch();
  hipLaunchKernelGGL(vgg_apply_stats_kernel, dim3(sr::cdiv(maxc, 256), P.nbn), dim3(256), 0, stream, t);
  SR_CHECK_LAUNCH("vgg_apply_stats launch");
  return SR_OK;
}

// ---- pack -----------------------------------------------------------------------------------------------------------
int pack(const Plan& P, const float* const* hp, char* blob, hipStream_t stream) {
  for (int i = 0; i < P.nl; ++i) {
    const Layer& l = P.L[i];
    const float* w = hp[l.p_w];
    const float* b = l.bias ? hp[l.p_b] : nullptr;
    SR_CHECK_ARG(w && (!l.bias || b), "sr_vgg_pack: null parameter of conv layer %d", i);
    float* bp = l.bias ? (float*)(blob + l.b_off) : nullptr;
    if (!P.bf16) {
      if (l.k == 3) {
        SR_TRY(sr_conv3x3_pack_f32(w, b, l.cout, l.cin, l.cin, 0, 0, (float*)(blob + l.w_off), bp, stream));
        SR_TRY(sr_conv3x3_pack_f32(w, nullptr, l.cout, l.cin, l.cin, 0, 1, (float*)(blob + l.dg_off), nullptr, stream));
      } else {
        SR_TRY(sr_conv4x4s2_pack_f32(w, b, l.cout, l.cin, 0, (float*)(blob + l.w_off), bp, stream));
        SR_TRY(sr_conv4x4s2_pack_f32(w, nullptr, l.cout, l.cin, 1, (float*)(blob + l.dg_off), nullptr, stream));
      }
    } else {
      const float* w3 = w;
      int cin3 = l.cin;
      if (l.k == 4) {  // the 4x4 / s2 conv runs as a 3x3 conv of the pixel-unshuffled source (disc_bf16.hip)
        float* e = (float*)(blob + l.w3_off);
        SR_TRY(sr_conv4x4s2_weight_as_3x3_f32(const_cast<float*>(w), e, l.cout, l.cin, 0, stream));
        w3 = e;
        cin3 = 4 * l.cin;
      }
      SR_TRY(sr_conv3x3_pack_bf16(w3, b, l.cout, cin3, cin3, 0, 0, blob + l.w_off, bp, stream));
      SR_TRY(sr_conv3x3_pack_bf16(w3, nullptr, l.cout, cin3, cin3, 0, 1, blob + l.dg_off, nullptr, stream));
    }
  }
  return SR_OK;
}

// ---- forward --------------------------------------------------------------------------------------------------------
int forward(const Plan& P, const char* blob, const float* const* hp, void* const* hb, const float* x, float* logits, int n, int train,
            char* saved, char* ws, size_t ws_bytes, hipStream_t stream) {
  const bool bf = P.bf16;
  const int esz = bf ? 2 : 4;
  auto blocks = [&](int c) { return bf ? r16(c) / 16 : r8(c) / 8; };
  auto img = [&](int c, int s) { return (int64_t)(bf ? r16(c) : r8(c)) * s * s; };  // elements per image
  const size_t red_bytes = sr_reduce_workspace_bytes(P.feat_ch > 8 ? P.feat_ch : 8);
  SR_CHECK_ARG(ws && ws_bytes >= red_bytes, "sr_vgg_forward: workspace too small");
  if (train) {
    if (hipMemsetAsync(saved + P.delta_off, 0, P.delta_bytes, stream) != hipSuccess) {
      sr::set_error("sr_vgg_forward: hipMemsetAsync failed");
      return SR_ELAUNCH;
    }
  }
  char* xin = saved + P.xin_off;
  if (!bf)
    SR_TRY(sr_nchw_to_cb8_f32(x, (float*)xin, n, P.cin0, P.S, P.S, 1, blocks(P.cin0), img(P.cin0, P.S), stream));
  else
    SR_TRY(sr_nchw_to_cb16_bf16(x, xin, n, P.cin0, P.S, P.S, 1, blocks(P.cin0), img(P.cin0, P.S), stream));
  const char* src = xin;
  for (int i = 0; i < P.nl; ++i) {
    const Layer& l = P.L[i];
    char* out = saved + (l.bn ? l.z_off : l.a_off);
    sr_conv3x3_desc d = {};
    d.wpacked = (const float*)(blob + l.w_off);
    d.bpacked = l.bias ? (const float*)(blob + l.b_off) : nullptr;
    d.cout = l.cout;
    d.out = (float*)out;
    d.out_img_stride = img(l.cout, l.out_s);
    d.n = n;
    d.act_slope = l.bn ? 1.0f : kSlope;
    d.alpha = 1.0f;
    if (!bf) {
      d.in = (const float*)src;
      d.in_img_stride = img(l.cin, l.in_s);
      d.cin_pad = r8(l.cin);
      d.in_h = d.in_w = l.in_s;
      SR_TRY(l.k == 3 ? sr_conv3x3_f32(&d, stream) : sr_conv4x4s2_f32(&d, stream));
    } else {
      if (l.k == 4) {
        char* u = saved + l.u_off;
        SR_TRY(sr_cb16_unshuffle2_bf16(src, img(l.cin, l.in_s), u, img(4 * l.cin, l.out_s), n, blocks(l.cin), l.out_s, l.out_s, 0, stream));
        d.in = (const float*)u;
        d.in_img_stride = img(4 * l.cin, l.out_s);
        d.cin_pad = r16(4 * l.cin);
        d.in_h = d.in_w = l.out_s;
        d.s2_channels = l.cin % 64 == 0 ? l.cin : 0;
      } else {
        d.in = (const float*)src;
        d.in_img_stride = img(l.cin, l.in_s);
        d.cin_pad = r16(l.cin);
        d.in_h = d.in_w = l.in_s;
      }
      SR_TRY(sr_conv3x3_bf16(&d, stream));
    }
    if (l.bn) {
      char* a = saved + l.a_off;
      const float* gamma = hp[l.p_gamma];
      const float* beta = hp[l.p_beta];
      SR_CHECK_ARG(gamma && beta, "sr_vgg_forward: null BatchNorm parameter of layer %d", i);
      float* rm;
      float* rv;
      if (train) {  // the update goes to the zeroed delta vectors: momentum * batch statistic
        rm = (float*)(saved + l.dm_off);
        rv = (float*)(saved + l.dv_off);
      } else {
        rm = (float*)hb[3 * l.bn_index];
        rv = (float*)hb[3 * l.bn_index + 1];
      }
      const int64_t ns = img(l.cout, l.out_s);
      if (!bf)
        SR_TRY(sr_bn_lrelu_fwd_f32((const float*)out, ns, (float*)a, ns, n, l.cout, l.out_s, l.out_s, gamma, beta, rm, rv, train, kMomentum,
                                   kEps, kSlope, (float*)(saved + l.mean_off), (float*)(saved + l.invstd_off), ws, red_bytes, stream));
      else
        SR_TRY(sr_bn_lrelu_fwd_bf16(out, ns, a, ns, n, l.cout, l.out_s, l.out_s, gamma, beta, rm, rv, train, kMomentum, kEps, kSlope,
                                    (float*)(saved + l.mean_off), (float*)(saved + l.invstd_off), ws, red_bytes, stream));
      src = a;
    } else {
      src = out;
    }
    (void)esz;
  }
  float* feat = (float*)(saved + P.feat_off);
  if (!bf)
    SR_TRY(sr_cb8_to_nchw_f32((const float*)src, img(P.feat_ch, P.feat_s), feat, n, P.feat_ch, P.feat_s, P.feat_s, 1, stream));
  else
    SR_TRY(sr_cb16_to_nchw_f32(src, img(P.feat_ch, P.feat_s), feat, n, P.feat_ch, P.feat_s, P.feat_s, 1, stream));
  float* y1 = (float*)(saved + P.y1_off);
  float* lg = (float*)(saved + P.logits_off);
  SR_CHECK_ARG(hp[P.p_l1w] && hp[P.p_l1b] && hp[P.p_l2w] && hp[P.p_l2b], "sr_vgg_forward: null linear parameter");
  SR_TRY(sr_linear_fwd_f32(feat, hp[P.p_l1w], hp[P.p_l1b], y1, n, P.nin1, kHidden, kSlope, stream));
  SR_TRY(sr_linear_fwd_f32(y1, hp[P.p_l2w], hp[P.p_l2b], lg, n, kHidden, 1, 1.0f, stream));
  if (logits && hipMemcpyAsync(logits, lg, (size_t)n * 4, hipMemcpyDeviceToDevice, stream) != hipSuccess) {
    sr::set_error("sr_vgg_forward: hipMemcpyAsync failed");
    return SR_ELAUNCH;
  }
  if (train && hb) SR_TRY(apply_stats(P, saved, hb, 1, stream));
  return SR_OK;
}

// ---- backward -------------------------------------------------------------------------------------------------------
int backward(const Plan& P, const char* blob, const float* const* hp, const char* saved, const float* dlogits, int n, int train,
             float* const* dp, int accumulate, float* dx, char* wsbase, hipStream_t stream) {
  const bool bf = P.bf16;
  auto blocks = [&](int c) { return bf ? r16(c) / 16 : r8(c) / 8; };
  auto img = [&](int c, int s) { return (int64_t)(bf ? r16(c) : r8(c)) * s * s; };
  Space W = carve(P, n, wsbase);
  const bool need_p = dp && dp[P.L[1].p_w];  // parameters travel together: all or none (a frozen discriminator gives none)
  if (dp)
    for (int i = 0; i < P.nparams; ++i)
      SR_CHECK_ARG((dp[i] != nullptr) == need_p, "sr_vgg_backward: parameter gradients must be requested for all parameters or none");
  AddTable adds = {};
  adds.accumulate = accumulate;
  int nadd = 0;
  float* small = W.small;
  auto via_add = [&](int pidx, size_t count) {  // scratch for a gradient whose kernel cannot accumulate; added at the end
    float* s = small;
    small += sr::align_up(count, 64);
    AddRow& r = adds.row[nadd++];
    r.dst = dp[pidx];
    r.src = s;
    r.n = (long long)count;
    return s;
  };
  // linear head (discriminator_arch.py:69-71)
  const float* feat = (const float*)(saved + P.feat_off);
  const float* y1 = (const float*)(saved + P.y1_off);
  const float* lg = (const float*)(saved + P.logits_off);
  {
    float* dw2 = need_p ? via_add(P.p_l2w, kHidden) : nullptr;
    float* db2 = need_p ? via_add(P.p_l2b, 1) : nullptr;
    SR_TRY(sr_linear_bwd_f32(y1, hp[P.p_l2w], lg, dlogits, n, kHidden, 1, 1.0f, W.dz2, W.dy1, dw2, db2, stream));
    float* dw1 = need_p ? via_add(P.p_l1w, (size_t)kHidden * P.nin1) : nullptr;
    float* db1 = need_p ? via_add(P.p_l1b, kHidden) : nullptr;
    SR_TRY(sr_linear_bwd_f32(feat, hp[P.p_l1w], y1, W.dy1, n, P.nin1, kHidden, kSlope, W.dz1, W.dfeat, dw1, db1, stream));
  }
  char* gA = W.A;  // gradient wrt the activation that leaves layer i
  // Weight gradients depend on the data gradients issued so far and feed only the optimiser: they go to the lane (a side stream for
  // launches of this size; sr_dev_set_backward_overlap 0 = the caller's stream) and are joined at the end of the call.
  sr::WgradLane lane;
  {
    const int mode = sr::backward_overlap();
    lane.begin(stream, g_vgg_lane && need_p && (mode > 0 || (mode < 0 && (long long)n * P.S * P.S <= 64ll * 128 * 128)));
  }
  long long ticket = 0;
  if (!bf)
    SR_TRY(sr_nchw_to_cb8_f32(W.dfeat, (float*)gA, n, P.feat_ch, P.feat_s, P.feat_s, 1, blocks(P.feat_ch), img(P.feat_ch, P.feat_s), stream));
  else
    SR_TRY(sr_nchw_to_cb16_bf16(W.dfeat, gA, n, P.feat_ch, P.feat_s, P.feat_s, 1, blocks(P.feat_ch), img(P.feat_ch, P.feat_s), stream));
  for (int i = P.nl - 1; i >= 0; --i) {
    const Layer& l = P.L[i];
    const int64_t ns = img(l.cout, l.out_s);
    const char* a = saved + l.a_off;
    char* gZ = W.Z[i];  // gradient wrt the conv result of layer i
    if (l.bn) {
      float* dgamma = need_p ? via_add(l.p_gamma, l.cout) : (float*)W.small;  // (a frozen network: written, never read)
      float* dbeta = need_p ? via_add(l.p_beta, l.cout) : (float*)W.small + sr::align_up((size_t)l.cout, 64);
      const char* z = saved + l.z_off;
      if (!bf)
        SR_TRY(sr_bn_lrelu_bwd_f32((const float*)z, ns, (const float*)gA, ns, (const float*)a, ns, (float*)gZ, ns, n, l.cout, l.out_s, l.out_s,
                                   hp[l.p_gamma], (const float*)(saved + l.mean_off), (const float*)(saved + l.invstd_off), train, kSlope,
                                   dgamma, dbeta, W.red, W.red_bytes, stream));
      else
        SR_TRY(sr_bn_lrelu_bwd_bf16(z, ns, gA, ns, a, ns, gZ, ns, n, l.cout, l.out_s, l.out_s, hp[l.p_gamma],
                                    (const float*)(saved + l.mean_off), (const float*)(saved + l.invstd_off), train, kSlope, dgamma, dbeta,
                                    W.red, W.red_bytes, stream));
    } else {  // conv0_0: LeakyReLU backward against the saved output
      if (!bf)
        SR_TRY(sr_lrelu_bwd_f32((const float*)gA, (const float*)a, (float*)gZ, kSlope, (int64_t)n * ns, stream));
      else
        SR_TRY(sr_lrelu_bwd_bf16(gA, a, gZ, kSlope, (int64_t)n * ns, stream));
    }
    const char* src = i == 0 ? saved + P.xin_off : saved + P.L[i - 1].a_off;  // the conv's forward source (plain layout)
    const bool need_x = i > 0 || dx != nullptr;
    // data gradient: gZ -> gA (gradient wrt the previous activation)
    if (need_x) {
      sr_conv3x3_desc d = {};
      d.in = (const float*)gZ;
      d.in_img_stride = ns;
      d.in_h = d.in_w = l.out_s;
      d.wpacked = (const float*)(blob + l.dg_off);
      d.n = n;
      d.act_slope = 1.0f;
      d.alpha = 1.0f;
      if (!bf) {
        d.cin_pad = r8(l.cout);
        d.cout = r8(l.cin);
        d.out = (float*)gA;
        d.out_img_stride = img(l.cin, l.in_s);
        if (l.k == 3) {
          SR_TRY(sr_conv3x3_f32(&d, stream));
        } else {
          d.out_h = d.out_w = l.in_s;
          SR_TRY(sr_conv4x4s2_dgrad_f32(&d, stream));
        }
      } else {
        d.cin_pad = r16(l.cout);
        d.s2_side = 1;
        if (l.k == 3) {
          d.cout = r16(l.cin);
          d.out = (float*)gA;
          d.out_img_stride = img(l.cin, l.in_s);
          SR_TRY(sr_conv3x3_bf16(&d, stream));
        } else {  // gradient of the unshuffled source, then the way back through the unshuffle
          d.cout = r16(4 * l.cin);
          d.out = (float*)W.Cb;
          d.out_img_stride = img(4 * l.cin, l.out_s);
          d.s2_channels = l.cin % 64 == 0 ? l.cin : 0;
          SR_TRY(sr_conv3x3_bf16(&d, stream));
          SR_TRY(sr_cb16_unshuffle2_bf16(W.Cb, img(4 * l.cin, l.out_s), gA, img(l.cin, l.in_s), n, blocks(l.cin), l.out_s, l.out_s, 1, stream));
        }
      }
    }
    // weight gradient
    if (need_p) {
      hipStream_t ws = lane.hand();
      struct Done {
        sr::WgradLane& l;
        long long k;
        ~Done() { l.done(k); }
      } mark{lane, ticket++};
      sr_conv3x3_wgrad_desc g = {};
      g.dy = (const float*)gZ;
      g.dy_img_stride = ns;
      g.cout = l.cout;
      g.n = n;
      g.scale = 1.0f;
      g.slab = W.slab;
      g.slab_bytes = bf ? sr_conv3x3_wgrad_slab_bytes_bf16(n, l.out_s, l.out_s) : sr_conv3x3_wgrad_slab_bytes(n, l.out_s, l.out_s);
      g.dbias = l.bias ? dp[l.p_b] : nullptr;
      g.accumulate = accumulate;
      if (!bf) {
        g.x = (const float*)src;
        g.x_img_stride = img(l.cin, l.in_s);
        g.cin_pad = r8(l.cin);
        g.in_h = g.in_w = l.in_s;
        g.cin = g.first_seg = l.cin;
        g.dweight = dp[l.p_w];
        SR_TRY(l.k == 3 ? sr_conv3x3_wgrad_f32(&g, ws) : sr_conv4x4s2_wgrad_f32(&g, ws));
      } else if (l.k == 3) {
        g.x = (const float*)src;
        g.x_img_stride = img(l.cin, l.in_s);
        g.cin_pad = r16(l.cin);
        g.in_h = g.in_w = l.in_s;
        g.cin = g.first_seg = l.cin;
        g.dweight = dp[l.p_w];
        SR_TRY(sr_conv3x3_wgrad_bf16(&g, ws));
      } else {  // 3x3 gradient of the embedded weight, folded back to 4x4, added at the end
        float* dw3 = small;
        small += sr::align_up((size_t)l.cout * l.cin * 36, 64);
        float* dw4 = via_add(l.p_w, (size_t)l.cout * l.cin * 16);
        g.x = (const float*)(saved + l.u_off);
        g.x_img_stride = img(4 * l.cin, l.out_s);
        g.cin_pad = r16(4 * l.cin);
        g.in_h = g.in_w = l.out_s;
        g.cin = g.first_seg = 4 * l.cin;
        g.dweight = dw3;
        g.accumulate = 0;
        SR_TRY(sr_conv3x3_wgrad_bf16(&g, ws));
        SR_TRY(sr_conv4x4s2_weight_as_3x3_f32(dw4, dw3, l.cout, l.cin, 1, ws));
      }
    }
  }
  if (dx) {
    if (!bf)
      SR_TRY(sr_cb8_to_nchw_f32((const float*)gA, img(P.cin0, P.S), dx, n, P.cin0, P.S, P.S, 1, stream));
    else
      SR_TRY(sr_cb16_to_nchw_f32(gA, img(P.cin0, P.S), dx, n, P.cin0, P.S, P.S, 1, stream));
  }
  lane.end();  // the caller's stream waits for the last weight gradient (the folded bf16 4x4 gradients are added below)
  if (nadd > 0) {
    if ((size_t)(small - W.small) > W.small_floats || nadd > kMaxAdd) {
      sr::set_error("sr_vgg_backward: internal scratch overflow (%d rows)", nadd);
      return SR_EINVAL;
    }
    long long maxn = 0;
    for (int i = 0; i < nadd; ++i) maxn = std::max(maxn, adds.row[i].n);
    const int gx = (int)std::min<long long>((maxn + 255) / 256, 256);
    hipLaunchKernelGGL(vgg_grad_add_kernel, dim3(gx, nadd), dim3(256), 0, stream, adds);
    SR_CHECK_LAUNCH("vgg_grad_add launch");
  }
  return SR_OK;
}

}  // namespace

// ---- C ABI ----------------------------------------------------------------------------------------------------------
extern "C" int sr_vgg_num_params(const sr_vgg_cfg* cfg) {
  Plan P;
  return make_plan(cfg, 0, false, &P) ? P.nparams : 0;
}
extern "C" int sr_vgg_num_batchnorm(const sr_vgg_cfg* cfg) {
  Plan P;
  return make_plan(cfg, 0, false, &P) ? P.nbn : 0;
}
static size_t packed_bytes(const sr_vgg_cfg* cfg, bool bf16) {
  Plan P;
  return make_plan(cfg, 0, bf16, &P) ? P.packed_bytes : 0;
}
static size_t saved_bytes(const sr_vgg_cfg* cfg, int n, bool bf16) {
  Plan P;
  return (n > 0 && make_plan(cfg, n, bf16, &P)) ? P.saved_bytes : 0;
}
static size_t workspace_bytes(const sr_vgg_cfg* cfg, int n, bool bf16) {
  Plan P;
  if (n <= 0 || !make_plan(cfg, n, bf16, &P)) return 0;
  return carve(P, n, nullptr).bytes;
}
extern "C" size_t sr_vgg_packed_bytes(const sr_vgg_cfg* cfg) { return packed_bytes(cfg, false); }
extern "C" size_t sr_vgg_packed_bytes_bf16(const sr_vgg_cfg* cfg) { return packed_bytes(cfg, true); }
extern "C" size_t sr_vgg_saved_bytes(const sr_vgg_cfg* cfg, int n) { return saved_bytes(cfg, n, false); }
extern "C" size_t sr_vgg_saved_bytes_bf16(const sr_vgg_cfg* cfg, int n) { return saved_bytes(cfg, n, true); }
extern "C" size_t sr_vgg_workspace_bytes(const sr_vgg_cfg* cfg, int n) { return workspace_bytes(cfg, n, false); }
extern "C" size_t sr_vgg_workspace_bytes_bf16(const sr_vgg_cfg* cfg, int n) { return workspace_bytes(cfg, n, true); }

static int pack_any(const sr_vgg_cfg* cfg, const float* const* host_params, void* packed, bool bf16, void* stream) {
  Plan P;
  SR_CHECK_ARG(make_plan(cfg, 0, bf16, &P), "sr_vgg_pack: bad configuration");
  SR_CHECK_ARG(host_params && packed, "sr_vgg_pack: null argument");
  return pack(P, host_params, (char*)packed, (hipStream_t)stream);
}
extern "C" int sr_vgg_pack_f32(const sr_vgg_cfg* cfg, const float* const* host_params, void* packed, void* stream) {
  return pack_any(cfg, host_params, packed, false, stream);
}
extern "C" int sr_vgg_pack_bf16(const sr_vgg_cfg* cfg, const float* const* host_params, void* packed, void* stream) {
  return pack_any(cfg, host_params, packed, true, stream);
}

static int forward_any(const sr_vgg_cfg* cfg, const void* packed, const float* const* host_params, void* const* host_buffers,
                       const float* x, float* logits, int n, int train, void* saved, size_t saved_bytes_, void* workspace,
                       size_t workspace_bytes_, bool bf16, void* stream) {
  Plan P;
  SR_CHECK_ARG(n > 0 && make_plan(cfg, n, bf16, &P), "sr_vgg_forward: bad configuration or batch");
  SR_CHECK_ARG(packed && host_params && x && saved && workspace, "sr_vgg_forward: null argument");
  SR_CHECK_ARG(train || host_buffers, "sr_vgg_forward: eval mode needs the running statistics");
  SR_CHECK_ARG(saved_bytes_ >= P.saved_bytes, "sr_vgg_forward: saved block %zu B < %zu B", saved_bytes_, P.saved_bytes);
  return forward(P, (const char*)packed, host_params, host_buffers, x, logits, n, train ? 1 : 0, (char*)saved, (char*)workspace,
                 workspace_bytes_, (hipStream_t)stream);
}
extern "C" int sr_vgg_forward_f32(const sr_vgg_cfg* cfg, const void* packed, const float* const* host_params, void* const* host_buffers,
                                  const float* x, float* logits, int n, int train, void* saved, size_t saved_bytes_, void* workspace,
                                  size_t workspace_bytes_, void* stream) {
  return forward_any(cfg, packed, host_params, host_buffers, x, logits, n, train, saved, saved_bytes_, workspace, workspace_bytes_, false, stream);
}
extern "C" int sr_vgg_forward_bf16(const sr_vgg_cfg* cfg, const void* packed, const float* const* host_params, void* const* host_buffers,
                                   const float* x, float* logits, int n, int train, void* saved, size_t saved_bytes_, void* workspace,
                                   size_t workspace_bytes_, void* stream) {
  return forward_any(cfg, packed, host_params, host_buffers, x, logits, n, train, saved, saved_bytes_, workspace, workspace_bytes_, true, stream);
}

static int apply_any(const sr_vgg_cfg* cfg, const void* saved, size_t saved_bytes_, int n, void* const* host_buffers, int repeats,
                     bool bf16, void* stream) {
  Plan P;
  SR_CHECK_ARG(n > 0 && make_plan(cfg, n, bf16, &P), "sr_vgg_apply_stats: bad configuration or batch");
  SR_CHECK_ARG(saved && host_buffers && saved_bytes_ >= P.saved_bytes && repeats >= 1, "sr_vgg_apply_stats: bad argument");
  for (int r = 0; r < repeats; ++r) SR_TRY(apply_stats(P, (const char*)saved, host_buffers, 1, (hipStream_t)stream));
  return SR_OK;
}
extern "C" int sr_vgg_apply_stats_f32(const sr_vgg_cfg* cfg, const void* saved, size_t saved_bytes_, int n, void* const* host_buffers,
                                      int repeats, void* stream) {
  return apply_any(cfg, saved, saved_bytes_, n, host_buffers, repeats, false, stream);
}
extern "C" int sr_vgg_apply_stats_bf16(const sr_vgg_cfg* cfg, const void* saved, size_t saved_bytes_, int n, void* const* host_buffers,
                                       int repeats, void* stream) {
  return apply_any(cfg, saved, saved_bytes_, n, host_buffers, repeats, true, stream);
}

static int backward_any(const sr_vgg_cfg* cfg, const void* packed, const float* const* host_params, const void* saved, size_t saved_bytes_,
                        const float* dlogits, int n, int train, float* const* host_dparams, int accumulate, float* dx, void* workspace,
                        size_t workspace_bytes_, bool bf16, void* stream) {
  Plan P;
  SR_CHECK_ARG(n > 0 && make_plan(cfg, n, bf16, &P), "sr_vgg_backward: bad configuration or batch");
  SR_CHECK_ARG(packed && host_params && saved && dlogits && workspace, "sr_vgg_backward: null argument");
  SR_CHECK_ARG(saved_bytes_ >= P.saved_bytes, "sr_vgg_backward: saved block too small");
  SR_CHECK_ARG(workspace_bytes_ >= carve(P, n, nullptr).bytes, "sr_vgg_backward: workspace %zu B too small", workspace_bytes_);
  return backward(P, (const char*)packed, host_params, (const char*)saved, dlogits, n, train ? 1 : 0, host_dparams, accumulate ? 1 : 0, dx,
                  (char*)workspace, (hipStream_t)stream);
}
extern "C" int sr_vgg_backward_f32(const sr_vgg_cfg* cfg, const void* packed, const float* const* host_params, const void* saved,
                                   size_t saved_bytes_, const float* dlogits, int n, int train, float* const* host_dparams, int accumulate, float* dx,
                                   void* workspace, size_t workspace_bytes_, void* stream) {
  return backward_any(cfg, packed, host_params, saved, saved_bytes_, dlogits, n, train, host_dparams, accumulate, dx, workspace, workspace_bytes_,
                      false, stream);
}
extern "C" int sr_vgg_backward_bf16(const sr_vgg_cfg* cfg, const void* packed, const float* const* host_params, const void* saved,
                                    size_t saved_bytes_, const float* dlogits, int n, int train, float* const* host_dparams, int accumulate, float* dx,
                                    void* workspace, size_t workspace_bytes_, void* stream) {
  return backward_any(cfg, packed, host_params, saved, saved_bytes_, dlogits, n, train, host_dparams, accumulate, dx, workspace, workspace_bytes_,
                      true, stream);
}

extern "C" void sr_dev_set_vgg_lane(int on) { g_vgg_lane = on; }
