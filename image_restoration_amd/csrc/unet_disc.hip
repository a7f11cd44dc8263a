// Whole-discriminator host drivers for UNetDiscriminatorSN with compute_dtype = 'bf16' (the discriminator BASELINE configs 3-4 name;
// absent from the reference: SURVEY.md section 0 D2, the oracle restates the published architecture): forward and autograd backward as
// fixed sequences of launches on the caller's stream, shaped like the generator's and the VGG discriminator's drivers — no
// allocation, no synchronisation, graph-capturable.  Every launch is one of the single-op entry points of include/sr_hip.h with the
// descriptor the per-layer host path (hip_autograd_bf16.py: ConvFn16 / ForkU2Fn16 / Bilinear2xFn16 and archs/
// unet_discriminator_arch.py::_forward_bf16) builds, in the same order: results are bit-identical to that path.
//
// Spectral normalisation stays outside: the ten "parameters" of these calls are the EFFECTIVE weights of one forward (conv0 / conv9
// weight + bias as they are, conv1-8 the normalised weights sr_spectral_norm_fwd_batch_f32 produced for this forward); the gradients
// that come back are the gradients wrt those, and the host maps them through sr_spectral_norm_bwd_f32.  One power iteration per
// forward changes the effective weights, so — unlike the BatchNorm VGG network — repeated forwards of a step are NOT identical and a
// `saved` block serves exactly the forward that filled it.
//
// Network (hw = input size, nf = num_feat; u* = tensors that only exist pixel-unshuffled, sr_conv3x3_desc.out_unshuffle2):
//   u0 = lrelu(conv0(x))                     nf  @ hw      conv3x3 + bias
//   u1 = lrelu(conv1(u0))                    2nf @ hw/2    conv4x4/s2 = 3x3 conv of the unshuffled source (disc_bf16.hip)
//   u2 = lrelu(conv2(u1))                    4nf @ hw/4
//   x3 = lrelu(conv3(u2))                    8nf @ hw/8
//   x4 = lrelu(conv4(up(x3)))                4nf @ hw/4    up = bilinear x2 (align_corners = False)
//   x5 = lrelu(conv5(up(x4 + u2)))           2nf @ hw/2    (skip connections when cfg->skip_connection)
//   x6 = lrelu(conv6(up(x5 + u1))) + u0      nf  @ hw      the last skip rides in conv6's epilogue (res1_keep_sign)
//   out = conv9(lrelu(conv8(lrelu(conv7(x6)))))            1 logit per pixel, fp32 NCHW
#include <algorithm>

#include "sr_internal.h"

namespace {

constexpr float kSlope = 0.2f;
constexpr int kConvs = 10;

struct ConvL {
  int k, cin, cout;   // reference kernel size and channel counts
  int cin3;           // channels of the 3x3 conv that runs: 4 cin for a 4x4 / s2 conv
  int s;              // divisor of the input size at which the conv's OUTPUT lives (1, 2, 4, 8)
  bool bias;
  int s2;             // sr_conv3x3_desc.s2_channels
  size_t w3_off, w_off, b_off, dg_off;  // packed blob (bytes): fp32 3x3 embedding (4x4 convs), forward image, packed bias, data-gradient image
};

struct Plan {
  int cin0, nf, skip;
  ConvL L[kConvs];
  size_t packed_bytes;
};

int r16(int v) { return (v + 15) / 16 * 16; }

bool make_plan(const sr_unet_cfg* c, Plan* P) {
  if (!c || c->num_in_ch <= 0 || c->num_feat <= 0 || c->num_feat % 16 != 0) return false;
  const int nf = c->num_feat;
  P->cin0 = c->num_in_ch;
  P->nf = nf;
  P->skip = c->skip_connection ? 1 : 0;
  const int spec[kConvs][4] = {{3, c->num_in_ch, nf, 1},      {4, nf, 2 * nf, 2},     {4, 2 * nf, 4 * nf, 4}, {4, 4 * nf, 8 * nf, 8},
                               {3, 8 * nf, 4 * nf, 4},        {3, 4 * nf, 2 * nf, 2}, {3, 2 * nf, nf, 1},     {3, nf, nf, 1},
                               {3, nf, nf, 1},                {3, nf, 1, 1}};
  size_t off = 0;
  for (int i = 0; i < kConvs; ++i) {
    ConvL& l = P->L[i];
    l.k = spec[i][0];
    l.cin = spec[i][1];
    l.cout = spec[i][2];
    l.s = spec[i][3];
    l.cin3 = l.k == 4 ? 4 * l.cin : l.cin;
    l.bias = i == 0 || i == 9;
    l.s2 = (l.k == 4 && l.cin % 64 == 0) ? l.cin : 0;
    l.w3_off = off;
    if (l.k == 4) off += sr::align_up((size_t)l.cout * l.cin3 * 9 * 4, 256);
    l.w_off = off;
    off += sr::align_up(sr_conv3x3_packed_weight_elems_bf16(l.cout, l.cin3, l.cin3, 0, 0) * 2, 256);
    l.b_off = off;
    if (l.bias) off += sr::align_up(sr_conv3x3_packed_bias_floats(l.cout) * 4, 256);
    l.dg_off = off;
    off += sr::align_up(sr_conv3x3_packed_weight_elems_bf16(l.cout, l.cin3, l.cin3, 0, 1) * 2, 256);
  }
  P->packed_bytes = off;
  return true;
}

// activations a forward keeps (bf16 CB16; sizes in elements per image = r16(channels) * h * w)
struct Saved {
  char *xin, *u0, *u1, *u2, *x3, *b3, *x4, *b4, *x5, *b5, *x6, *x7, *x8;
  size_t bytes;
};
// gradients and scratch of a backward
struct Work {
  char *dz9, *g8, *g7, *g6, *dz6, *gb5, *g5, *gp5, *gb4, *g4, *gp4, *gb3, *g3, *d2, *dz2, *d1, *dz1, *d0, *dz0, *dxin;
  char* slab;
  size_t slab_bytes;
  float* dw3;  // 3x3 gradient of an embedded 4x4 weight before it is folded back
  size_t bytes;
};

struct Carver {
  char* base;
  size_t off = 0;
  char* take(size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += sr::align_up(bytes, 256);
    return p;
  }
};

size_t act(int n, int c, int h, int w) { return (size_t)n * r16(c) * h * w * 2; }

Saved carve_saved(const Plan& P, int n, int h, int w, char* base) {
  Saved S;
  Carver cv{base};
  const int nf = P.nf;
  S.xin = cv.take(act(n, P.cin0, h, w));
  S.u0 = cv.take(act(n, nf, h, w));
  S.u1 = cv.take(act(n, 2 * nf, h / 2, w / 2));
  S.u2 = cv.take(act(n, 4 * nf, h / 4, w / 4));
  S.x3 = cv.take(act(n, 8 * nf, h / 8, w / 8));
  S.b3 = cv.take(act(n, 8 * nf, h / 4, w / 4));
  S.x4 = cv.take(act(n, 4 * nf, h / 4, w / 4));
  S.b4 = cv.take(act(n, 4 * nf, h / 2, w / 2));
  S.x5 = cv.take(act(n, 2 * nf, h / 2, w / 2));
  S.b5 = cv.take(act(n, 2 * nf, h, w));
  S.x6 = cv.take(act(n, nf, h, w));
  S.x7 = cv.take(act(n, nf, h, w));
  S.x8 = cv.take(act(n, nf, h, w));
  S.bytes = cv.off;
  return S;
}

Work carve_work(const Plan& P, int n, int h, int w, char* base) {
  Work W;
  Carver cv{base};
  const int nf = P.nf;
  W.dz9 = cv.take(act(n, 1, h, w));
  W.g8 = cv.take(act(n, nf, h, w));
  W.g7 = cv.take(act(n, nf, h, w));
  W.g6 = cv.take(act(n, nf, h, w));
  W.dz6 = cv.take(act(n, nf, h, w));
  W.gb5 = cv.take(act(n, 2 * nf, h, w));
  W.g5 = cv.take(act(n, 2 * nf, h / 2, w / 2));
  W.gp5 = cv.take(act(n, 2 * nf, h / 2, w / 2));
  W.gb4 = cv.take(act(n, 4 * nf, h / 2, w / 2));
  W.g4 = cv.take(act(n, 4 * nf, h / 4, w / 4));
  W.gp4 = cv.take(act(n, 4 * nf, h / 4, w / 4));
  W.gb3 = cv.take(act(n, 8 * nf, h / 4, w / 4));
  W.g3 = cv.take(act(n, 8 * nf, h / 8, w / 8));
  W.d2 = cv.take(act(n, 4 * nf, h / 4, w / 4));  // gradient of u2 (unshuffled: 16 nf channels at hw/8 = the same bytes)
  W.dz2 = cv.take(act(n, 4 * nf, h / 4, w / 4));
  W.d1 = cv.take(act(n, 2 * nf, h / 2, w / 2));
  W.dz1 = cv.take(act(n, 2 * nf, h / 2, w / 2));
  W.d0 = cv.take(act(n, nf, h, w));
  W.dz0 = cv.take(act(n, nf, h, w));
  W.dxin = cv.take(act(n, P.cin0, h, w));
  size_t slab = 0, dw3 = 0;
  for (int i = 0; i < kConvs; ++i) {
    const ConvL& l = P.L[i];
    slab = std::max(slab, sr_conv3x3_wgrad_slab_bytes_bf16(n, h / l.s, w / l.s));
    if (l.k == 4) dw3 = std::max(dw3, (size_t)l.cout * l.cin3 * 9 * 4);
  }
  W.slab_bytes = slab;
  W.slab = cv.take(slab);
  W.dw3 = (float*)cv.take(dw3);
  W.bytes = cv.off;
  return W;
}

#define SR_TRY(call)         \
  do {                       \
    const int rc__ = (call); \
    if (rc__) return rc__;   \
  } while (0)

// one conv of the forward: src (channels of l.cin3 at the conv's output size) -> out
struct ConvOpt {
  float act = 1.f;
  int out_u2 = 0;              // sr_conv3x3_desc.out_unshuffle2
  const char* res1_u2 = nullptr;  // conv6: the skip that only exists unshuffled, added with the sign-keeping rounding
  int64_t res1_stride = 0;
  float* nchw = nullptr;       // conv9: fp32 NCHW destination
};
int conv_fwd(const ConvL& l, const char* blob, const char* src, char* out, int n, int h, int w, const ConvOpt& o, hipStream_t stream) {
  sr_conv3x3_desc d = {};
  d.in = (const float*)src;
  d.in_img_stride = (int64_t)r16(l.cin3) * h * w;
  d.cin_pad = r16(l.cin3);
  d.in_h = h;
  d.in_w = w;
  d.wpacked = (const float*)(blob + l.w_off);
  d.bpacked = l.bias ? (const float*)(blob + l.b_off) : nullptr;
  d.cout = l.cout;
  d.n = n;
  d.act_slope = o.act;
  d.alpha = 1.f;
  d.s2_channels = l.s2;
  if (o.nchw) {
    d.out = o.nchw;
    d.out_img_stride = (int64_t)l.cout * h * w;
    d.out_nchw = 1;
  } else {
    d.out = (float*)out;
    d.out_img_stride = (int64_t)r16(l.cout) * h * w;  // (unshuffled: 4 cout x h/2 x w/2 = the same count)
    d.out_unshuffle2 = o.out_u2;
  }
  if (o.res1_u2) {
    d.res1 = (const float*)o.res1_u2;
    d.res1_img_stride = o.res1_stride;
    d.beta1 = 1.f;
    d.res1_u2 = 1;
    d.res1_keep_sign = 1;
  }
  return sr_conv3x3_bf16(&d, stream);
}

int pack(const Plan& P, const float* const* hp, char* blob, hipStream_t stream) {
  // host_params: w0, b0, w1 .. w8, w9, b9
  for (int i = 0; i < kConvs; ++i) {
    const ConvL& l = P.L[i];
    const float* w = hp[i == 0 ? 0 : i + 1];
    const float* b = i == 0 ? hp[1] : i == 9 ? hp[11] : nullptr;
    SR_CHECK_ARG(w && (!l.bias || b), "sr_unet_pack_bf16: null parameter of conv%d", i);
    const float* w3 = w;
    if (l.k == 4) {
      float* e = (float*)(blob + l.w3_off);
      SR_TRY(sr_conv4x4s2_weight_as_3x3_f32(const_cast<float*>(w), e, l.cout, l.cin, 0, stream));
      w3 = e;
    }
    SR_TRY(sr_conv3x3_pack_bf16(w3, b, l.cout, l.cin3, l.cin3, 0, 0, blob + l.w_off, l.bias ? (float*)(blob + l.b_off) : nullptr, stream));
    SR_TRY(sr_conv3x3_pack_bf16(w3, nullptr, l.cout, l.cin3, l.cin3, 0, 1, blob + l.dg_off, nullptr, stream));
  }
  return SR_OK;
}

int forward(const Plan& P, const char* blob, const float* x, float* logits, int n, int h, int w, const Saved& S, hipStream_t stream) {
  const int nf = P.nf;
  auto img = [&](int c, int hh, int ww) { return (int64_t)r16(c) * hh * ww; };
  SR_TRY(sr_nchw_to_cb16_bf16(x, S.xin, n, P.cin0, h, w, 1, r16(P.cin0) / 16, img(P.cin0, h, w), stream));
  ConvOpt o;
  o.act = kSlope;
  o.out_u2 = 1;
  SR_TRY(conv_fwd(P.L[0], blob, S.xin, S.u0, n, h, w, o, stream));                  // u0
  SR_TRY(conv_fwd(P.L[1], blob, S.u0, S.u1, n, h / 2, w / 2, o, stream));            // u1: the source already is the unshuffled tensor
  SR_TRY(conv_fwd(P.L[2], blob, S.u1, S.u2, n, h / 4, w / 4, o, stream));            // u2
  o.out_u2 = 0;
  SR_TRY(conv_fwd(P.L[3], blob, S.u2, S.x3, n, h / 8, w / 8, o, stream));            // x3
  SR_TRY(sr_bilinear2x_fwd_bf16(S.x3, img(8 * nf, h / 8, w / 8), nullptr, 0, S.b3, img(8 * nf, h / 4, w / 4), n, 8 * nf / 16, h / 8, w / 8, stream));
  SR_TRY(conv_fwd(P.L[4], blob, S.b3, S.x4, n, h / 4, w / 4, o, stream));            // x4
  if (P.skip)
    SR_TRY(sr_bilinear2x_fwd_u2_bf16(S.x4, img(4 * nf, h / 4, w / 4), S.u2, img(4 * nf, h / 4, w / 4), S.b4, img(4 * nf, h / 2, w / 2), n, 4 * nf / 16,
                                     h / 4, w / 4, stream));
  else
    SR_TRY(sr_bilinear2x_fwd_bf16(S.x4, img(4 * nf, h / 4, w / 4), nullptr, 0, S.b4, img(4 * nf, h / 2, w / 2), n, 4 * nf / 16, h / 4, w / 4, stream));
  SR_TRY(conv_fwd(P.L[5], blob, S.b4, S.x5, n, h / 2, w / 2, o, stream));            // x5
  if (P.skip)
    SR_TRY(sr_bilinear2x_fwd_u2_bf16(S.x5, img(2 * nf, h / 2, w / 2), S.u1, img(2 * nf, h / 2, w / 2), S.b5, img(2 * nf, h, w), n, 2 * nf / 16, h / 2,
                                     w / 2, stream));
  else
    SR_TRY(sr_bilinear2x_fwd_bf16(S.x5, img(2 * nf, h / 2, w / 2), nullptr, 0, S.b5, img(2 * nf, h, w), n, 2 * nf / 16, h / 2, w / 2, stream));
  ConvOpt o6;
  o6.act = kSlope;
  if (P.skip) {
    o6.res1_u2 = S.u0;
    o6.res1_stride = img(nf, h, w);
  }
  SR_TRY(conv_fwd(P.L[6], blob, S.b5, S.x6, n, h, w, o6, stream));                   // x6 (+ u0)
  SR_TRY(conv_fwd(P.L[7], blob, S.x6, S.x7, n, h, w, o, stream));
  SR_TRY(conv_fwd(P.L[8], blob, S.x7, S.x8, n, h, w, o, stream));
  ConvOpt o9;
  o9.nchw = logits;
  SR_TRY(conv_fwd(P.L[9], blob, S.x8, nullptr, n, h, w, o9, stream));
  return SR_OK;
}

// data gradient of conv l: dz (cout channels at the conv's output size) -> out (cin3 channels there); mask != null: multiplied by the
// LeakyReLU derivative of the conv's own input (the producer's pre-activation gradient comes out)
int conv_dgrad(const ConvL& l, const char* blob, const char* dz, char* out, int n, int h, int w, const char* mask, hipStream_t stream) {
  sr_conv3x3_desc d = {};
  d.in = (const float*)dz;
  d.in_img_stride = (int64_t)r16(l.cout) * h * w;
  d.cin_pad = r16(l.cout);
  d.in_h = h;
  d.in_w = w;
  d.wpacked = (const float*)(blob + l.dg_off);
  d.cout = r16(l.cin3);
  d.out = (float*)out;
  d.out_img_stride = (int64_t)r16(l.cin3) * h * w;
  d.n = n;
  d.act_slope = 1.f;
  d.alpha = 1.f;
  if (mask) {
    d.mask_src = (const float*)mask;
    d.mask_img_stride = (int64_t)r16(l.cin3) * h * w;
    d.mask_cbn = r16(l.cin3) / 16;
    d.mask_slope = kSlope;
  } else {
    d.s2_channels = l.s2;
    d.s2_side = 1;
  }
  return sr_conv3x3_bf16(&d, stream);
}

// weight (and bias) gradient of conv l from its forward source and dz; dw / db: destinations shaped like the parameter (null: skipped)
int conv_wgrad(const ConvL& l, const char* src, const char* dz, int n, int h, int w, float* dw, float* db, const Work& W, hipStream_t stream) {
  if (!dw) return SR_OK;
  sr_conv3x3_wgrad_desc g = {};
  g.x = (const float*)src;
  g.x_img_stride = (int64_t)r16(l.cin3) * h * w;
  g.cin_pad = r16(l.cin3);
  g.in_h = h;
  g.in_w = w;
  g.dy = (const float*)dz;
  g.dy_img_stride = (int64_t)r16(l.cout) * h * w;
  g.cout = l.cout;
  g.cin = g.first_seg = l.cin3;
  g.n = n;
  g.scale = 1.f;
  g.dweight = l.k == 4 ? W.dw3 : dw;
  g.dbias = l.bias ? db : nullptr;
  g.slab = W.slab;
  g.slab_bytes = sr_conv3x3_wgrad_slab_bytes_bf16(n, h, w);
  SR_TRY(sr_conv3x3_wgrad_bf16(&g, stream));
  if (l.k == 4) SR_TRY(sr_conv4x4s2_weight_as_3x3_f32(dw, W.dw3, l.cout, l.cin, 1, stream));
  return SR_OK;
}

int backward(const Plan& P, const char* blob, const Saved& S, const float* dlogits, int n, int h, int w, float* const* dp, float* dx,
             const Work& W, hipStream_t stream) {
  const int nf = P.nf;
  auto img = [&](int c, int hh, int ww) { return (int64_t)r16(c) * hh * ww; };
  auto dwp = [&](int i) -> float* { return dp ? dp[i == 0 ? 0 : i + 1] : nullptr; };
  float* db0 = dp ? dp[1] : nullptr;
  float* db9 = dp ? dp[11] : nullptr;
  // conv9 (no activation; its input x8 is conv8's LeakyReLU output: the data gradient comes out premasked)
  SR_TRY(sr_nchw_to_cb16_bf16(dlogits, W.dz9, n, 1, h, w, 1, 1, img(1, h, w), stream));
  SR_TRY(conv_dgrad(P.L[9], blob, W.dz9, W.g8, n, h, w, S.x8, stream));
  SR_TRY(conv_wgrad(P.L[9], S.x8, W.dz9, n, h, w, dwp(9), db9, W, stream));
  // conv8 (premasked gradient; its input x7 is conv7's LeakyReLU output)
  SR_TRY(conv_dgrad(P.L[8], blob, W.g8, W.g7, n, h, w, S.x7, stream));
  SR_TRY(conv_wgrad(P.L[8], S.x7, W.g8, n, h, w, dwp(8), nullptr, W, stream));
  // conv7 (premasked gradient; plain data gradient wrt x6)
  SR_TRY(conv_dgrad(P.L[7], blob, W.g7, W.g6, n, h, w, nullptr, stream));
  SR_TRY(conv_wgrad(P.L[7], S.x6, W.g7, n, h, w, dwp(7), nullptr, W, stream));
  // conv6: x6 = lrelu(conv) [+ u0]; the skip's gradient is g6 itself
  if (P.skip)
    SR_TRY(sr_lrelu_bwd_diff_u2_bf16(W.g6, S.x6, S.u0, W.dz6, kSlope, n, nf / 16, h / 2, w / 2, stream));
  else
    SR_TRY(sr_lrelu_bwd_bf16(W.g6, S.x6, W.dz6, kSlope, (int64_t)n * img(nf, h, w), stream));
  SR_TRY(conv_dgrad(P.L[6], blob, W.dz6, W.gb5, n, h, w, nullptr, stream));
  SR_TRY(conv_wgrad(P.L[6], S.b5, W.dz6, n, h, w, dwp(6), nullptr, W, stream));
  // up(x5 [+ u1]): the resampling gradient, times conv5's LeakyReLU derivative for x5, plain for the skip
  SR_TRY(sr_bilinear2x_bwd_lrelu_bf16(W.gb5, img(2 * nf, h, w), W.g5, img(2 * nf, h / 2, w / 2), S.x5, img(2 * nf, h / 2, w / 2), kSlope,
                                      P.skip ? W.gp5 : nullptr, P.skip ? img(2 * nf, h / 2, w / 2) : 0, n, 2 * nf / 16, h / 2, w / 2, stream));
  SR_TRY(conv_dgrad(P.L[5], blob, W.g5, W.gb4, n, h / 2, w / 2, nullptr, stream));
  SR_TRY(conv_wgrad(P.L[5], S.b4, W.g5, n, h / 2, w / 2, dwp(5), nullptr, W, stream));
  SR_TRY(sr_bilinear2x_bwd_lrelu_bf16(W.gb4, img(4 * nf, h / 2, w / 2), W.g4, img(4 * nf, h / 4, w / 4), S.x4, img(4 * nf, h / 4, w / 4), kSlope,
                                      P.skip ? W.gp4 : nullptr, P.skip ? img(4 * nf, h / 4, w / 4) : 0, n, 4 * nf / 16, h / 4, w / 4, stream));
  SR_TRY(conv_dgrad(P.L[4], blob, W.g4, W.gb3, n, h / 4, w / 4, nullptr, stream));
  SR_TRY(conv_wgrad(P.L[4], S.b3, W.g4, n, h / 4, w / 4, dwp(4), nullptr, W, stream));
  SR_TRY(sr_bilinear2x_bwd_lrelu_bf16(W.gb3, img(8 * nf, h / 4, w / 4), W.g3, img(8 * nf, h / 8, w / 8), S.x3, img(8 * nf, h / 8, w / 8), kSlope, nullptr, 0,
                                      n, 8 * nf / 16, h / 8, w / 8, stream));
  // conv3 (4x4 / s2 on the unshuffled u2): the data gradient stays in u2's layout
  SR_TRY(conv_dgrad(P.L[3], blob, W.g3, W.d2, n, h / 8, w / 8, nullptr, stream));
  SR_TRY(conv_wgrad(P.L[3], S.u2, W.g3, n, h / 8, w / 8, dwp(3), nullptr, W, stream));
  // u2 feeds the skip (gp4) and conv3 (d2): one pass adds them, undoes the unshuffle and applies conv2's LeakyReLU derivative
  SR_TRY(sr_cb16_fork_bwd_u2_bf16(P.skip ? W.gp4 : nullptr, W.d2, S.u2, W.dz2, kSlope, n, 4 * nf / 16, h / 8, w / 8, stream));
  SR_TRY(conv_dgrad(P.L[2], blob, W.dz2, W.d1, n, h / 4, w / 4, nullptr, stream));
  SR_TRY(conv_wgrad(P.L[2], S.u1, W.dz2, n, h / 4, w / 4, dwp(2), nullptr, W, stream));
  SR_TRY(sr_cb16_fork_bwd_u2_bf16(P.skip ? W.gp5 : nullptr, W.d1, S.u1, W.dz1, kSlope, n, 2 * nf / 16, h / 4, w / 4, stream));
  SR_TRY(conv_dgrad(P.L[1], blob, W.dz1, W.d0, n, h / 2, w / 2, nullptr, stream));
  SR_TRY(conv_wgrad(P.L[1], S.u0, W.dz1, n, h / 2, w / 2, dwp(1), nullptr, W, stream));
  SR_TRY(sr_cb16_fork_bwd_u2_bf16(P.skip ? W.g6 : nullptr, W.d0, S.u0, W.dz0, kSlope, n, nf / 16, h / 2, w / 2, stream));
  // conv0
  if (dx) {
    SR_TRY(conv_dgrad(P.L[0], blob, W.dz0, W.dxin, n, h, w, nullptr, stream));
    SR_TRY(sr_cb16_to_nchw_f32(W.dxin, img(P.cin0, h, w), dx, n, P.cin0, h, w, 1, stream));
  }
  SR_TRY(conv_wgrad(P.L[0], S.xin, W.dz0, n, h, w, dwp(0), db0, W, stream));
  return SR_OK;
}

bool shape_ok(int n, int h, int w) { return n > 0 && h > 0 && w > 0 && h % 8 == 0 && w % 8 == 0; }

}  // namespace

extern "C" int sr_unet_num_params(const sr_unet_cfg* cfg) {
  Plan P;
  return make_plan(cfg, &P) ? 12 : 0;
}
extern "C" size_t sr_unet_packed_bytes_bf16(const sr_unet_cfg* cfg) {
  Plan P;
  return make_plan(cfg, &P) ? P.packed_bytes : 0;
}
extern "C" size_t sr_unet_saved_bytes_bf16(const sr_unet_cfg* cfg, int n, int h, int w) {
  Plan P;
  return (make_plan(cfg, &P) && shape_ok(n, h, w)) ? carve_saved(P, n, h, w, nullptr).bytes : 0;
}
extern "C" size_t sr_unet_workspace_bytes_bf16(const sr_unet_cfg* cfg, int n, int h, int w) {
  Plan P;
  return (make_plan(cfg, &P) && shape_ok(n, h, w)) ? carve_work(P, n, h, w, nullptr).bytes : 0;
}
extern "C" int sr_unet_pack_bf16(const sr_unet_cfg* cfg, const float* const* host_params, void* packed, void* stream) {
  Plan P;
  SR_CHECK_ARG(make_plan(cfg, &P), "sr_unet_pack_bf16: bad configuration (num_feat must be a multiple of 16)");
  SR_CHECK_ARG(host_params && packed, "sr_unet_pack_bf16: null argument");
  return pack(P, host_params, (char*)packed, (hipStream_t)stream);
}
extern "C" int sr_unet_forward_bf16(const sr_unet_cfg* cfg, const void* packed, const float* x, float* logits, int n, int h, int w, void* saved,
                                    size_t saved_bytes, void* stream) {
  Plan P;
  SR_CHECK_ARG(make_plan(cfg, &P), "sr_unet_forward_bf16: bad configuration");
  SR_CHECK_ARG(shape_ok(n, h, w), "sr_unet_forward_bf16: the input size must be a positive multiple of 8");
  SR_CHECK_ARG(packed && x && logits && saved, "sr_unet_forward_bf16: null argument");
  const Saved S = carve_saved(P, n, h, w, (char*)saved);
  SR_CHECK_ARG(saved_bytes >= S.bytes, "sr_unet_forward_bf16: saved block %zu B < %zu B", saved_bytes, S.bytes);
  return forward(P, (const char*)packed, x, logits, n, h, w, S, (hipStream_t)stream);
}
extern "C" int sr_unet_backward_bf16(const sr_unet_cfg* cfg, const void* packed, const void* saved, size_t saved_bytes, const float* dlogits, int n,
                                     int h, int w, float* const* host_dparams, float* dx, void* workspace, size_t workspace_bytes, void* stream) {
  Plan P;
  SR_CHECK_ARG(make_plan(cfg, &P), "sr_unet_backward_bf16: bad configuration");
  SR_CHECK_ARG(shape_ok(n, h, w), "sr_unet_backward_bf16: the input size must be a positive multiple of 8");
  SR_CHECK_ARG(packed && saved && dlogits && workspace, "sr_unet_backward_bf16: null argument");
  const Saved S = carve_saved(P, n, h, w, (char*)saved);
  const Work W = carve_work(P, n, h, w, (char*)workspace);
  SR_CHECK_ARG(saved_bytes >= S.bytes && workspace_bytes >= W.bytes, "sr_unet_backward_bf16: saved %zu/%zu B, workspace %zu/%zu B", saved_bytes,
               S.bytes, workspace_bytes, W.bytes);
  if (host_dparams) {
    const bool any = host_dparams[0] != nullptr;
    for (int i = 0; i < 12; ++i)
      SR_CHECK_ARG((host_dparams[i] != nullptr) == any, "sr_unet_backward_bf16: parameter gradients for all parameters or none");
    if (!any) host_dparams = nullptr;
  }
  return backward(P, (const char*)packed, S, dlogits, n, h, w, host_dparams, dx, W, (hipStream_t)stream);
}
