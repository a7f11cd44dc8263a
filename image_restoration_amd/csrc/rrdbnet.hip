// Whole-generator host drivers: RRDBNet.forward (rrdbnet_arch.py:105-119) and its autograd
// backward as fixed sequences of fused launches on one stream — no allocation, no sync,
// graph-capturable.
//
// HBM plan for an [n][cin][h][w] trunk input (h, w after the optional pixel_unshuffle):
//   xin    CB8 [n][cinb][h][w][8]          network input
//   feat0  CB8 [n][nfb ][h][w][8]          conv_first output (kept for feat + body_feat, :114)
//   cat[q] CB8 [n][nfb+4*gcb][h][w][8]     dense-block concat buffers: conv_k of an RDB writes its
//                                          growth channels straight behind x, so torch.cat (:34-37)
//                                          never exists.  Inference rotates 4 buffers (the three
//                                          RDBs of an RRDB + the next RRDB's input); training keeps
//                                          one per RDB (3*num_block+1) because backward re-reads them.
//   trunk  CB8 [n][nfb][h][w][8], up1 [.. 2h 2w], up2 / hr [.. 4h 4w]
// The RDB residual (x5*0.2 + x, :39) and the RRDB residual (out*0.2 + x, :63) are epilogues of
// conv5:  rdb3.conv5 writes 0.04*conv + 0.2*x_rdb3 + x_rrdb.
//
// Backward mirrors it with a rotating set of 4 concat-GRADIENT buffers G: conv5's data gradient
// writes all 192 channels of G (plus the residual branch on the first 64), conv4..conv1 accumulate
// into the channels below their own slice, and the LeakyReLU backward of x_{k-1} is applied in the
// epilogue of the pass that completes its gradient.  Weight gradients use sr_conv3x3_wgrad_f32.
#include <algorithm>
#include <vector>

#include "sr_internal.h"

namespace {

struct ConvPlan {
  int cout, cin, first_seg, seg, cin_pad;
  size_t w_off, b_off;  // float offsets in the packed (forward) blob
  size_t dg_off;        // float offset in the packed data-gradient blob
};

struct NetPlan {
  int nfp, gcp, cin0, cin0_pad, unshuffle;
  std::vector<ConvPlan> convs;  // state_dict order
  std::vector<size_t> rdb_dg;   // [rdb][step s = 0..4]: transposed-dense-block images in the data-gradient blob
  size_t packed_floats, dgrad_floats;
};

int r8(int v) { return (v + 7) / 8 * 8; }

bool make_plan(const sr_rrdbnet_cfg* c, NetPlan* P) {
  if (!c || c->num_in_ch <= 0 || c->num_out_ch <= 0 || c->num_feat <= 0 || c->num_block < 0 || c->num_grow_ch <= 0)
    return false;
  if (c->scale != 4 && c->scale != 2 && c->scale != 1) return false;
  P->unshuffle = c->scale == 4 ? 1 : (c->scale == 2 ? 2 : 4);  // rrdbnet_arch.py:90-93
  P->cin0 = c->num_in_ch * P->unshuffle * P->unshuffle;
  P->cin0_pad = r8(P->cin0);
  P->nfp = r8(c->num_feat);
  P->gcp = r8(c->num_grow_ch);
  size_t off = 0, dg = 0;
  auto add = [&](int cout, int cin, int first_seg, int seg) {
    ConvPlan cp;
    cp.cout = cout;
    cp.cin = cin;
    cp.first_seg = first_seg;
    cp.seg = seg;
    cp.cin_pad = sr_conv3x3_cin_pad(cin, first_seg, seg);
    cp.w_off = off;
    off += sr::align_up(sr_conv3x3_packed_weight_floats(cout, cp.cin_pad), 64);
    cp.b_off = off;
    off += sr::align_up(sr_conv3x3_packed_bias_floats(cout), 64);
    cp.dg_off = dg;
    dg += sr::align_up(sr_conv3x3_packed_weight_floats(cp.cin_pad, r8(cout)), 64);
    P->convs.push_back(cp);
  };
  const int nf = c->num_feat, gc = c->num_grow_ch;
  add(nf, P->cin0, P->cin0, 0);  // conv_first
  for (int b = 0; b < c->num_block; ++b)
    for (int r = 0; r < 3; ++r) {
      const size_t dg_before = dg;
      for (int k = 1; k <= 4; ++k) add(gc, nf + (k - 1) * gc, nf, gc);  // conv1..conv4 (:21-24)
      add(nf, nf + 4 * gc, nf, gc);                                     // conv5 (:25)
      // the five per-conv data-gradient images are replaced by the five step images of the transposed dense block
      dg = dg_before;
      for (int s = 0; s < 5; ++s) {
        P->rdb_dg.push_back(dg);
        dg += sr::align_up(sr::rdb_dgrad_step_floats(nf, gc, s), 64);
      }
    }
  add(nf, nf, nf, 0);             // conv_body
  add(nf, nf, nf, 0);             // conv_up1
  add(nf, nf, nf, 0);             // conv_up2
  add(nf, nf, nf, 0);             // conv_hr
  add(c->num_out_ch, nf, nf, 0);  // conv_last
  P->packed_floats = off;
  P->dgrad_floats = dg;
  return true;
}

struct Carver {
  char* base;
  size_t off = 0;
  float* take(size_t floats) {
    float* p = (float*)(base + off);
    off += sr::align_up(floats * sizeof(float), 256);
    return p;
  }
};

// Forward workspace.  train = false: 4 rotating concat buffers; train = true: one per RDB + 1.
struct FwdSpace {
  float *xin, *feat0, *trunk, *up1, *up2, *hr, *last;
  std::vector<float*> cat;
  int32_t* sync;     // hand-off words of the dense-block chain launches (sr_conv3x3_chain_f32), one block per image group
  size_t sync_ints;  // ints per block
  size_t bytes;
};
constexpr int kSyncBlocks = 4;
int g_group_min_wgs = 512;  // development switch (sr_dev_set_group_min_wgs): workgroups a forward image group keeps per launch

FwdSpace carve_fwd(const sr_rrdbnet_cfg* c, const NetPlan& P, int n, int h, int w, char* base, bool train) {
  FwdSpace W;
  Carver cv{base};
  const size_t hw = (size_t)h * w;
  const int ctot = P.nfp + 4 * P.gcp;
  W.xin = cv.take((size_t)n * P.cin0_pad * hw);
  W.feat0 = cv.take((size_t)n * P.nfp * hw);
  const int ncat = train ? 3 * c->num_block + 1 : 4;
  for (int i = 0; i < ncat; ++i) W.cat.push_back(cv.take((size_t)n * ctot * hw));
  W.trunk = cv.take((size_t)n * P.nfp * hw);
  W.up1 = cv.take((size_t)n * P.nfp * hw * 4);
  W.up2 = cv.take((size_t)n * P.nfp * hw * 16);
  W.hr = cv.take((size_t)n * P.nfp * hw * 16);
  W.last = c->num_out_ch > 4 ? cv.take((size_t)n * r8(c->num_out_ch) * hw * 16) : nullptr;
  W.sync_ints = sr_conv3x3_chain_sync_ints(n, h, w);
  W.sync = (int32_t*)cv.take(W.sync_ints * kSyncBlocks);  // take() counts floats: same size as int32
  W.bytes = cv.off;
  return W;
}

struct BwdSpace {
  float *dyl, *a16, *b16, *a4, *b4, *dtrunk, *g[4], *dxin;
  void* slab;
  size_t slab_bytes, bytes;
};

BwdSpace carve_bwd(const sr_rrdbnet_cfg* c, const NetPlan& P, int n, int h, int w, char* base) {
  BwdSpace B;
  Carver cv{base};
  const size_t hw = (size_t)h * w;
  const int ctot = P.nfp + 4 * P.gcp;
  B.dyl = cv.take((size_t)n * r8(c->num_out_ch) * hw * 16);
  B.a16 = cv.take((size_t)n * P.nfp * hw * 16);
  B.b16 = cv.take((size_t)n * P.nfp * hw * 16);
  B.a4 = cv.take((size_t)n * P.nfp * hw * 4);
  B.b4 = cv.take((size_t)n * P.nfp * hw * 4);
  B.dtrunk = cv.take((size_t)n * P.nfp * hw);
  for (int i = 0; i < 4; ++i) B.g[i] = cv.take((size_t)n * ctot * hw);
  B.dxin = cv.take((size_t)n * P.cin0_pad * hw);
  B.slab_bytes = std::max(sr_conv3x3_wgrad_slab_bytes(n, 4 * h, 4 * w), sr::rdb_wgrad_slab_bytes_f32(n, h, w, c->num_feat, c->num_grow_ch));
  B.slab = cv.take(B.slab_bytes / sizeof(float));
  B.bytes = cv.off;
  return B;
}

// The launch sequence of one forward over images [0, n) of the (already carved / shifted) workspace.
int forward_body(const sr_rrdbnet_cfg* cfg, const NetPlan& P, const FwdSpace& W, const float* packed, const float* x,
                 float* y, int n, int h, int w, hipStream_t stream, bool train, int32_t* sync) {
  const long long hw = (long long)h * w;
  const int ctot = P.nfp + 4 * P.gcp;
  const long long cat_ns = (long long)ctot * hw, feat_ns = (long long)P.nfp * hw;
  int rc = sr_nchw_to_cb8_f32(x, W.xin, n, cfg->num_in_ch, h, w, P.unshuffle, P.cin0_pad / 8,
                              (long long)P.cin0_pad * hw, stream);
  if (rc) return rc;

  size_t ci = 0;
  auto conv = [&](const float* in, long long in_ns, int ih, int iw, int ups, float* out, long long out_ns, float slope,
                  float alpha, const float* r1, long long r1_ns, float b1, const float* r2, long long r2_ns, float b2,
                  int out_nchw) -> int {
    const ConvPlan& cp = P.convs[ci++];
    sr_conv3x3_desc d = {};
    d.in = in;
    d.in_img_stride = in_ns;
    d.cin_pad = cp.cin_pad;
    d.cin_real = cp.cin;
    d.in_h = ih;
    d.in_w = iw;
    d.upsample = ups;
    d.wpacked = packed + cp.w_off;
    d.bpacked = packed + cp.b_off;
    d.cout = cp.cout;
    d.out = out;
    d.out_img_stride = out_ns;
    d.out_nchw = out_nchw;
    d.n = n;
    d.act_slope = slope;
    d.alpha = alpha;
    d.res1 = r1;
    d.res1_img_stride = r1_ns;
    d.beta1 = b1;
    d.res2 = r2;
    d.res2_img_stride = r2_ns;
    d.beta2 = b2;
    return sr_conv3x3_f32(&d, stream);
  };
  auto cat = [&](int q) { return W.cat[train ? q : (q & 3)]; };

  // conv_first (:112): no activation.  With blocks it lands in the first concat buffer; inference copies it to
  // feat0 for the long skip (the buffer is recycled), training reads it in place.
  const bool blocks = cfg->num_block > 0;
  float* first_dst = blocks ? cat(0) : W.feat0;
  const long long first_ns = blocks ? cat_ns : feat_ns;
  rc = conv(W.xin, (long long)P.cin0_pad * hw, h, w, 0, first_dst, first_ns, 1.f, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0);
  if (rc) return rc;
  const float* skip = W.feat0;
  long long skip_ns = feat_ns;
  if (blocks && train) {
    skip = cat(0);
    skip_ns = cat_ns;
  } else if (blocks) {
    hipError_t e = hipMemcpy2DAsync(W.feat0, feat_ns * sizeof(float), cat(0), cat_ns * sizeof(float),
                                    feat_ns * sizeof(float), n, hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) {
      sr::set_error("sr_rrdbnet_forward: feat0 copy: %s", hipGetErrorString(e));
      return SR_ELAUNCH;
    }
  }
  // body (:113): 3 RDBs per RRDB, each dense block ONE launch (sr_conv3x3_chain_f32: conv1..conv5 with tile-level hand-offs;
  // the entry point runs the five convs one by one when the shape is not eligible)
  auto desc = [&](const float* in, float* out, float slope, float alpha, const float* r1, float b1, const float* r2, float b2) {
    const ConvPlan& cp = P.convs[ci++];
    sr_conv3x3_desc d = {};
    d.in = in;
    d.in_img_stride = cat_ns;
    d.cin_pad = cp.cin_pad;
    d.cin_real = cp.cin;
    d.in_h = h;
    d.in_w = w;
    d.wpacked = packed + cp.w_off;
    d.bpacked = packed + cp.b_off;
    d.cout = cp.cout;
    d.out = out;
    d.out_img_stride = cat_ns;
    d.n = n;
    d.act_slope = slope;
    d.alpha = alpha;
    d.res1 = r1;
    d.res1_img_stride = cat_ns;
    d.beta1 = b1;
    d.res2 = r2;
    d.res2_img_stride = cat_ns;
    d.beta2 = b2;
    return d;
  };
  int chain_call = 0;
  for (int b = 0; b < cfg->num_block; ++b) {
    const float* x_rrdb = cat(3 * b);
    for (int r = 0; r < 3; ++r) {
      float* buf = cat(3 * b + r);
      float* nxt = cat(3 * b + r + 1);
      sr_conv3x3_desc d[5];
      for (int k = 1; k <= 4; ++k)  // x_k = lrelu(conv_k(cat(x, x1..x_{k-1})))  (:33-36)
        d[k - 1] = desc(buf, buf + (long long)(P.nfp + (k - 1) * P.gcp) * hw, 0.2f, 1.f, nullptr, 0.f, nullptr, 0.f);
      if (r < 2)  // x5*0.2 + x (:39)
        d[4] = desc(buf, nxt, 1.f, 0.2f, buf, 1.f, nullptr, 0.f);
      else  // (x5*0.2 + x)*0.2 + x_rrdb (:39, :63)
        d[4] = desc(buf, nxt, 1.f, 0.04f, buf, 0.2f, x_rrdb, 1.f);
      rc = sr_conv3x3_chain_f32(d, 5, sync, chain_call++, stream);
      if (rc) return rc;
    }
  }
  // feat = feat + conv_body(body(feat))  (:113-114)
  const float* body_out = blocks ? cat(3 * cfg->num_block) : W.feat0;
  const long long body_ns = blocks ? cat_ns : feat_ns;
  rc = conv(body_out, body_ns, h, w, 0, W.trunk, feat_ns, 1.f, 1.f, skip, skip_ns, 1.f, nullptr, 0, 0.f, 0);
  if (rc) return rc;
  // head (:116-118)
  rc = conv(W.trunk, feat_ns, h, w, 1, W.up1, feat_ns * 4, 0.2f, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0);
  if (rc) return rc;
  rc = conv(W.up1, feat_ns * 4, 2 * h, 2 * w, 1, W.up2, feat_ns * 16, 0.2f, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0);
  if (rc) return rc;
  rc = conv(W.up2, feat_ns * 16, 4 * h, 4 * w, 0, W.hr, feat_ns * 16, 0.2f, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0);
  if (rc) return rc;
  const long long ohw = hw * 16;
  if (cfg->num_out_ch <= 4) {
    rc = conv(W.hr, feat_ns * 16, 4 * h, 4 * w, 0, y, (long long)cfg->num_out_ch * ohw, 1.f, 1.f, nullptr, 0, 0.f,
              nullptr, 0, 0.f, 1);
    if (rc) return rc;
  } else {
    const long long last_ns = (long long)r8(cfg->num_out_ch) * ohw;
    rc = conv(W.hr, feat_ns * 16, 4 * h, 4 * w, 0, W.last, last_ns, 1.f, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0);
    if (rc) return rc;
    rc = sr_cb8_to_nchw_f32(W.last, last_ns, y, n, cfg->num_out_ch, 4 * h, 4 * w, 1, stream);
    if (rc) return rc;
  }
  return SR_OK;
}

// Side streams for the image-group split of the inference forward (see forward_impl).  One set per host thread;
// created on first use, never destroyed (process lifetime), no device memory.
FwdSpace shift_space(const sr_rrdbnet_cfg* c, const NetPlan& P, const FwdSpace& W, int n0, int h, int w) {
  FwdSpace S = W;
  const size_t hw = (size_t)h * w;
  const int ctot = P.nfp + 4 * P.gcp;
  S.xin += (size_t)n0 * P.cin0_pad * hw;
  S.feat0 += (size_t)n0 * P.nfp * hw;
  for (auto& p : S.cat) p += (size_t)n0 * ctot * hw;
  S.trunk += (size_t)n0 * P.nfp * hw;
  S.up1 += (size_t)n0 * P.nfp * hw * 4;
  S.up2 += (size_t)n0 * P.nfp * hw * 16;
  S.hr += (size_t)n0 * P.nfp * hw * 16;
  if (S.last) S.last += (size_t)n0 * r8(c->num_out_ch) * hw * 16;
  return S;
}

int forward_impl(const sr_rrdbnet_cfg* cfg, const float* packed, const float* x, float* y, int n, int h_in, int w_in,
                 void* workspace, size_t workspace_bytes, hipStream_t stream, bool train) {
  NetPlan P;
  SR_CHECK_ARG(make_plan(cfg, &P), "sr_rrdbnet_forward: bad config");
  SR_CHECK_ARG(packed && x && y && workspace && n > 0 && h_in > 0 && w_in > 0, "sr_rrdbnet_forward: bad argument");
  // pixel_unshuffle divisibility: the reference asserts (arch_util.py:197)
  SR_CHECK_ARG(h_in % P.unshuffle == 0 && w_in % P.unshuffle == 0,
               "sr_rrdbnet_forward: %dx%d input is not divisible by the pixel_unshuffle factor %d", h_in, w_in,
               P.unshuffle);
  SR_CHECK_ARG((uintptr_t)workspace % 256 == 0, "sr_rrdbnet_forward: workspace must be 256-byte aligned");
  const int h = h_in / P.unshuffle, w = w_in / P.unshuffle;
  if (int rc = sr::chain_check("sr_rrdbnet_forward"))
    return rc;  // a hand-off time-out of an earlier call (capi.hip; the fp32 chain launch is opt-in, sr_set_conv_chain_f32)
  const FwdSpace W = carve_fwd(cfg, P, n, h, w, (char*)workspace, train);
  if (W.bytes > workspace_bytes) {
    sr::set_error("sr_rrdbnet_forward: workspace %zu B < required %zu B", workspace_bytes, W.bytes);
    return SR_ENOSPACE;
  }
  // Image-group split: images are independent, so the batch is cut into G groups that run the same launch
  // sequence on G HIP streams (the caller's + side streams forked/joined with events).  Two kernels are then
  // resident at any time and one group's per-launch ramp-up / drain (measured ~11.6 us of a 79-430 us conv) is
  // covered by the other group's steady state.  Each group keeps >= 512 workgroups per launch (2 per CU).
  const long long wg_per_image = (long long)sr::cdiv(w, 32) * sr::cdiv(h, 8);
  int groups = sr::forward_groups();
  if (groups == 0) groups = 1;  // fp32 default: no grouping (include/sr_hip.h)
  while (groups > 1 && (n / groups) * wg_per_image < g_group_min_wgs) --groups;
  if (groups > n) groups = n;
  if (hipMemsetAsync(W.sync, 0, W.sync_ints * kSyncBlocks * sizeof(int32_t), stream) != hipSuccess) {
    sr::set_error("sr_rrdbnet_forward: sync memset failed");
    return SR_ELAUNCH;
  }
  if (groups <= 1 || sr::prof_on()) {
    const int rc = forward_body(cfg, P, W, packed, x, y, n, h, w, stream, train, W.sync);
    sr::chain_watch(W.sync, stream);
    return rc;
  }
  const size_t in_img = (size_t)cfg->num_in_ch * h_in * w_in;
  const size_t out_img = (size_t)cfg->num_out_ch * (size_t)(h * 4) * (w * 4);
  const int rc = sr::run_image_groups(n, groups, stream, [&](int g, int n0, int cnt, hipStream_t s) {
    const FwdSpace S = shift_space(cfg, P, W, n0, h, w);
    return forward_body(cfg, P, S, packed, x + n0 * in_img, y + n0 * out_img, cnt, h, w, s, train,
                        W.sync + (size_t)(g % kSyncBlocks) * W.sync_ints);
  });
  for (int g = 0; g < groups && g < kSyncBlocks; ++g) sr::chain_watch(W.sync + (size_t)g * W.sync_ints, stream);
  return rc;
}

size_t space_bytes(const sr_rrdbnet_cfg* cfg, int n, int h, int w, int which) {
  NetPlan P;
  if (!make_plan(cfg, &P) || n <= 0 || h <= 0 || w <= 0) return 0;
  if (h % P.unshuffle || w % P.unshuffle) return 0;
  h /= P.unshuffle;
  w /= P.unshuffle;
  if (which == 2) return carve_bwd(cfg, P, n, h, w, nullptr).bytes;
  return carve_fwd(cfg, P, n, h, w, nullptr, which == 1).bytes;
}

}  // namespace

extern "C" int sr_rrdbnet_num_params(const sr_rrdbnet_cfg* cfg) {
  NetPlan P;
  if (!make_plan(cfg, &P)) return SR_EINVAL;
  return 2 * (int)P.convs.size();
}

extern "C" size_t sr_rrdbnet_packed_bytes(const sr_rrdbnet_cfg* cfg) {
  NetPlan P;
  if (!make_plan(cfg, &P)) return 0;
  return P.packed_floats * sizeof(float) + sr::pack_table_bytes(P.convs.size());  // images + the pack table (pack_net.hip)
}

extern "C" size_t sr_rrdbnet_packed_dgrad_bytes(const sr_rrdbnet_cfg* cfg) {
  NetPlan P;
  if (!make_plan(cfg, &P)) return 0;
  return P.dgrad_floats * sizeof(float) + sr::pack_table_bytes(P.convs.size());
}

extern "C" size_t sr_rrdbnet_workspace_bytes(const sr_rrdbnet_cfg* cfg, int n, int h, int w) {
  return space_bytes(cfg, n, h, w, 0);
}
extern "C" size_t sr_rrdbnet_saved_bytes(const sr_rrdbnet_cfg* cfg, int n, int h, int w) {
  return space_bytes(cfg, n, h, w, 1);
}
extern "C" size_t sr_rrdbnet_backward_workspace_bytes(const sr_rrdbnet_cfg* cfg, int n, int h, int w) {
  return space_bytes(cfg, n, h, w, 2);
}

extern "C" int sr_rrdbnet_pack_f32(const sr_rrdbnet_cfg* cfg, const float* const* host_params, float* packed,
                                   void* stream) {
  NetPlan P;
  SR_CHECK_ARG(make_plan(cfg, &P), "sr_rrdbnet_pack_f32: bad config");
  SR_CHECK_ARG(host_params && packed, "sr_rrdbnet_pack_f32: null argument");
  std::vector<sr::PackEntry> tab(P.convs.size());  // one launch for all images (pack_net.hip)
  for (size_t i = 0; i < P.convs.size(); ++i) {
    const ConvPlan& cp = P.convs[i];
    SR_CHECK_ARG(host_params[2 * i] && host_params[2 * i + 1], "sr_rrdbnet_pack_f32: null parameter %zu", i);
    sr::PackEntry& e = tab[i];
    e = sr::PackEntry{};
    e.kind = 0;
    e.w[0] = host_params[2 * i];
    e.bias = host_params[2 * i + 1];
    e.out = packed + cp.w_off;
    e.bout = packed + cp.b_off;
    e.cout = cp.cout;
    e.cin = cp.cin;
    e.first_seg = cp.first_seg;
    e.seg = cp.seg > 0 ? cp.seg : 1;
    e.cin_pad = cp.cin_pad;
  }
  return sr::pack_table_run(tab, packed, P.packed_floats * sizeof(float), false, (hipStream_t)stream);
}

extern "C" int sr_rrdbnet_pack_dgrad_f32(const sr_rrdbnet_cfg* cfg, const float* const* host_params,
                                         float* packed_dgrad, void* stream) {
  NetPlan P;
  SR_CHECK_ARG(make_plan(cfg, &P), "sr_rrdbnet_pack_dgrad_f32: bad config");
  SR_CHECK_ARG(host_params && packed_dgrad, "sr_rrdbnet_pack_dgrad_f32: null argument");
  const int n_rdb = 3 * cfg->num_block;
  std::vector<sr::PackEntry> tab;
  for (size_t i = 0; i < P.convs.size(); ++i) {
    const ConvPlan& cp = P.convs[i];
    SR_CHECK_ARG(host_params[2 * i], "sr_rrdbnet_pack_dgrad_f32: null parameter %zu", i);
    if (i >= 1 && i < 1 + 5 * (size_t)n_rdb) continue;  // dense-block convs: packed per step below
    sr::PackEntry e = {};
    e.kind = 1;
    e.w[0] = host_params[2 * i];
    e.out = packed_dgrad + cp.dg_off;
    e.cout = cp.cout;
    e.cin = cp.cin;
    e.first_seg = cp.first_seg;
    e.seg = cp.seg > 0 ? cp.seg : 1;
    e.cin_pad = cp.cin_pad;
    tab.push_back(e);
  }
  for (int q = 0; q < n_rdb; ++q)
    for (int s = 0; s < 5; ++s) {
      sr::PackEntry e = {};
      e.kind = 2;
      for (int k = 0; k < 5; ++k) e.w[k] = host_params[2 * (1 + 5 * q + k)];
      e.out = packed_dgrad + P.rdb_dg[q * 5 + s];
      e.nf = cfg->num_feat;
      e.gc = cfg->num_grow_ch;
      e.s = s;
      e.scale5 = (q % 3 == 2) ? 0.04f : 0.2f;  // x5*0.2 (+ the RRDB's *0.2 for rdb3), rrdbnet_arch.py:39,63
      tab.push_back(e);
    }
  return sr::pack_table_run(tab, packed_dgrad, P.dgrad_floats * sizeof(float), false, (hipStream_t)stream);
}

extern "C" int sr_rrdbnet_forward_f32(const sr_rrdbnet_cfg* cfg, const float* packed, const float* x, float* y, int n,
                                      int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
  return forward_impl(cfg, packed, x, y, n, h, w, workspace, workspace_bytes, (hipStream_t)stream, false);
}

extern "C" int sr_rrdbnet_forward_train_f32(const sr_rrdbnet_cfg* cfg, const float* packed, const float* x, float* y,
                                            int n, int h, int w, void* saved, size_t saved_bytes, void* stream) {
  return forward_impl(cfg, packed, x, y, n, h, w, saved, saved_bytes, (hipStream_t)stream, true);
}

extern "C" int sr_rrdbnet_backward_f32(const sr_rrdbnet_cfg* cfg, const float* packed_dgrad, const void* saved,
                                       size_t saved_bytes, const float* dy, int n, int h_in, int w_in,
                                       float* const* host_dparams, float* dx, void* workspace, size_t workspace_bytes,
                                       int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NetPlan P;
  SR_CHECK_ARG(make_plan(cfg, &P), "sr_rrdbnet_backward_f32: bad config");
  SR_CHECK_ARG(packed_dgrad && saved && dy && host_dparams && workspace && n > 0, "sr_rrdbnet_backward_f32: bad argument");
  SR_CHECK_ARG(h_in % P.unshuffle == 0 && w_in % P.unshuffle == 0, "sr_rrdbnet_backward_f32: bad spatial size");
  SR_CHECK_ARG((uintptr_t)workspace % 256 == 0 && (uintptr_t)saved % 256 == 0,
               "sr_rrdbnet_backward_f32: workspaces must be 256-byte aligned");
  const int h = h_in / P.unshuffle, w = w_in / P.unshuffle;
  const FwdSpace S = carve_fwd(cfg, P, n, h, w, (char*)saved, true);
  const BwdSpace B = carve_bwd(cfg, P, n, h, w, (char*)workspace);
  if (S.bytes > saved_bytes || B.bytes > workspace_bytes) {
    sr::set_error("sr_rrdbnet_backward_f32: saved %zu/%zu B, workspace %zu/%zu B", saved_bytes, S.bytes,
                  workspace_bytes, B.bytes);
    return SR_ENOSPACE;
  }
  const long long hw = (long long)h * w;
  const int nfb = P.nfp / 8, gcb = P.gcp / 8;
  const int ctot = P.nfp + 4 * P.gcp;
  const long long cat_ns = (long long)ctot * hw, feat_ns = (long long)P.nfp * hw;
  const int nconv = (int)P.convs.size();
  int rc;

  // Data-gradient pass of conv `ci`: out = alpha*dgrad(in) [+ residuals on the first nfb blocks] [accumulated]
  // [LeakyReLU-backward mask on blocks mask_cb0..].
  auto dgrad = [&](int ci, const float* in, long long in_ns, int oh, int ow, float* out, long long out_ns, float alpha,
                   const float* r1, long long r1_ns, float b1, const float* r2, long long r2_ns, float b2, int accumulate,
                   const float* mask, long long mask_ns, int mask_cb0, int mask_cbn) -> int {
    const ConvPlan& cp = P.convs[ci];
    sr_conv3x3_desc d = {};
    d.in = in;
    d.in_img_stride = in_ns;
    d.cin_pad = r8(cp.cout);
    d.cin_real = cp.cout;
    d.in_h = oh;
    d.in_w = ow;
    d.wpacked = packed_dgrad + cp.dg_off;
    d.cout = cp.cin_pad;
    d.out = out;
    d.out_img_stride = out_ns;
    d.n = n;
    d.act_slope = 1.f;
    d.alpha = alpha;
    d.res1 = r1;
    d.res1_img_stride = r1_ns;
    d.beta1 = b1;
    d.res2 = r2;
    d.res2_img_stride = r2_ns;
    d.beta2 = b2;
    d.res_cbn = nfb;
    d.accumulate = accumulate;
    d.mask_src = mask;
    d.mask_img_stride = mask_ns;
    d.mask_cb0 = mask_cb0;
    d.mask_cbn = mask_cbn;
    d.mask_slope = 0.2f;
    return sr_conv3x3_f32(&d, stream);
  };
  // Weight gradients go to the lane (sr_internal.h: a side stream on small launches, else the caller's stream): they depend on the
  // data gradients issued so far and only the optimiser waits for them.  Every call is one ticket; the caller's stream waits for a
  // ticket (lane.need) before it overwrites a buffer that job reads.  The slab is the lane's alone.
  sr::WgradLane lane;
  {
    const int mode = sr::backward_overlap();
    lane.begin(stream, mode > 0 || (mode < 0 && (long long)n * hw < 256ll * 16 * 32));
  }
  long long ticket = 0;
  auto wgrad = [&](int ci, const float* xsrc, long long x_ns, int ih, int iw, int ups, const float* dyp,
                   long long dy_ns, float scale) -> int {
    const ConvPlan& cp = P.convs[ci];
    float* dwp = host_dparams[2 * ci];
    float* dbp = host_dparams[2 * ci + 1];
    hipStream_t ws = lane.hand();
    struct Done {  // every exit marks the ticket (the numbering must not depend on frozen parameters)
      sr::WgradLane& l;
      long long k;
      ~Done() { l.done(k); }
    } mark{lane, ticket++};
    if (!dwp) return SR_OK;  // parameter does not need a gradient
    sr_conv3x3_wgrad_desc d = {};
    d.x = xsrc;
    d.x_img_stride = x_ns;
    d.cin_pad = cp.cin_pad;
    d.in_h = ih;
    d.in_w = iw;
    d.upsample = ups;
    d.dy = dyp;
    d.dy_img_stride = dy_ns;
    d.cout = cp.cout;
    d.cin = cp.cin;
    d.first_seg = cp.first_seg;
    d.seg = cp.seg;
    d.n = n;
    d.scale = scale;
    d.dweight = dwp;
    d.dbias = dbp;
    d.accumulate = accumulate;
    d.slab = B.slab;
    d.slab_bytes = B.slab_bytes;
    return sr_conv3x3_wgrad_f32(&d, ws);
  };

  const int i_first = 0, i_body = nconv - 5, i_up1 = nconv - 4, i_up2 = nconv - 3, i_hr = nconv - 2, i_last = nconv - 1;
  const int ocb = r8(cfg->num_out_ch) / 8;
  const long long ohw = hw * 16;
  // dL/dy (NCHW) -> CB8
  rc = sr_nchw_to_cb8_f32(dy, B.dyl, n, cfg->num_out_ch, 4 * h, 4 * w, 1, ocb, (long long)ocb * 8 * ohw, stream);
  if (rc) return rc;
  // conv_last (:118): no activation after it; its input hr = lrelu(conv_hr(..))
  rc = wgrad(i_last, S.hr, feat_ns * 16, 4 * h, 4 * w, 0, B.dyl, (long long)ocb * 8 * ohw, 1.f);
  if (rc) return rc;
  rc = dgrad(i_last, B.dyl, (long long)ocb * 8 * ohw, 4 * h, 4 * w, B.a16, feat_ns * 16, 1.f, nullptr, 0, 0.f, nullptr, 0,
             0.f, 0, S.hr, feat_ns * 16, 0, nfb);  // a16 = dL/d(conv_hr pre-activation)
  if (rc) return rc;
  // conv_hr
  const long long t_hr = ticket;
  rc = wgrad(i_hr, S.up2, feat_ns * 16, 4 * h, 4 * w, 0, B.a16, feat_ns * 16, 1.f);
  if (rc) return rc;
  rc = dgrad(i_hr, B.a16, feat_ns * 16, 4 * h, 4 * w, B.b16, feat_ns * 16, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0,
             S.up2, feat_ns * 16, 0, nfb);  // b16 = dL/d(conv_up2 pre-activation)
  if (rc) return rc;
  // conv_up2 reads up1 through the nearest x2 upsample (:117)
  rc = wgrad(i_up2, S.up1, feat_ns * 4, 2 * h, 2 * w, 1, B.b16, feat_ns * 16, 1.f);
  if (rc) return rc;
  lane.need(t_hr);  // a16 is written again below: conv_hr's weight gradient has read it
  rc = dgrad(i_up2, B.b16, feat_ns * 16, 4 * h, 4 * w, B.a16, feat_ns * 16, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0,
             nullptr, 0, 0, 0);  // a16 = dL/d(upsampled up1)
  if (rc) return rc;
  rc = sr_upsample2x_bwd_f32(B.a16, feat_ns * 16, B.a4, feat_ns * 4, S.up1, feat_ns * 4, 0.2f, n, nfb, 2 * h, 2 * w,
                             stream);  // a4 = dL/d(conv_up1 pre-activation)
  if (rc) return rc;
  // conv_up1 reads trunk through the upsample (:116)
  rc = wgrad(i_up1, S.trunk, feat_ns, h, w, 1, B.a4, feat_ns * 4, 1.f);
  if (rc) return rc;
  rc = dgrad(i_up1, B.a4, feat_ns * 4, 2 * h, 2 * w, B.b4, feat_ns * 4, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0, nullptr,
             0, 0, 0);
  if (rc) return rc;
  rc = sr_upsample2x_bwd_f32(B.b4, feat_ns * 4, B.dtrunk, feat_ns, nullptr, 0, 0.2f, n, nfb, h, w, stream);
  if (rc) return rc;  // dtrunk = dL/d(feat + body_feat)
  // conv_body (:113): input = body output
  const bool blocks = cfg->num_block > 0;
  const float* body_out = blocks ? S.cat[3 * cfg->num_block] : S.feat0;
  const long long body_ns = blocks ? cat_ns : feat_ns;
  rc = wgrad(i_body, body_out, body_ns, h, w, 0, B.dtrunk, feat_ns, 1.f);
  if (rc) return rc;
  int gi = 0;  // G buffer holding the gradient wrt the current block output in its first nfb blocks
  rc = dgrad(i_body, B.dtrunk, feat_ns, h, w, B.g[gi], cat_ns, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0, nullptr, 0, 0, 0);
  if (rc) return rc;
  // body, in reverse.  Each RDB's data gradient is a transposed dense block over the gradient concat buffer
  // D = [dY5 | dY4 | dY3 | dY2 | dY1] (layout.hip: pack_dense_dgrad): step s reads D[0 : nfp+(4-s)*gcp) and writes
  // slice s once, with the LeakyReLU backward of x_s in its epilogue; step 0 writes dL/dx into the next buffer.
  auto step = [&](int q, int sidx, const float* in, int cin_pad, float* out, int cout, const float* r1, float b1,
                  const float* r2, float b2, const float* mask, int mask_cbn) -> int {
    sr_conv3x3_desc d = {};
    d.in = in;
    d.in_img_stride = cat_ns;
    d.cin_pad = cin_pad;
    d.cin_real = cin_pad;
    d.in_h = h;
    d.in_w = w;
    d.wpacked = packed_dgrad + P.rdb_dg[q * 5 + sidx];
    d.cout = cout;
    d.out = out;
    d.out_img_stride = cat_ns;
    d.n = n;
    d.act_slope = 1.f;
    d.alpha = 1.f;
    d.res1 = r1;
    d.res1_img_stride = cat_ns;
    d.beta1 = b1;
    d.res2 = r2;
    d.res2_img_stride = cat_ns;
    d.beta2 = b2;
    d.mask_src = mask;
    d.mask_img_stride = cat_ns;
    d.mask_cb0 = 0;
    d.mask_cbn = mask_cbn;
    d.mask_slope = 0.2f;
    return sr_conv3x3_f32(&d, stream);
  };
  long long block_ticket[4] = {-1, -1, -1, -1};  // last lane job that reads B.g[i]
  for (int b = cfg->num_block - 1; b >= 0; --b) {
    const float* d_rrdb = B.g[gi];  // dL/d(RRDB output)
    for (int r = 2; r >= 0; --r) {
      const int q = 3 * b + r;
      const float* cat = S.cat[q];
      float* D = B.g[gi];                  // D[0:nf] = dL/d(block output), written by the previous step 0 / conv_body
      float* Dn = B.g[(gi + 1) & 3];
      const float s5 = r == 2 ? 0.04f : 0.2f, sres = r == 2 ? 0.2f : 1.f;
      const bool one_launch = sr::rdb_wgrad_f32_enabled();  // the block's five weight gradients as one launch, after its data gradients
      if (!one_launch) {
        rc = wgrad(1 + 5 * q + 4, cat, cat_ns, h, w, 0, D, cat_ns, s5);  // conv5: dY5 = s5 * D[0:nf]
        if (rc) return rc;
      }
      for (int sl = 4; sl >= 1; --sl) {  // dY_sl = lrelu'(x_sl) * sum_{k > sl} W_k[:, x_sl]^T dY_k
        float* dys = D + (long long)(P.nfp + (4 - sl) * P.gcp) * hw;
        rc = step(q, sl, D, P.nfp + (4 - sl) * P.gcp, dys, cfg->num_grow_ch, nullptr, 0.f, nullptr, 0.f,
                  cat + (long long)(P.nfp + (sl - 1) * P.gcp) * hw, gcb);
        if (rc) return rc;
        if (!one_launch) {
          rc = wgrad(1 + 5 * q + (sl - 1), cat, cat_ns, h, w, 0, dys, cat_ns, 1.f);
          if (rc) return rc;
        }
      }
      if (one_launch) {  // D = [dY5 | dY4 | dY3 | dY2 | dY1] is complete: one ticket for the block
        hipStream_t ws = lane.hand();
        rc = sr::rdb_wgrad_f32(cat, D, cat_ns, n, h, w, cfg->num_feat, cfg->num_grow_ch, host_dparams + 2 * (1 + 5 * q), s5, accumulate,
                               B.slab, B.slab_bytes, ws);
        lane.done(ticket++);
        if (rc < 0) return rc;
        if (rc > 0) {  // widths whose tile-group sets do not fit one launch: conv by conv (D is complete, the order is free)
          rc = wgrad(1 + 5 * q + 4, cat, cat_ns, h, w, 0, D, cat_ns, s5);
          for (int sl = 4; sl >= 1 && !rc; --sl)
            rc = wgrad(1 + 5 * q + (sl - 1), cat, cat_ns, h, w, 0, D + (long long)(P.nfp + (4 - sl) * P.gcp) * hw, cat_ns, 1.f);
          if (rc) return rc;
        }
      }
      // dL/dx = sum_k W_k[:, x]^T dY_k + sres * dL/d(out)  (+ dL/d(RRDB out) at the RRDB input, :63)
      lane.need(block_ticket[(gi + 1) & 3]);  // Dn was the D of the block three steps ago: its weight gradients have read it
      rc = step(q, 0, D, P.nfp + 4 * P.gcp, Dn, cfg->num_feat, D, sres, r == 0 ? d_rrdb : nullptr, 1.f, nullptr, 0);
      if (rc) return rc;
      block_ticket[gi] = ticket - 1;
      gi = (gi + 1) & 3;
    }
  }
  // dL/d(conv_first output) = gradient through the body + the long skip (:114)
  rc = sr_cb8_axpby_f32(B.g[gi], cat_ns, B.dtrunk, feat_ns, 1.f, 1.f, n, nfb, h, w, stream);
  if (rc) return rc;
  rc = wgrad(i_first, S.xin, (long long)P.cin0_pad * hw, h, w, 0, B.g[gi], cat_ns, 1.f);
  if (rc) return rc;
  if (dx) {
    rc = dgrad(i_first, B.g[gi], cat_ns, h, w, B.dxin, (long long)P.cin0_pad * hw, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f,
               0, nullptr, 0, 0, 0);
    if (rc) return rc;
    rc = sr_cb8_to_nchw_f32(B.dxin, (long long)P.cin0_pad * hw, dx, n, cfg->num_in_ch, h, w, P.unshuffle, stream);
    if (rc) return rc;
  }
  return SR_OK;
}

extern "C" void sr_dev_set_group_min_wgs(int wgs) { g_group_min_wgs = wgs > 0 ? wgs : 512; }
