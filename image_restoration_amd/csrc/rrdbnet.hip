// Whole-generator host driver: RRDBNet.forward (rrdbnet_arch.py:105-119) as a fixed
// sequence of fused conv launches on one stream — no allocation, no sync, graph-capturable.
//
// HBM plan for an [n][cin][h][w] trunk input (h, w after the optional pixel_unshuffle):
//   xin   CB8 [n][cinb][h][w][8]          network input
//   feat0 CB8 [n][nfb ][h][w][8]          conv_first output (kept for feat + body_feat, :114)
//   cat[4] CB8 [n][nfb+4*gcb][h][w][8]    dense-block concat buffers: block k of an RDB writes
//                                         its growth channels straight behind x, so torch.cat
//                                         (:34-37) never exists.  Four buffers rotate: the three
//                                         RDBs of an RRDB + the next RRDB's input.
//   trunk CB8 [n][nfb][h][w][8], up1 [.. 2h 2w], up2 / hr [.. 4h 4w]
// The RDB residual (x5*0.2 + x, :39) and the RRDB residual (out*0.2 + x, :63) are epilogues
// of conv5:  rdb3.conv5 writes 0.04*conv + 0.2*x_rdb3 + x_rrdb.
#include <vector>

#include "sr_internal.h"

namespace {

struct ConvPlan {
  int cout, cin, first_seg, seg, cin_pad;
  size_t w_off, b_off;  // float offsets in the packed blob
};

struct NetPlan {
  int nfp, gcp, cin0, cin0_pad, unshuffle;
  std::vector<ConvPlan> convs;  // state_dict order
  size_t packed_floats;
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
  size_t off = 0;
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
    P->convs.push_back(cp);
  };
  const int nf = c->num_feat, gc = c->num_grow_ch;
  add(nf, P->cin0, P->cin0, 0);  // conv_first
  for (int b = 0; b < c->num_block; ++b)
    for (int r = 0; r < 3; ++r) {
      for (int k = 1; k <= 4; ++k) add(gc, nf + (k - 1) * gc, nf, gc);  // conv1..conv4 (:21-24)
      add(nf, nf + 4 * gc, nf, gc);                                     // conv5 (:25)
    }
  add(nf, nf, nf, 0);              // conv_body
  add(nf, nf, nf, 0);              // conv_up1
  add(nf, nf, nf, 0);              // conv_up2
  add(nf, nf, nf, 0);              // conv_hr
  add(c->num_out_ch, nf, nf, 0);   // conv_last
  P->packed_floats = off;
  return true;
}

struct Workspace {
  float *xin, *feat0, *cat[4], *trunk, *up1, *up2, *hr, *last;
  size_t bytes;
};

Workspace carve(const sr_rrdbnet_cfg* c, const NetPlan& P, int n, int h, int w, char* base) {
  Workspace W;
  size_t off = 0;
  const size_t hw = (size_t)h * w;
  auto take = [&](size_t floats) {
    float* p = (float*)(base + off);
    off += sr::align_up(floats * sizeof(float), 256);
    return p;
  };
  const int ctot = P.nfp + 4 * P.gcp;
  W.xin = take((size_t)n * P.cin0_pad * hw);
  W.feat0 = take((size_t)n * P.nfp * hw);
  for (int i = 0; i < 4; ++i) W.cat[i] = take((size_t)n * ctot * hw);
  W.trunk = take((size_t)n * P.nfp * hw);
  W.up1 = take((size_t)n * P.nfp * hw * 4);
  W.up2 = take((size_t)n * P.nfp * hw * 16);
  W.hr = take((size_t)n * P.nfp * hw * 16);
  W.last = c->num_out_ch > 4 ? take((size_t)n * r8(c->num_out_ch) * hw * 16) : nullptr;
  W.bytes = off;
  return W;
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
  return P.packed_floats * sizeof(float);
}

extern "C" size_t sr_rrdbnet_workspace_bytes(const sr_rrdbnet_cfg* cfg, int n, int h, int w) {
  NetPlan P;
  if (!make_plan(cfg, &P) || n <= 0 || h <= 0 || w <= 0) return 0;
  if (h % P.unshuffle || w % P.unshuffle) return 0;
  return carve(cfg, P, n, h / P.unshuffle, w / P.unshuffle, nullptr).bytes;
}

extern "C" int sr_rrdbnet_pack_f32(const sr_rrdbnet_cfg* cfg, const float* const* host_params, float* packed,
                                   void* stream) {
  NetPlan P;
  SR_CHECK_ARG(make_plan(cfg, &P), "sr_rrdbnet_pack_f32: bad config");
  SR_CHECK_ARG(host_params && packed, "sr_rrdbnet_pack_f32: null argument");
  for (size_t i = 0; i < P.convs.size(); ++i) {
    const ConvPlan& cp = P.convs[i];
    SR_CHECK_ARG(host_params[2 * i] && host_params[2 * i + 1], "sr_rrdbnet_pack_f32: null parameter %zu", i);
    int rc = sr_conv3x3_pack_f32(host_params[2 * i], host_params[2 * i + 1], cp.cout, cp.cin, cp.first_seg, cp.seg, 0,
                                 packed + cp.w_off, packed + cp.b_off, stream);
    if (rc) return rc;
  }
  return SR_OK;
}

extern "C" int sr_rrdbnet_forward_f32(const sr_rrdbnet_cfg* cfg, const float* packed, const float* x, float* y, int n,
                                      int h_in, int w_in, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NetPlan P;
  SR_CHECK_ARG(make_plan(cfg, &P), "sr_rrdbnet_forward_f32: bad config");
  SR_CHECK_ARG(packed && x && y && workspace && n > 0 && h_in > 0 && w_in > 0, "sr_rrdbnet_forward_f32: bad argument");
  // pixel_unshuffle divisibility: the reference asserts (arch_util.py:197)
  SR_CHECK_ARG(h_in % P.unshuffle == 0 && w_in % P.unshuffle == 0,
               "sr_rrdbnet_forward_f32: %dx%d input is not divisible by the pixel_unshuffle factor %d", h_in, w_in,
               P.unshuffle);
  SR_CHECK_ARG((uintptr_t)workspace % 256 == 0, "sr_rrdbnet_forward_f32: workspace must be 256-byte aligned");
  const int h = h_in / P.unshuffle, w = w_in / P.unshuffle;
  const Workspace W = carve(cfg, P, n, h, w, (char*)workspace);
  if (W.bytes > workspace_bytes) {
    sr::set_error("sr_rrdbnet_forward_f32: workspace %zu B < required %zu B", workspace_bytes, W.bytes);
    return SR_ENOSPACE;
  }
  const long long hw = (long long)h * w;
  const int nf = cfg->num_feat, gc = cfg->num_grow_ch;
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

  // conv_first (:112): no activation.  With blocks, it lands in the first concat buffer and is
  // copied to feat0 for the long skip; without blocks it is the trunk itself.
  float* first_dst = cfg->num_block > 0 ? W.cat[0] : W.feat0;
  const long long first_ns = cfg->num_block > 0 ? cat_ns : feat_ns;
  rc = conv(W.xin, (long long)P.cin0_pad * hw, h, w, 0, first_dst, first_ns, 1.f, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0);
  if (rc) return rc;
  if (cfg->num_block > 0) {
    hipError_t e = hipMemcpy2DAsync(W.feat0, feat_ns * sizeof(float), W.cat[0], cat_ns * sizeof(float),
                                    feat_ns * sizeof(float), n, hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) {
      sr::set_error("sr_rrdbnet_forward_f32: feat0 copy: %s", hipGetErrorString(e));
      return SR_ELAUNCH;
    }
  }
  // body (:113): 3 RDBs per RRDB
  int cur = 0;
  for (int b = 0; b < cfg->num_block; ++b) {
    const float* x_rrdb = W.cat[cur];
    for (int r = 0; r < 3; ++r) {
      float* buf = W.cat[(cur + r) & 3];
      float* nxt = W.cat[(cur + r + 1) & 3];
      for (int k = 1; k <= 4; ++k) {  // x_k = lrelu(conv_k(cat(x, x1..x_{k-1})))  (:33-36)
        rc = conv(buf, cat_ns, h, w, 0, buf + (long long)(P.nfp + (k - 1) * P.gcp) * hw, cat_ns, 0.2f, 1.f, nullptr, 0,
                  0.f, nullptr, 0, 0.f, 0);
        if (rc) return rc;
      }
      if (r < 2)  // x5*0.2 + x (:39)
        rc = conv(buf, cat_ns, h, w, 0, nxt, cat_ns, 1.f, 0.2f, buf, cat_ns, 1.f, nullptr, 0, 0.f, 0);
      else  // (x5*0.2 + x)*0.2 + x_rrdb (:39, :63)
        rc = conv(buf, cat_ns, h, w, 0, nxt, cat_ns, 1.f, 0.04f, buf, cat_ns, 0.2f, x_rrdb, cat_ns, 1.f, 0);
      if (rc) return rc;
    }
    cur = (cur + 3) & 3;
  }
  // feat = feat + conv_body(body(feat))  (:113-114)
  const float* body_out = cfg->num_block > 0 ? W.cat[cur] : W.feat0;
  const long long body_ns = cfg->num_block > 0 ? cat_ns : feat_ns;
  rc = conv(body_out, body_ns, h, w, 0, W.trunk, feat_ns, 1.f, 1.f, W.feat0, feat_ns, 1.f, nullptr, 0, 0.f, 0);
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
    rc = sr_cb8_to_nchw_f32(W.last, last_ns, y, n, cfg->num_out_ch, 4 * h, 4 * w, stream);
    if (rc) return rc;
  }
  (void)nf;
  (void)gc;
  return SR_OK;
}
