// RRDBNet drivers of the bf16 path (extension; the reference is fp32-only, SURVEY.md §0 D5, BASELINE configs 3-4):
// the launch sequences of rrdbnet.hip — RRDBNet.forward (rrdbnet_arch.py:105-119), the same forward under autograd and
// autograd's backward through it (esrgan_model.py:18,47) — on CB16 bf16 activations / activation gradients with bf16
// weight images, fp32 bias, fp32 accumulation and fp32 parameter gradients.  x, y, dy, dx are fp32 NCHW like the fp32
// entry points, so the Python side (archs/rrdbnet_autograd.py) only switches symbols.
#include <vector>

#include "sr_internal.h"

namespace {
int r16(int v) { return (v + 15) / 16 * 16; }

struct ConvPlanH {
  int cout, cin, first_seg, seg, cin_pad;
  size_t w_off, b_off;  // byte offsets in the packed (forward) blob
  size_t dg_off;        // byte offset in the packed data-gradient blob (non-RDB convs)
};
struct NetPlanH {
  int nfp, gcp, cin0, cin0_pad, unshuffle;
  std::vector<ConvPlanH> convs;  // state_dict order
  std::vector<size_t> rdb_dg;    // [rdb][step s = 0..4] byte offsets of the transposed-dense-block images
  size_t packed_bytes, dgrad_bytes;
};

bool make_plan_h(const sr_rrdbnet_cfg* c, NetPlanH* P) {
  if (!c || c->num_in_ch <= 0 || c->num_out_ch <= 0 || c->num_feat <= 0 || c->num_block < 0 || c->num_grow_ch <= 0)
    return false;
  if (c->scale != 4 && c->scale != 2 && c->scale != 1) return false;
  P->unshuffle = c->scale == 4 ? 1 : (c->scale == 2 ? 2 : 4);  // rrdbnet_arch.py:90-93
  P->cin0 = c->num_in_ch * P->unshuffle * P->unshuffle;
  P->cin0_pad = r16(P->cin0);
  P->nfp = r16(c->num_feat);
  P->gcp = r16(c->num_grow_ch);
  size_t off = 0, dg = 0;
  auto add = [&](int cout, int cin, int first_seg, int seg) {
    ConvPlanH cp;
    cp.cout = cout;
    cp.cin = cin;
    cp.first_seg = first_seg;
    cp.seg = seg;
    cp.cin_pad = sr_conv3x3_cin_pad16(cin, first_seg, seg);
    cp.w_off = off;
    off += sr::align_up(sr_conv3x3_packed_weight_elems_bf16(cout, cin, first_seg, seg, 0) * 2, 256);
    cp.b_off = off;
    off += sr::align_up(sr_conv3x3_packed_bias_floats(cout) * 4, 256);
    cp.dg_off = dg;
    dg += sr::align_up(sr_conv3x3_packed_weight_elems_bf16(cout, cin, first_seg, seg, 1) * 2, 256);
    P->convs.push_back(cp);
  };
  const int nf = c->num_feat, gc = c->num_grow_ch;
  add(nf, P->cin0, P->cin0, 0);  // conv_first
  for (int b = 0; b < c->num_block; ++b)
    for (int r = 0; r < 3; ++r) {
      const size_t dg_before = dg;
      for (int k = 1; k <= 4; ++k) add(gc, nf + (k - 1) * gc, nf, gc);
      add(nf, nf + 4 * gc, nf, gc);
      dg = dg_before;  // per-conv images replaced by the five step images of the transposed dense block
      for (int s = 0; s < 5; ++s) {
        P->rdb_dg.push_back(dg);
        dg += sr::align_up(sr::rdb_dgrad_step_elems16(nf, gc, s) * 2, 256);
      }
    }
  add(nf, nf, nf, 0);             // conv_body
  add(nf, nf, nf, 0);             // conv_up1
  add(nf, nf, nf, 0);             // conv_up2
  add(nf, nf, nf, 0);             // conv_hr
  add(c->num_out_ch, nf, nf, 0);  // conv_last
  P->packed_bytes = off;
  P->dgrad_bytes = dg;
  return true;
}

struct CarverH {
  char* base;
  size_t off = 0;
  __bf16* take(size_t elems) {
    __bf16* p = (__bf16*)(base + off);
    off += sr::align_up(elems * 2, 256);
    return p;
  }
};

struct FwdSpaceH {
  __bf16 *xin, *feat0, *trunk, *up1, *up2, *hr;
  std::vector<__bf16*> cat;
  int32_t* sync;      // hand-off words of the dense-block chain launches (sr_conv3x3_chain_bf16), one block per image group
  size_t sync_ints;   // ints per block
  size_t bytes;
};
constexpr int kSyncBlocks = 4;  // image groups of a forward (sr_set_forward_groups <= 4)
// train = false: 4 rotating concat buffers; train = true: one per RDB + 1 (they are the saved activations)
FwdSpaceH carve_fwd_h(const sr_rrdbnet_cfg* c, const NetPlanH& P, int n, int h, int w, char* base, bool train) {
  FwdSpaceH W;
  CarverH cv{base};
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
  W.sync_ints = sr_conv3x3_chain_sync_ints(n, h, w);
  W.sync = (int32_t*)cv.take(W.sync_ints * kSyncBlocks * 2);  // take() counts 2-byte elements
  W.bytes = cv.off;
  return W;
}

struct BwdSpaceH {
  __bf16 *dyl, *a16, *a16b, *b16, *a4, *b4, *dtrunk, *dxin;
  std::vector<__bf16*> g;  // gradient concat buffers: a ring of 4, or one per dense block + 2 when the weight gradients are deferred
  void* slab;
  int32_t* sync;  // hand-off words of the transposed dense blocks' chain launches
  size_t sync_ints;
  size_t slab_bytes, bytes;
};
// sr_set_backward_wgrad_deferred: the generator's weight gradients run on the lane and the backward call returns WITHOUT waiting
// for them; nothing on the caller's stream may wait for the lane either, so every buffer a weight gradient reads stays untouched
// for the rest of the call: one gradient concat buffer per dense block (13.9 GB at batch 32 of 128x128: HBM is 288 GB) and a second
// 512x512 head buffer.
int g_deferred_wgrad = 0;

BwdSpaceH carve_bwd_h(const sr_rrdbnet_cfg* c, const NetPlanH& P, int n, int h, int w, char* base) {
  BwdSpaceH B;
  CarverH cv{base};
  const size_t hw = (size_t)h * w;
  const int ctot = P.nfp + 4 * P.gcp;
  B.dyl = cv.take((size_t)n * r16(c->num_out_ch) * hw * 16);
  B.a16 = cv.take((size_t)n * P.nfp * hw * 16);
  B.a16b = g_deferred_wgrad ? cv.take((size_t)n * P.nfp * hw * 16) : B.a16;
  B.b16 = cv.take((size_t)n * P.nfp * hw * 16);
  B.a4 = cv.take((size_t)n * P.nfp * hw * 4);
  B.b4 = cv.take((size_t)n * P.nfp * hw * 4);
  B.dtrunk = cv.take((size_t)n * P.nfp * hw);
  const int ng = g_deferred_wgrad ? 3 * c->num_block + 2 : 4;
  for (int i = 0; i < ng; ++i) B.g.push_back(cv.take((size_t)n * ctot * hw));
  B.dxin = cv.take((size_t)n * P.cin0_pad * hw);
  B.slab_bytes = sr_conv3x3_wgrad_slab_bytes_bf16(n, 4 * h, 4 * w);
  const size_t rdb_bytes = sr_rdb_wgrad_slab_bytes_bf16(n, h, w, c->num_feat, c->num_grow_ch);
  if (rdb_bytes > B.slab_bytes) B.slab_bytes = rdb_bytes;
  B.slab = cv.take(B.slab_bytes / 2);
  B.sync_ints = sr_conv3x3_chain_sync_ints(n, h, w);
  B.sync = (int32_t*)cv.take(B.sync_ints * 2);  // take() counts 2-byte elements
  B.bytes = cv.off;
  return B;
}

// One image range of the forward on one stream; W already points at the range's first image in every buffer.
int forward_body_h(const sr_rrdbnet_cfg* cfg, const NetPlanH& P, const FwdSpaceH& W, const void* packed, const float* x, float* y,
                   int n, int h, int w, hipStream_t stream, bool train, const char* who, int32_t* sync) {
  const long long hw = (long long)h * w;
  const int ctot = P.nfp + 4 * P.gcp;
  const long long cat_ns = (long long)ctot * hw, feat_ns = (long long)P.nfp * hw;  // bf16 elements
  int rc = sr_nchw_to_cb16_bf16(x, W.xin, n, cfg->num_in_ch, h, w, P.unshuffle, P.cin0_pad / 16, (long long)P.cin0_pad * hw,
                                stream);
  if (rc) return rc;
  size_t ci = 0;
  auto conv = [&](const __bf16* in, long long in_ns, int ih, int iw, int ups, void* out, long long out_ns, float slope,
                  float alpha, const __bf16* r1, long long r1_ns, float b1, const __bf16* r2, long long r2_ns, float b2,
                  int out_nchw) -> int {
    const ConvPlanH& cp = P.convs[ci++];
    sr_conv3x3_desc d = {};
    d.in = (const float*)in;
    d.in_img_stride = in_ns;
    d.cin_pad = cp.cin_pad;
    d.cin_real = cp.cin;
    d.in_h = ih;
    d.in_w = iw;
    d.upsample = ups;
    d.wpacked = (const float*)((const char*)packed + cp.w_off);
    d.bpacked = (const float*)((const char*)packed + cp.b_off);
    d.cout = cp.cout;
    d.out = (float*)out;
    d.out_img_stride = out_ns;
    d.out_nchw = out_nchw;
    d.n = n;
    d.act_slope = slope;
    d.alpha = alpha;
    d.res1 = (const float*)r1;
    d.res1_img_stride = r1_ns;
    d.beta1 = b1;
    d.res2 = (const float*)r2;
    d.res2_img_stride = r2_ns;
    d.beta2 = b2;
    return sr_conv3x3_bf16(&d, stream);
  };
  const bool blocks = cfg->num_block > 0;
  const int ncat = (int)W.cat.size();
  auto catbuf = [&](int q) { return W.cat[train ? q : (q & 3)]; };
  (void)ncat;
  __bf16* first_dst = blocks ? catbuf(0) : W.feat0;
  const long long first_ns = blocks ? cat_ns : feat_ns;
  rc = conv(W.xin, (long long)P.cin0_pad * hw, h, w, 0, first_dst, first_ns, 1.f, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0);
  if (rc) return rc;
  if (blocks) {  // keep conv_first's output for the long skip (:114); the concat buffers rotate / are overwritten
    if (hipMemcpy2DAsync(W.feat0, feat_ns * 2, catbuf(0), cat_ns * 2, feat_ns * 2, n, hipMemcpyDeviceToDevice, stream) != hipSuccess) {
      sr::set_error("%s: feat0 copy failed", who);
      return SR_ELAUNCH;
    }
  }
  // One launch per residual dense block (sr_conv3x3_chain_bf16: conv1..conv5 with tile-level hand-offs; it falls back to five
  // launches by itself when the shape is not eligible).
  auto desc = [&](const __bf16* in, long long in_ns, void* out, long long out_ns, float slope, float alpha, const __bf16* r1,
                  long long r1_ns, float b1, const __bf16* r2, long long r2_ns, float b2) {
    const ConvPlanH& cp = P.convs[ci++];
    sr_conv3x3_desc d = {};
    d.in = (const float*)in;
    d.in_img_stride = in_ns;
    d.cin_pad = cp.cin_pad;
    d.cin_real = cp.cin;
    d.in_h = h;
    d.in_w = w;
    d.wpacked = (const float*)((const char*)packed + cp.w_off);
    d.bpacked = (const float*)((const char*)packed + cp.b_off);
    d.cout = cp.cout;
    d.out = (float*)out;
    d.out_img_stride = out_ns;
    d.n = n;
    d.act_slope = slope;
    d.alpha = alpha;
    d.res1 = (const float*)r1;
    d.res1_img_stride = r1_ns;
    d.beta1 = b1;
    d.res2 = (const float*)r2;
    d.res2_img_stride = r2_ns;
    d.beta2 = b2;
    return d;
  };
  int chain_call = 0;
  for (int b = 0; b < cfg->num_block; ++b) {
    const __bf16* x_rrdb = catbuf(3 * b);
    for (int r = 0; r < 3; ++r) {
      __bf16* buf = catbuf(3 * b + r);
      __bf16* nxt = catbuf(3 * b + r + 1);
      sr_conv3x3_desc d[5];
      for (int k = 1; k <= 4; ++k)
        d[k - 1] = desc(buf, cat_ns, buf + (long long)(P.nfp + (k - 1) * P.gcp) * hw, cat_ns, 0.2f, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f);
      if (r < 2)
        d[4] = desc(buf, cat_ns, nxt, cat_ns, 1.f, 0.2f, buf, cat_ns, 1.f, nullptr, 0, 0.f);
      else
        d[4] = desc(buf, cat_ns, nxt, cat_ns, 1.f, 0.04f, buf, cat_ns, 0.2f, x_rrdb, cat_ns, 1.f);
      rc = sr::conv3x3_chain_bf16(d, 5, sync, chain_call++, stream, /*mids_scratch=*/!train);  // (train: x1..x4 are saved activations)
      if (rc) return rc;
    }
  }
  const __bf16* body_out = blocks ? catbuf(3 * cfg->num_block) : W.feat0;
  const long long body_ns = blocks ? cat_ns : feat_ns;
  rc = conv(body_out, body_ns, h, w, 0, W.trunk, feat_ns, 1.f, 1.f, W.feat0, feat_ns, 1.f, nullptr, 0, 0.f, 0);
  if (rc) return rc;
  rc = conv(W.trunk, feat_ns, h, w, 1, W.up1, feat_ns * 4, 0.2f, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0);
  if (rc) return rc;
  rc = conv(W.up1, feat_ns * 4, 2 * h, 2 * w, 1, W.up2, feat_ns * 16, 0.2f, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0);
  if (rc) return rc;
  rc = conv(W.up2, feat_ns * 16, 4 * h, 4 * w, 0, W.hr, feat_ns * 16, 0.2f, 1.f, nullptr, 0, 0.f, nullptr, 0, 0.f, 0);
  if (rc) return rc;
  return conv(W.hr, feat_ns * 16, 4 * h, 4 * w, 0, y, (long long)cfg->num_out_ch * hw * 16, 1.f, 1.f, nullptr, 0, 0.f, nullptr, 0,
              0.f, 1);
}

FwdSpaceH shift_space_h(const NetPlanH& P, const FwdSpaceH& W, int n0, int h, int w) {
  FwdSpaceH S = W;
  const size_t hw = (size_t)h * w;
  const int ctot = P.nfp + 4 * P.gcp;
  S.xin += (size_t)n0 * P.cin0_pad * hw;
  S.feat0 += (size_t)n0 * P.nfp * hw;
  for (auto& p : S.cat) p += (size_t)n0 * ctot * hw;
  S.trunk += (size_t)n0 * P.nfp * hw;
  S.up1 += (size_t)n0 * P.nfp * hw * 4;
  S.up2 += (size_t)n0 * P.nfp * hw * 16;
  S.hr += (size_t)n0 * P.nfp * hw * 16;
  return S;
}

int forward_h(const sr_rrdbnet_cfg* cfg, const void* packed, const float* x, float* y, int n, int h_in, int w_in,
              void* workspace, size_t workspace_bytes, hipStream_t stream, bool train, const char* who) {
  NetPlanH P;
  SR_CHECK_ARG(make_plan_h(cfg, &P), "%s: bad config", who);
  SR_CHECK_ARG(packed && x && y && workspace && n > 0 && h_in > 0 && w_in > 0, "%s: bad argument", who);
  SR_CHECK_ARG(h_in % P.unshuffle == 0 && w_in % P.unshuffle == 0, "%s: input not divisible by %d", who, P.unshuffle);
  SR_CHECK_ARG((uintptr_t)workspace % 256 == 0, "%s: workspace must be 256-byte aligned", who);
  const int h = h_in / P.unshuffle, w = w_in / P.unshuffle;
  if (int rc = sr::chain_check(who)) return rc;  // a hand-off time-out of an earlier call (capi.hip)
  const FwdSpaceH W = carve_fwd_h(cfg, P, n, h, w, (char*)workspace, train);
  if (W.bytes > workspace_bytes) {
    sr::set_error("%s: workspace %zu B < required %zu B", who, workspace_bytes, W.bytes);
    return SR_ENOSPACE;
  }
  // Image groups on concurrent streams (sr_set_forward_groups): the bf16 layers are 20-80 us launches whose ramp-up, tail
  // and epilogue drain are a third of their duration; a second group's launches fill those gaps.  Each group keeps at
  // least 64 workgroups of 32x32 pixels per launch so the tile dispatch of the conv does not fall to the smallest tiles.
  int groups = sr::forward_groups();
  if (groups == 0) groups = train ? 1 : 4;  // the training forward (one saved buffer per dense block) measured no gain
  const long long wg_per_image = (long long)sr::cdiv(w, 32) * sr::cdiv(h, 32);
  while (groups > 1 && (n / groups) * wg_per_image < 64) --groups;
  // The fused dense-block kernel needs a window of ~6 tile rows in flight (conv_bf16.hip): on large images — the tiler's cells — a
  // group's share of the CUs is too small for it, and one launch fills the chip by itself anyway.
  if (sr::forward_groups() == 0 && sr::chain_mode() >= 3 && cfg->num_feat == 64 && cfg->num_grow_ch == 32 && h % 16 == 0 &&
      6 * sr::cdiv(w, 32) + 2 > 256 / (groups > 1 ? groups : 1))
    groups = 1;
  // hand-off words of the chain launches: zero once per forward, before the image groups fork
  if (hipMemsetAsync(W.sync, 0, W.sync_ints * kSyncBlocks * sizeof(int32_t), stream) != hipSuccess) {
    sr::set_error("%s: sync memset failed", who);
    return SR_ELAUNCH;
  }
  if (groups <= 1 || sr::prof_on()) {
    const int rc = forward_body_h(cfg, P, W, packed, x, y, n, h, w, stream, train, who, W.sync);
    sr::chain_watch(W.sync, stream);
    return rc;
  }
  const size_t in_img = (size_t)cfg->num_in_ch * h_in * w_in;
  const size_t out_img = (size_t)cfg->num_out_ch * (size_t)(h * 4) * (w * 4);
  const int rc = sr::run_image_groups(n, groups, stream, [&](int g, int n0, int cnt, hipStream_t s) {
    return forward_body_h(cfg, P, shift_space_h(P, W, n0, h, w), packed, x + n0 * in_img, y + n0 * out_img, cnt, h, w, s, train,
                          who, W.sync + (size_t)(g % kSyncBlocks) * W.sync_ints);
  });
  for (int g = 0; g < groups && g < kSyncBlocks; ++g) sr::chain_watch(W.sync + (size_t)g * W.sync_ints, stream);
  return rc;
}
}  // namespace

extern "C" size_t sr_rrdbnet_packed_bytes_bf16(const sr_rrdbnet_cfg* cfg) {
  NetPlanH P;
  return make_plan_h(cfg, &P) ? P.packed_bytes + sr::pack_table_bytes(P.convs.size()) : 0;
}

extern "C" size_t sr_rrdbnet_packed_dgrad_bytes_bf16(const sr_rrdbnet_cfg* cfg) {
  NetPlanH P;
  return make_plan_h(cfg, &P) ? P.dgrad_bytes + sr::pack_table_bytes(P.convs.size()) : 0;
}

static size_t fwd_bytes_h(const sr_rrdbnet_cfg* cfg, int n, int h, int w, bool train) {
  NetPlanH P;
  if (!make_plan_h(cfg, &P) || n <= 0 || h <= 0 || w <= 0 || h % P.unshuffle || w % P.unshuffle) return 0;
  return carve_fwd_h(cfg, P, n, h / P.unshuffle, w / P.unshuffle, nullptr, train).bytes;
}
extern "C" size_t sr_rrdbnet_workspace_bytes_bf16(const sr_rrdbnet_cfg* cfg, int n, int h, int w) {
  return fwd_bytes_h(cfg, n, h, w, false);
}
extern "C" size_t sr_rrdbnet_saved_bytes_bf16(const sr_rrdbnet_cfg* cfg, int n, int h, int w) {
  return fwd_bytes_h(cfg, n, h, w, true);
}
extern "C" int sr_set_backward_wgrad_deferred(int on) {
  g_deferred_wgrad = on < 0 ? 0 : on > 2 ? 2 : on;  // 2: the dense blocks' weight gradients start behind the call's data-gradient chain
  return SR_OK;
}

extern "C" size_t sr_rrdbnet_backward_workspace_bytes_bf16(const sr_rrdbnet_cfg* cfg, int n, int h, int w) {
  NetPlanH P;
  if (!make_plan_h(cfg, &P) || n <= 0 || h <= 0 || w <= 0 || h % P.unshuffle || w % P.unshuffle) return 0;
  return carve_bwd_h(cfg, P, n, h / P.unshuffle, w / P.unshuffle, nullptr).bytes;
}

extern "C" int sr_rrdbnet_pack_bf16(const sr_rrdbnet_cfg* cfg, const float* const* host_params, void* packed, void* stream) {
  NetPlanH P;
  SR_CHECK_ARG(make_plan_h(cfg, &P), "sr_rrdbnet_pack_bf16: bad config");
  SR_CHECK_ARG(host_params && packed, "sr_rrdbnet_pack_bf16: null argument");
  std::vector<sr::PackEntry> tab(P.convs.size());  // one launch for all images (pack_net.hip)
  for (size_t i = 0; i < P.convs.size(); ++i) {
    const ConvPlanH& cp = P.convs[i];
    SR_CHECK_ARG(host_params[2 * i] && host_params[2 * i + 1], "sr_rrdbnet_pack_bf16: null parameter %zu", i);
    sr::PackEntry& e = tab[i];
    e = sr::PackEntry{};
    e.kind = 0;
    e.w[0] = host_params[2 * i];
    e.bias = host_params[2 * i + 1];
    e.out = (char*)packed + cp.w_off;
    e.bout = (float*)((char*)packed + cp.b_off);
    e.cout = cp.cout;
    e.cin = cp.cin;
    e.first_seg = cp.first_seg;
    e.seg = cp.seg > 0 ? cp.seg : 1;
    e.cin_pad = cp.cin_pad;
  }
  return sr::pack_table_run(tab, packed, P.packed_bytes, true, (hipStream_t)stream);
}

extern "C" int sr_rrdbnet_pack_dgrad_bf16(const sr_rrdbnet_cfg* cfg, const float* const* host_params, void* packed_dgrad,
                                          void* stream) {
  NetPlanH P;
  SR_CHECK_ARG(make_plan_h(cfg, &P), "sr_rrdbnet_pack_dgrad_bf16: bad config");
  SR_CHECK_ARG(host_params && packed_dgrad, "sr_rrdbnet_pack_dgrad_bf16: null argument");
  const int n_rdb = 3 * cfg->num_block;
  std::vector<sr::PackEntry> tab;
  for (size_t i = 0; i < P.convs.size(); ++i) {
    const ConvPlanH& cp = P.convs[i];
    SR_CHECK_ARG(host_params[2 * i], "sr_rrdbnet_pack_dgrad_bf16: null parameter %zu", i);
    if (i >= 1 && i < 1 + 5 * (size_t)n_rdb) continue;  // dense-block convs: packed per step below
    sr::PackEntry e = {};
    e.kind = 1;
    e.w[0] = host_params[2 * i];
    e.out = (char*)packed_dgrad + cp.dg_off;
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
      e.out = (char*)packed_dgrad + P.rdb_dg[q * 5 + s];
      e.nf = cfg->num_feat;
      e.gc = cfg->num_grow_ch;
      e.s = s;
      e.scale5 = (q % 3 == 2) ? 0.04f : 0.2f;  // x5*0.2 (+ the RRDB's *0.2 for rdb3), rrdbnet_arch.py:39,63
      tab.push_back(e);
    }
  return sr::pack_table_run(tab, packed_dgrad, P.dgrad_bytes, true, (hipStream_t)stream);
}

extern "C" int sr_rrdbnet_forward_bf16(const sr_rrdbnet_cfg* cfg, const void* packed, const float* x, float* y, int n, int h,
                                       int w, void* workspace, size_t workspace_bytes, void* stream) {
  return forward_h(cfg, packed, x, y, n, h, w, workspace, workspace_bytes, (hipStream_t)stream, false, "sr_rrdbnet_forward_bf16");
}

extern "C" int sr_rrdbnet_forward_train_bf16(const sr_rrdbnet_cfg* cfg, const void* packed, const float* x, float* y, int n,
                                             int h, int w, void* saved, size_t saved_bytes, void* stream) {
  return forward_h(cfg, packed, x, y, n, h, w, saved, saved_bytes, (hipStream_t)stream, true, "sr_rrdbnet_forward_train_bf16");
}

extern "C" int sr_rrdbnet_backward_bf16(const sr_rrdbnet_cfg* cfg, const void* packed_dgrad, const void* saved, size_t saved_bytes,
                                        const float* dy, int n, int h_in, int w_in, float* const* host_dparams, float* dx,
                                        void* workspace, size_t workspace_bytes, int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NetPlanH P;
  SR_CHECK_ARG(make_plan_h(cfg, &P), "sr_rrdbnet_backward_bf16: bad config");
  SR_CHECK_ARG(packed_dgrad && saved && dy && host_dparams && workspace && n > 0, "sr_rrdbnet_backward_bf16: bad argument");
  SR_CHECK_ARG(h_in % P.unshuffle == 0 && w_in % P.unshuffle == 0, "sr_rrdbnet_backward_bf16: bad spatial size");
  SR_CHECK_ARG((uintptr_t)workspace % 256 == 0 && (uintptr_t)saved % 256 == 0,
               "sr_rrdbnet_backward_bf16: workspaces must be 256-byte aligned");
  const int h = h_in / P.unshuffle, w = w_in / P.unshuffle;
  if (int rcw = sr::chain_check("sr_rrdbnet_backward_bf16")) return rcw;
  const FwdSpaceH S = carve_fwd_h(cfg, P, n, h, w, (char*)saved, true);
  const BwdSpaceH B = carve_bwd_h(cfg, P, n, h, w, (char*)workspace);
  if (S.bytes > saved_bytes || B.bytes > workspace_bytes) {
    sr::set_error("sr_rrdbnet_backward_bf16: saved %zu/%zu B, workspace %zu/%zu B", saved_bytes, S.bytes, workspace_bytes,
                  B.bytes);
    return SR_ENOSPACE;
  }
  const long long hw = (long long)h * w;
  const int nfb = P.nfp / 16, gcb = P.gcp / 16;
  const int ctot = P.nfp + 4 * P.gcp;
  const long long cat_ns = (long long)ctot * hw, feat_ns = (long long)P.nfp * hw;
  const int nconv = (int)P.convs.size();
  int rc;

  // data gradient of conv `ci`: out = dgrad(in) [LeakyReLU-backward mask of the activation that fed the conv]
  auto dgrad = [&](int ci, const __bf16* in, long long in_ns, int oh, int ow, __bf16* out, long long out_ns,
                   const __bf16* mask, long long mask_ns, int mask_cbn) -> int {
    const ConvPlanH& cp = P.convs[ci];
    sr_conv3x3_desc d = {};
    d.in = (const float*)in;
    d.in_img_stride = in_ns;
    d.cin_pad = r16(cp.cout);
    d.cin_real = cp.cout;
    d.in_h = oh;
    d.in_w = ow;
    d.wpacked = (const float*)((const char*)packed_dgrad + cp.dg_off);
    d.cout = cp.cin_pad;
    d.out = (float*)out;
    d.out_img_stride = out_ns;
    d.n = n;
    d.act_slope = 1.f;
    d.alpha = 1.f;
    d.mask_src = (const float*)mask;
    d.mask_img_stride = mask_ns;
    d.mask_cbn = mask_cbn;
    d.mask_slope = 0.2f;
    return sr_conv3x3_bf16(&d, stream);
  };
  // Weight gradients go to the lane (sr_internal.h: a side stream on small launches, else the caller's stream): they depend on the
  // data gradients issued so far and only the optimiser waits for them.  Ticket numbers count the lane's jobs; the caller's stream
  // waits for a ticket (lane.need) before it overwrites a buffer that job reads.  The slab is the lane's alone.
  sr::WgradLane lane;
  {
    const int mode = sr::backward_overlap();
    lane.begin(stream, g_deferred_wgrad || mode > 0 || (mode < 0 && (long long)n * hw < 256ll * 16 * 32));
  }
  const bool deferred = g_deferred_wgrad && lane.on;  // (off under the launch profiler: everything on the caller's stream)
  const int ng = (int)B.g.size();
  long long ticket = 0;
  auto wgrad = [&](int ci, const __bf16* xsrc, long long x_ns, int ih, int iw, int ups, const __bf16* dyp, long long dy_ns,
                   float scale) -> int {
    const ConvPlanH& cp = P.convs[ci];
    float* dwp = host_dparams[2 * ci];
    float* dbp = host_dparams[2 * ci + 1];
    hipStream_t ws = lane.hand();
    struct Done {  // every exit marks the ticket, also the one of a frozen parameter (the numbering must not depend on it)
      sr::WgradLane& l;
      long long k;
      ~Done() { l.done(k); }
    } mark{lane, ticket++};
    if (!dwp) return SR_OK;  // parameter does not need a gradient
    sr_conv3x3_wgrad_desc d = {};
    d.x = (const float*)xsrc;
    d.x_img_stride = x_ns;
    d.cin_pad = cp.cin_pad;
    d.in_h = ih;
    d.in_w = iw;
    d.upsample = ups;
    d.dy = (const float*)dyp;
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
    return sr_conv3x3_wgrad_bf16(&d, ws);
  };

  const int i_first = 0, i_body = nconv - 5, i_up1 = nconv - 4, i_up2 = nconv - 3, i_hr = nconv - 2, i_last = nconv - 1;
  const int ocb = r16(cfg->num_out_ch) / 16;
  const long long ohw = hw * 16, dyl_ns = (long long)ocb * 16 * ohw;
  rc = sr_nchw_to_cb16_bf16(dy, B.dyl, n, cfg->num_out_ch, 4 * h, 4 * w, 1, ocb, dyl_ns, stream);  // dL/dy (NCHW) -> CB16
  if (rc) return rc;
  // conv_last (:118): no activation after it; its input hr = lrelu(conv_hr(..))
  rc = wgrad(i_last, S.hr, feat_ns * 16, 4 * h, 4 * w, 0, B.dyl, dyl_ns, 1.f);
  if (rc) return rc;
  rc = dgrad(i_last, B.dyl, dyl_ns, 4 * h, 4 * w, B.a16, feat_ns * 16, S.hr, feat_ns * 16, nfb);  // a16 = dL/d(conv_hr pre-act)
  if (rc) return rc;
  const long long t_hr = ticket;
  rc = wgrad(i_hr, S.up2, feat_ns * 16, 4 * h, 4 * w, 0, B.a16, feat_ns * 16, 1.f);
  if (rc) return rc;
  rc = dgrad(i_hr, B.a16, feat_ns * 16, 4 * h, 4 * w, B.b16, feat_ns * 16, S.up2, feat_ns * 16, nfb);  // b16 = dL/d(conv_up2 pre-act)
  if (rc) return rc;
  // conv_up2 reads up1 through the nearest x2 upsample (:117)
  rc = wgrad(i_up2, S.up1, feat_ns * 4, 2 * h, 2 * w, 1, B.b16, feat_ns * 16, 1.f);
  if (rc) return rc;
  if (B.a16b == B.a16) lane.need(t_hr);  // a16 is written again below: conv_hr's weight gradient has read it
  rc = dgrad(i_up2, B.b16, feat_ns * 16, 4 * h, 4 * w, B.a16b, feat_ns * 16, nullptr, 0, 0);  // a16b = dL/d(upsampled up1)
  if (rc) return rc;
  rc = sr_upsample2x_bwd_bf16(B.a16b, feat_ns * 16, B.a4, feat_ns * 4, S.up1, feat_ns * 4, 0.2f, n, nfb, 2 * h, 2 * w, stream);
  if (rc) return rc;  // a4 = dL/d(conv_up1 pre-activation)
  rc = wgrad(i_up1, S.trunk, feat_ns, h, w, 1, B.a4, feat_ns * 4, 1.f);
  if (rc) return rc;
  rc = dgrad(i_up1, B.a4, feat_ns * 4, 2 * h, 2 * w, B.b4, feat_ns * 4, nullptr, 0, 0);
  if (rc) return rc;
  rc = sr_upsample2x_bwd_bf16(B.b4, feat_ns * 4, B.dtrunk, feat_ns, nullptr, 0, 0.2f, n, nfb, h, w, stream);
  if (rc) return rc;  // dtrunk = dL/d(feat + body_feat)
  // conv_body (:113): input = body output
  const bool blocks = cfg->num_block > 0;
  const __bf16* body_out = blocks ? S.cat[3 * cfg->num_block] : S.feat0;
  const long long body_ns = blocks ? cat_ns : feat_ns;
  rc = wgrad(i_body, body_out, body_ns, h, w, 0, B.dtrunk, feat_ns, 1.f);
  if (rc) return rc;
  int gi = 0;  // G buffer holding the gradient wrt the current block output in its first nfb blocks
  rc = dgrad(i_body, B.dtrunk, feat_ns, h, w, B.g[gi], cat_ns, nullptr, 0, 0);
  if (rc) return rc;
  // body, in reverse: each RDB's data gradient is a transposed dense block over D = [dY5 | dY4 | dY3 | dY2 | dY1]
  auto step = [&](int q, int sidx, const __bf16* in, int cin_pad, __bf16* out, int cout, const __bf16* r1, float b1,
                  const __bf16* r2, float b2, const __bf16* mask, int mask_cbn) {
    sr_conv3x3_desc d = {};
    d.in = (const float*)in;
    d.in_img_stride = cat_ns;
    d.cin_pad = cin_pad;
    d.cin_real = cin_pad;
    d.in_h = h;
    d.in_w = w;
    d.wpacked = (const float*)((const char*)packed_dgrad + P.rdb_dg[q * 5 + sidx]);
    d.cout = cout;
    d.out = (float*)out;
    d.out_img_stride = cat_ns;
    d.n = n;
    d.act_slope = 1.f;
    d.alpha = 1.f;
    d.res1 = (const float*)r1;
    d.res1_img_stride = cat_ns;
    d.beta1 = b1;
    d.res2 = (const float*)r2;
    d.res2_img_stride = cat_ns;
    d.beta2 = b2;
    d.mask_src = (const float*)mask;
    d.mask_img_stride = cat_ns;
    d.mask_cbn = mask_cbn;
    d.mask_slope = 0.2f;
    return d;
  };
  if (hipMemsetAsync(B.sync, 0, B.sync_ints * sizeof(int32_t), stream) != hipSuccess) {
    sr::set_error("sr_rrdbnet_backward_bf16: sync memset failed");
    return SR_ELAUNCH;
  }
  int chain_call = 0;
  std::vector<long long> block_ticket(ng, -1);  // lane job that reads B.g[i]
  struct Later {
    const __bf16* cat;
    const __bf16* D;
    int q;
    float s5;
  };
  std::vector<Later> later;  // deferred mode 2: the dense blocks' weight gradients, issued behind the end of the data-gradient chain
  for (int b = cfg->num_block - 1; b >= 0; --b) {
    const __bf16* d_rrdb = B.g[gi];  // dL/d(RRDB output)
    for (int r = 2; r >= 0; --r) {
      const int q = 3 * b + r;
      const __bf16* cat = S.cat[q];
      __bf16* D = B.g[gi];  // D[0:nf] = dL/d(block output)
      __bf16* Dn = B.g[(gi + 1) % ng];
      lane.need(block_ticket[(gi + 1) % ng]);  // ring of 4: Dn was the D of the block three steps ago, its weight gradients have read it
      const float s5 = r == 2 ? 0.04f : 0.2f, sres = r == 2 ? 0.2f : 1.f;
      // The transposed dense block as one chain of five convs over D (sr_conv3x3_chain_bf16: one launch where the shape allows):
      //   dY_sl = lrelu'(x_sl) * sum_{k > sl} W_k[:, x_sl]^T dY_k   for sl = 4..1, each appended to D, then
      //   dL/dx = sum_k W_k[:, x]^T dY_k + sres * dL/d(out)  (+ dL/d(RRDB out) at the RRDB input, :63) into the next D.
      sr_conv3x3_desc dd[5];
      for (int sl = 4; sl >= 1; --sl)
        dd[4 - sl] = step(q, sl, D, P.nfp + (4 - sl) * P.gcp, D + (long long)(P.nfp + (4 - sl) * P.gcp) * hw, cfg->num_grow_ch, nullptr, 0.f,
                          nullptr, 0.f, cat + (long long)(P.nfp + (sl - 1) * P.gcp) * hw, gcb);
      dd[4] = step(q, 0, D, P.nfp + 4 * P.gcp, Dn, cfg->num_feat, D, sres, r == 0 ? d_rrdb : nullptr, 1.f, nullptr, 0);
      rc = sr_conv3x3_chain_bf16(dd, 5, B.sync, chain_call++, stream);
      if (rc) return rc;
      // all five weight gradients of the block in one launch (conv5: dY5 = s5 * D[0:nf]); D is complete and stays intact
      if (deferred && g_deferred_wgrad >= 2) {
        later.push_back({cat, D, q, s5});  // behind the whole data-gradient chain: under whatever the caller issues next
      } else {
        rc = sr::rdb_wgrad_bf16(cat, D, cat_ns, n, h, w, cfg->num_feat, cfg->num_grow_ch, host_dparams + 2 * (1 + 5 * q), s5,
                                accumulate, B.slab, B.slab_bytes, lane.hand());
        block_ticket[gi] = ticket;
        lane.done(ticket++);
        if (rc) return rc;
      }
      gi = (gi + 1) % ng;
    }
  }
  // dL/d(conv_first output) = gradient through the body + the long skip (:114)
  rc = sr_cb16_axpby_bf16(B.g[gi], cat_ns, B.dtrunk, feat_ns, 1.f, 1.f, n, nfb, h, w, stream);
  if (rc) return rc;
  rc = wgrad(i_first, S.xin, (long long)P.cin0_pad * hw, h, w, 0, B.g[gi], cat_ns, 1.f);
  if (rc) return rc;
  if (dx) {
    rc = dgrad(i_first, B.g[gi], cat_ns, h, w, B.dxin, (long long)P.cin0_pad * hw, nullptr, 0, 0);
    if (rc) return rc;
    rc = sr_cb16_to_nchw_f32(B.dxin, (long long)P.cin0_pad * hw, dx, n, cfg->num_in_ch, h, w, P.unshuffle, stream);
    if (rc) return rc;
  }
  if (!later.empty()) {
    hipStream_t ws = lane.hand();  // everything the caller's stream got in this call comes first
    for (const Later& l : later) {
      rc = sr::rdb_wgrad_bf16(l.cat, l.D, cat_ns, n, h, w, cfg->num_feat, cfg->num_grow_ch, host_dparams + 2 * (1 + 5 * l.q), l.s5,
                              accumulate, B.slab, B.slab_bytes, ws);
      if (rc) return rc;
    }
    lane.done(ticket++);
  }
  if (deferred)
    sr::lane_detach(lane);  // the lane's tail is left pending: sr_backward_lane_join makes a stream wait for it
  else
    lane.end();  // the caller's stream (the optimiser step comes next) waits for the last weight gradient
  sr::chain_watch(B.sync, stream);
  return SR_OK;
}
