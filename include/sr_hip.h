/*
 * sr_hip.h — C ABI of the MI355X (gfx950) RRDBNet/ESRGAN hot path.
 *
 * The reference (ChuRuaNh0/Image_Restoration, BasicSR 1.3.3.10 fork) has NO native
 * interface for this path: RRDBNet is a Python nn.Module whose arithmetic is
 * torch.nn.Conv2d / LeakyReLU / cat / interpolate.  The entry points
 * below are therefore "what an FFI for this path would bind": each one names the
 * reference Python it replaces.  Host code (the image_restoration_amd package) reaches them
 * through ctypes; a C/C++ host can link libsr_hip.so directly.
 *
 * Conventions
 *  - every function returns 0 on success, a negative SR_E* code on failure;
 *    sr_last_error() returns a thread-local message for the last failure.
 *  - all pointers are DEVICE pointers unless a parameter is named host_*.
 *  - `stream` is the caller's hipStream_t passed as void*; every launch goes to exactly that
 *    stream (NULL = HIP's default stream, which is what torch.cuda.current_stream() is
 *    unless the caller selected another).  The library holds no stream of its own and is
 *    re-entrant per stream (SURVEY.md §8b "Threading").
 *  - no function allocates device memory or synchronises the device; workspaces
 *    are sized by the *_bytes() helpers and owned by the caller (graph-capturable).
 *
 * Device activation layout "CB8" (channel-blocked, 32-byte pixels):
 *    float feat[N][C/8][H][W][8]     (C padded up to a multiple of 8 with zeros)
 * A channel slice [c0, c0+C') with c0 % 8 == 0 of a wider CB8 tensor is the same
 * layout at pointer + (c0/8)*H*W*8 with the parent's image stride, which is how the
 * dense block's torch.cat (rrdbnet_arch.py:34-37) is never materialised.
 */
#ifndef SR_HIP_H
#define SR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_OK 0
#define SR_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define SR_ELAUNCH (-2)  /* HIP launch/runtime error */
#define SR_ENOSPACE (-3) /* workspace too small */

#define SR_ABI_VERSION 3

/* ABI version of this library (SR_ABI_VERSION it was built with). */
int sr_version(void);
/* Message of the last error on this thread ("" if none). */
const char* sr_last_error(void);

/* ---------------------------------------------------------------- layout ---- */

/* NCHW fp32 (caller tensor) -> CB8.  `unshuffle` in {1,2,4} fuses the reference's
 * pixel_unshuffle (arch_util.py:185-201, used by RRDBNet.forward :106-109): the
 * source is [N][C][H*u][W*u] and the CB8 tensor has C*u*u channels at H x W.
 * dst has `dst_cblocks` channel blocks (>= ceil(C*u*u/8)); pad channels are zeroed. */
int sr_nchw_to_cb8_f32(const float* src, float* dst, int N, int C, int H, int W, int unshuffle,
                       int dst_cblocks, int64_t dst_img_stride, void* stream);
/* CB8 -> NCHW fp32, first C*shuffle^2 channels; `shuffle` in {1,2,4} is the inverse of `unshuffle`
 * above (dst is [N][C][H*shuffle][W*shuffle]); used for the input gradient at scale 2 / 1. */
int sr_cb8_to_nchw_f32(const float* src, int64_t src_img_stride, float* dst, int N, int C, int H, int W, int shuffle,
                       void* stream);

/* Backward of the head's nearest-x2 upsample (autograd of F.interpolate, rrdbnet_arch.py:116-117):
 * dst[n][cb][y][x] = sum of the 2x2 block of g; optional LeakyReLU backward against `mask`
 * (the activation tensor that was upsampled; NULL = none). g is [.., 2h, 2w], dst/mask [.., h, w]. */
int sr_upsample2x_bwd_f32(const float* g, int64_t g_img_stride, float* dst, int64_t dst_img_stride, const float* mask,
                          int64_t mask_img_stride, float mask_slope, int n, int cblocks, int h, int w, void* stream);
/* dst = a*dst + b*src on CB8 windows of `cblocks` channel blocks (residual-branch gradient sums). */
int sr_cb8_axpby_f32(float* dst, int64_t dst_img_stride, const float* src, int64_t src_img_stride, float a, float b,
                     int n, int cblocks, int h, int w, void* stream);

/* ------------------------------------------------------------- conv3x3 ------ */

/* Number of floats of the packed weight / packed bias image of one 3x3 conv with
 * `cin_pad` (multiple of 8) input and `cout` output channels. */
size_t sr_conv3x3_packed_weight_floats(int cout, int cin_pad);
size_t sr_conv3x3_packed_bias_floats(int cout);

/* Pack nn.Conv2d(k=3) parameters (weight OIHW [cout][cin][3][3], bias [cout] or NULL)
 * into the MFMA operand image read by sr_conv3x3_f32.
 *   The cin reference channels are a first segment of `first_seg` channels followed by
 *   (cin-first_seg)/seg segments of `seg` channels (seg = 0: none).  In the CB8 source
 *   every segment starts on a multiple of 8 channels: this is where the dense block's
 *   concat order (x, x1..x4; rrdbnet_arch.py:33-37) meets the padded concat buffer.
 *   cin_pad = roundup8(first_seg) + nseg*roundup8(seg) is returned by sr_conv3x3_cin_pad.
 *   mode 0 : forward weights.
 *   mode 1 : data-gradient weights (roles of cin/cout swapped, taps flipped): the
 *            packed image convolves dY (cout channels, padded to 8) into dX (cin_pad
 *            channels, laid out like the forward source); bias is ignored. */
int sr_conv3x3_cin_pad(int cin, int first_seg, int seg);
int sr_conv3x3_pack_f32(const float* weight, const float* bias, int cout, int cin, int first_seg, int seg, int mode,
                        float* wpacked, float* bpacked, void* stream);

typedef struct sr_conv3x3_desc {
  const float* in;        /* CB8 source (channel slice allowed) */
  int64_t in_img_stride;  /* floats between images of `in` */
  int cin_pad;            /* input channels read (multiple of 8) */
  int cin_real;           /* reference input channels (accounting only; 0 = cin_pad) */
  int in_h, in_w;         /* source spatial size; output is (in_h, in_w) or 2x when upsample */
  int upsample;           /* 1: source is nearest-x2 upsampled on the fly
                             (F.interpolate(scale_factor=2, mode='nearest'), rrdbnet_arch.py:116-117) */
  const float* wpacked;   /* from sr_conv3x3_pack_f32 */
  const float* bpacked;   /* may be NULL (no bias) */
  int cout;               /* real output channels */
  float* out;             /* CB8 destination (channel slice allowed), or NCHW when out_nchw */
  int64_t out_img_stride;
  int out_nchw;           /* 1: write plain NCHW [N][cout][H][W] (cout <= 4), used by conv_last */
  int n;                  /* batch */
  float act_slope;        /* LeakyReLU negative slope applied to conv+bias; 1.0f = none */
  float alpha;            /* out = alpha*act(conv+bias) + beta1*res1 + beta2*res2 */
  const float* res1; int64_t res1_img_stride; float beta1;  /* CB8, same shape as out; NULL = none */
  const float* res2; int64_t res2_img_stride; float beta2;
  int res_cbn;            /* residuals apply to the first res_cbn channel blocks of out only (0 = all) */
  int out_h, out_w;       /* destination spatial size; only sr_conv4x4s2_dgrad_f32 needs it (0 elsewhere) */
  int accumulate;         /* 1: out += result (dgrad accumulation into a concat-gradient buffer) */
  const float* mask_src;  /* optional CB8 tensor of mask_cbn channel blocks: where mask_src <= 0 the
                             final value is multiplied by mask_slope — LeakyReLU backward fused on
                             the channel blocks [mask_cb0, mask_cb0+mask_cbn) of out (block 0 of
                             mask_src pairs with block mask_cb0 of out) */
  int64_t mask_img_stride; int mask_cb0, mask_cbn; float mask_slope;
  int s2_channels;        /* sr_conv3x3_bf16 only, 0 = dense.  C > 0 (multiple of 64): the conv carries a 4x4/s2 convolution on
                             a pixel-unshuffled operand of 4C channels (sr_conv4x4s2_weight_as_3x3_f32): per parity class of
                             16-channel blocks only 2x2 of the 9 taps have non-zero weights and the kernel skips the rest
                             (same result: the skipped products are exact zeros) */
  int s2_side;            /* 0: the parity classes are the input channels (forward); 1: the output channels (data gradient) */
  int out_unshuffle2;     /* sr_conv3x3_bf16 only (ABI 2).  1: the CB16 destination is written PIXEL-UNSHUFFLED — out[n][(2 ry + rx) CB + cb]
                             [y / 2][x / 2][16] with ry = y & 1, rx = x & 1, CB = cout / 16 (cout % 16 == 0, even output H and W): the layout
                             sr_cb16_unshuffle2_bf16 produces and the next 4x4 / stride-2 conv of a U-Net encoder reads, so the activation
                             is never stored in the plain layout at all (the *_u2 entry points below read it where it is) */
  int res1_u2;            /* sr_conv3x3_bf16 only.  1: res1 only exists pixel-unshuffled (an out_unshuffle2 tensor of the output's size and
                             channel count) and is read where it is */
  int res1_keep_sign;     /* sr_conv3x3_bf16 only; needs res1, beta1 = 1, alpha > 0, no res2 / mask.  1: out = bf16(act(conv + bias) + res1)
                             is rounded so that sign(out - res1) == sign(conv + bias): where the activation is positive but the sum
                             would round to res1 itself, the next bf16 above res1 is stored (<= 1 ulp, the size of the rounding).
                             A skip connection added in the epilogue then still carries the LeakyReLU mask of the conv:
                             sr_lrelu_bwd_diff_u2_bf16 recovers it exactly from (out, res1) and neither the activation nor a
                             separate sum is ever stored (UNetDiscriminatorSN: conv6 + x0) */
} sr_conv3x3_desc;

/* Fused 3x3 / stride 1 / pad 1 convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces nn.Conv2d + LeakyReLU + cat + residual scale-add of
 * ResidualDenseBlock.forward (rrdbnet_arch.py:32-39), RRDB.forward (:58-63) and the
 * trunk/head convs of RRDBNet.forward (:112-118). */
int sr_conv3x3_f32(const sr_conv3x3_desc* d, void* stream);

/* 4x4 / stride 2 / pad 1 convolution of the discriminators (discriminator_arch.py:22-43: conv*_1;
 * UNetDiscriminatorSN conv1-3) and its data gradient, on the same descriptor:
 *   sr_conv4x4s2_f32       : in [n][cin_pad][in_h][in_w] -> out [n][cout][in_h/2][in_w/2]  (bias/act/alpha/res honoured)
 *   sr_conv4x4s2_dgrad_f32 : in = dY [in_h][in_w], out = dX [out_h][out_w] (set out_h/out_w), cout = channels of dX;
 *                            accumulate / mask_src honoured.
 * Both run as four parity passes of a 2x2-tap MFMA conv; weights come from sr_conv4x4s2_pack_f32
 * (weight OIHW [cout][cin][4][4]; mode 0 forward, mode 1 data gradient). */
size_t sr_conv4x4s2_packed_weight_floats(int cout, int cin, int mode);
int sr_conv4x4s2_pack_f32(const float* weight, const float* bias, int cout, int cin, int mode, float* wpacked,
                          float* bpacked, void* stream);
int sr_conv4x4s2_f32(const sr_conv3x3_desc* d, void* stream);
int sr_conv4x4s2_dgrad_f32(const sr_conv3x3_desc* d, void* stream);

/* Weight / bias gradient of the same convolution (what autograd computes for nn.Conv2d in
 * ESRGANModel.optimize_parameters' backward calls, esrgan_model.py:47,68,72):
 *   dweight[co][ci][tap] (+)= scale * sum_{n,y,x} dy[co][y][x] * x[ci][y+dy-1][x+dx-1],  dbias[co] (+)= scale * sum dy[co]
 * Deterministic: partial sums go through a caller-provided slab, no atomics. */
typedef struct sr_conv3x3_wgrad_desc {
  const float* x;          /* CB8 forward source of the conv (channel slice allowed) */
  int64_t x_img_stride;
  int cin_pad;             /* = sr_conv3x3_cin_pad(cin, first_seg, seg) */
  int in_h, in_w;          /* source spatial size */
  int upsample;            /* 1: the conv read x through the nearest-x2 upsample */
  const float* dy;         /* CB8 gradient wrt the conv's pre-activation output, roundup8(cout) channels */
  int64_t dy_img_stride;
  int cout, cin, first_seg, seg; /* reference channel counts + concat segmentation (see sr_conv3x3_pack_f32) */
  int n;
  float scale;
  float* dweight;          /* OIHW [cout][cin][3][3] */
  float* dbias;            /* [cout] or NULL */
  int accumulate;          /* 1: add into dweight/dbias instead of overwriting */
  void* slab;              /* >= sr_conv3x3_wgrad_slab_bytes(n, out_h, out_w) */
  size_t slab_bytes;
} sr_conv3x3_wgrad_desc;
size_t sr_conv3x3_wgrad_slab_bytes(int n, int out_h, int out_w);
int sr_conv3x3_wgrad_f32(const sr_conv3x3_wgrad_desc* d, void* stream);
/* Same for the 4x4/s2 conv: dweight is [cout][cin][4][4]; x is the conv's source (in_h x in_w), dy its output
 * gradient ((in_h-2)/2+1 rows); upsample/seg unused.  Slab size: sr_conv3x3_wgrad_slab_bytes(n, out_h, out_w). */
int sr_conv4x4s2_wgrad_f32(const sr_conv3x3_wgrad_desc* d, void* stream);

/* ------------------------------------------------- non-conv training ops ---- */
/* All reductions are two-stage and deterministic.  `ws` is a scratch buffer of at least
 * sr_reduce_workspace_bytes(channels) bytes (channels = 8 for the flat reductions). */
size_t sr_reduce_workspace_bytes(int channels);

/* nn.BatchNorm2d(c, affine) + LeakyReLU(slope) on CB8 (discriminator_arch.py:23-49,57-70):
 *   train=1: batch statistics (biased var for normalisation), running stats updated with `momentum` and the
 *            unbiased variance (torch semantics); train=0: running statistics.
 *   save_mean / save_invstd [c] are written for the backward.  slope = 1 gives plain BatchNorm. */
int sr_bn_lrelu_fwd_f32(const float* x, int64_t x_img_stride, float* y, int64_t y_img_stride, int n, int c, int h, int w,
                        const float* gamma, const float* beta, float* running_mean, float* running_var, int train,
                        float momentum, float eps, float slope, float* save_mean, float* save_invstd, void* ws,
                        size_t ws_bytes, void* stream);
/* Backward of the above given dL/dy and the saved OUTPUT y (LeakyReLU mask = y > 0): writes dx, dgamma, dbeta. */
int sr_bn_lrelu_bwd_f32(const float* x, int64_t x_img_stride, const float* dy, int64_t dy_img_stride, const float* y,
                        int64_t y_img_stride, float* dx, int64_t dx_img_stride, int n, int c, int h, int w,
                        const float* gamma, const float* save_mean, const float* save_invstd, int train, float slope,
                        float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream);

/* dz = dy * (y > 0 ? 1 : slope): LeakyReLU backward from the saved OUTPUT (inplace=True semantics of the reference). */
int sr_lrelu_bwd_f32(const float* dy, const float* y, float* dz, float slope, int64_t n, void* stream);

/* nn.Linear(in, out) + LeakyReLU(act_slope; 1 = none) (discriminator_arch.py:45-46,69-71), row-major x [n][in]. */
int sr_linear_fwd_f32(const float* x, const float* w, const float* b, float* y, int n, int in, int out, float act_slope,
                      void* stream);
/* dz [n][out] scratch; dx [n][in], dw [out][in], db [out] may each be NULL. */
int sr_linear_bwd_f32(const float* x, const float* w, const float* y, const float* dy, int n, int in, int out,
                      float act_slope, float* dz, float* dx, float* dw, float* db, void* stream);

/* F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) on CB8 and its backward (UNetDiscriminatorSN).
 * fwd: src [.., h, w] -> dst [.., 2h, 2w]; bwd: g [.., 2h, 2w] -> gsrc [.., h, w]. */
int sr_bilinear2x_fwd_f32(const float* src, int64_t src_img_stride, float* dst, int64_t dst_img_stride, int n, int cblocks,
                          int h, int w, void* stream);
int sr_bilinear2x_bwd_f32(const float* g, int64_t g_img_stride, float* gsrc, int64_t gsrc_img_stride, int n, int cblocks,
                          int h, int w, void* stream);

/* torch.nn.utils.spectral_norm (dim 0, one power iteration, eps 1e-12) on a weight viewed as [rows][cols]:
 *   update=1 (train): v = normalize(W^T u), u = normalize(W v) in place; sigma = u.(W v); w_sn = W / sigma.
 *   update=0 (eval) : stored u, v.     ws >= (rows + 16*cols)*4 bytes.
 * backward (u, v constants): g_worig = (g_wsn - sum(g_wsn*w_sn) * u v^T) / sigma. */
int sr_spectral_norm_fwd_f32(const float* w_orig, float* u, float* v, int rows, int cols, int update, float eps,
                             float* w_sn, float* sigma, void* ws, size_t ws_bytes, void* stream);
int sr_spectral_norm_bwd_f32(const float* g_wsn, const float* w_sn, const float* u, const float* v, const float* sigma,
                             int rows, int cols, float* g_worig, void* ws, size_t ws_bytes, void* stream);
/* The same forward for up to SR_SN_BATCH_MAX weights at once (every spectral-norm layer of a UNetDiscriminatorSN forward): ONE launch
 * per stage of the power iteration for all layers instead of five launches per layer; per layer the arithmetic, its order and the
 * results (u, v, sigma, w_sn) are those of sr_spectral_norm_fwd_f32.  ws >= sum over layers of (rows + 16 cols) * 4 bytes. */
#define SR_SN_BATCH_MAX 16
typedef struct sr_sn_layer {
  const float* w_orig;
  float* u;
  float* v;
  int rows, cols;
  float* w_sn;
  float* sigma;
} sr_sn_layer;
int sr_spectral_norm_fwd_batch_f32(const sr_sn_layer* layers, int n_layers, int update, float eps, void* ws, size_t ws_bytes,
                                   void* stream);

/* out = a + b (skip connections of UNetDiscriminatorSN); n multiple of 4. */
int sr_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);

/* A dependency chain of 3x3 convs over one pixel grid as ONE persistent launch (bf16 path): conv k reads what convs < k wrote —
 * the five convs of a residual dense block, rrdbnet_arch.py:32-39 (x1..x4 grow the concat buffer, conv5 closes the block).  d[0..nconv)
 * are sr_conv3x3_bf16 descriptors with equal n / in_h / in_w.  Work items (conv, 16x32 tile) are claimed from a counter in conv-major
 * order and wait for conv k-1 on the 3x3 tile neighbourhood (write-through stores + agent-scope flags), so ramp-up, tail and the
 * kernel boundary are paid once per chain.  Results are bit-identical to calling sr_conv3x3_bf16 conv by conv, which is also what
 * the entry point does when the chain is not eligible (upsampling, NCHW output, > 64 couts, ragged height, small launches,
 * profiling) or when disabled.  sr_set_conv_chain(mode): 0 = off, 1 = 32-row ring tiles on one workgroup per CU, 2 = 16-row
 * tiles on two workgroups per CU, 3 (default) = the fused dense-block kernel where the five descriptors are one 64 + 4 x 32
 * channel block over a single concat buffer (a dense block, forward or transposed) and six rows of 16x32 tiles fit the CUs the
 * launch may use — one workgroup per tile keeps the partial sums of all unfinished convs in registers, every input is staged
 * once, images may have more tiles than the chip has CUs (tiles are claimed in dependency order; about six tile rows must be
 * resident) — and mode 2 otherwise.
 *   sync        device int32[sr_conv3x3_chain_sync_ints(n, h, w)], zeroed by the caller (hipMemsetAsync) before the first call that
 *               uses it; calls sharing a block pass increasing call_index 0, 1, 2, ... < 256 and the same n / h / w
 *   sync[0]     is raised by the kernel if a wait on a dependency timed out (bounded spins: never a hang) */
size_t sr_conv3x3_chain_sync_ints(int n, int h, int w);
int sr_conv3x3_chain_bf16(const sr_conv3x3_desc* d, int nconv, int32_t* sync, int call_index, void* stream);
int sr_set_conv_chain(int enabled);
/* Hand-off watchdog: SR_ELAUNCH (with a message) if a dense-block launch issued by sr_rrdbnet_*_bf16 on this thread and device timed
 * out waiting for a neighbour tile since the last check — its output was invalid; SR_OK otherwise.  The network entry points run the
 * same check when they are entered (on the abort words they copied to pinned host memory behind their earlier launches, without
 * synchronising), so a time-out surfaces on the next call at the latest; call this after a synchronisation to cover the work before it. */
int sr_chain_watchdog(void);
/* The same fact ON THE DEVICE, for consumers that must not wait for the host to notice: one int32 per device (the calling thread's
 * current device) that the whole-network drivers raise, by a one-wave launch behind their dense-block launches, when a launch's
 * abort word went up.  It stays raised until the host has REPORTED the time-out (sr_chain_watchdog or a driver's entry check returning
 * SR_ELAUNCH: the clear is queued on the null stream at that moment, behind everything issued before the report) or until
 * sr_abort_latch_clear (stream-ordered).  sr_adam_step_f32 / sr_axpby_f32 take it as
 * their abort_word: a training step whose launches timed out then leaves parameters, moments and the EMA shadow exactly as they
 * were, the host's check raises at the next hand-over, and the job can go on from intact state after sr_set_conv_chain(2). */
const int32_t* sr_abort_latch(void);
int sr_abort_latch_clear(void* stream);
/* fp32 twin: same contract and sync block layout; the calls that share a block must be chains of the same shape (cout of every
 * conv), and a block is used by one precision at a time. */
int sr_conv3x3_chain_f32(const sr_conv3x3_desc* d, int nconv, int32_t* sync, int call_index, void* stream);
int sr_set_conv_chain_f32(int enabled);

/* Training input pipeline on the device: crop window + flip / transpose + uint8 -> float32 + channel swap +
 * normalisation of a batch in one launch.  Replaces, per sample on the host: paired_random_crop and augment
 * (basicsr/data/transforms.py:26-158), img2tensor (utils/img_util.py:9-35) on imfrombytes(float32=True) images (:128-132) and
 * the mean / std normalisation of paired_image_dataset.py:94-96.
 *   src  uint8 [n][src_h][src_w][3] HWC in decode order (BGR), images src_img_stride BYTES apart (0 = dense)
 *   top, left  device int32[n], window origin (multiplied by origin_mul: LQ coordinates x scale address the GT tensor); both NULL = 0
 *   sym  device int32[n]: bit0 horizontal flip, bit1 vertical flip, bit2 transpose, applied in that order (NULL = none);
 *        a transposing batch needs patch_h == patch_w
 *   dst  float32 [n][3][patch_h][patch_w] CHW; channel c = source channel (swap_rb ? 2-c : c);
 *        value = (u8 / 255 - mean[c]) / std[c]  with IEEE division (bit-identical to the host pipeline); host_mean3 / host_std3
 *        are HOST pointers to 3 floats or NULL. */
int sr_patch_augment_u8_f32(const uint8_t* src, int64_t src_img_stride, int src_h, int src_w, const int32_t* top,
                            const int32_t* left, int origin_mul, const int32_t* sym, float* dst, int n, int patch_h, int patch_w,
                            int swap_rb, const float* host_mean3, const float* host_std3, void* stream);

/* Gram matrices of the style term (PerceptualLoss._gram_mat, basicsr/losses/losses.py:342-356): gram[n] = F[n] F[n]^T * scale with
 * F[n] = the NCHW fp32 feature map as [c, hw] (scale = 1 / (c h w) in the reference), and the gradient dfeat = (dgram + dgram^T) F *
 * scale.  fp32 FMA in pixel order (deterministic). */
int sr_gram_fwd_f32(const float* feat, int n, int c, int64_t hw, float scale, float* gram, void* stream);
int sr_gram_bwd_f32(const float* feat, const float* dgram, int n, int c, int64_t hw, float scale, float* dfeat, void* stream);

/* Validation PSNR numerator (psnr_ssim.py:8-46 on tensor2img outputs, img_util.py:38-94): per image n,
 * sse[n] = sum over channels and the border-cropped region of (round(clamp(a,0,1)*255) - round(clamp(b,0,1)*255))^2,
 * a, b NCHW float in [0,1].  ws >= n*64 floats. */
int sr_psnr_sse_f32(const float* a, const float* b, int n, int c, int h, int w, int crop_border, float* sse, void* ws,
                    size_t ws_bytes, void* stream);

/* Validation SSIM numerator (psnr_ssim.py:49-128 on tensor2img outputs): per image n, sum[n] = sum over channels and the
 * valid region of the SSIM map of the border-cropped, uint8-quantised images (11x11 Gaussian window, sigma 1.5,
 * C1 = (0.01*255)^2, C2 = (0.03*255)^2); SSIM = sum / (c * (h-2*crop-10) * (w-2*crop-10)).  ws >= n*64 floats. */
int sr_ssim_sum_f32(const float* a, const float* b, int n, int c, int h, int w, int crop_border, float* sum, void* ws,
                    size_t ws_bytes, void* stream);

/* VGG feature extractor of PerceptualLoss (vgg_arch.py:55-162, losses.py:249-356): nn.MaxPool2d(2, 2) on CB8 (floor
 * mode; backward routes a window's gradient to its first maximum in scan order, like torch), the input normalisation
 * y[n][c] = x[n][c] * a[c] + b[c] on NCHW (b NULL = its backward), and a stand-alone LeakyReLU / ReLU (slope 0) for
 * feature layers requested before their activation. */
int sr_maxpool2x2_fwd_f32(const float* x, float* y, int n, int cblocks, int h, int w, void* stream);
int sr_maxpool2x2_bwd_f32(const float* x, const float* dy, float* dx, int n, int cblocks, int h, int w, void* stream);
int sr_channel_affine_f32(const float* x, float* y, const float* a, const float* b, int n, int c, int64_t hw, void* stream);
int sr_lrelu_fwd_f32(const float* x, float* y, float slope, int64_t n, void* stream);

/* out[0] = mean(x) */
int sr_mean_f32(const float* x, int64_t n, float* out, void* ws, size_t ws_bytes, void* stream);
/* L1Loss(loss_weight, reduction='mean') (losses.py:80-106): loss[0] = weight*mean|pred-target|;
 * backward: dpred = gout[0]*weight/n*sign(pred-target). */
int sr_l1_loss_fwd_f32(const float* pred, const float* target, int64_t n, float weight, float* loss, void* ws,
                       size_t ws_bytes, void* stream);
int sr_l1_loss_bwd_f32(const float* pred, const float* target, int64_t n, float weight, const float* gout, float* dpred,
                       void* stream);
/* The other pixel criteria of the reference on the same reduction: kind 0 = L1Loss, 1 = MSELoss (losses.py:165-191),
 * 2 = CharbonnierLoss (:194-227, sqrt((pred - target)^2 + eps)); weight * mean over n elements and the gradient. */
int sr_pixel_loss_fwd_f32(const float* pred, const float* target, int64_t n, int kind, float eps, float weight, float* loss,
                          void* ws, size_t ws_bytes, void* stream);
int sr_pixel_loss_bwd_f32(const float* pred, const float* target, int64_t n, int kind, float eps, float weight, const float* gout,
                          float* dpred, void* stream);
/* The other criteria of GANLoss (losses.py:379-461; the default 'vanilla' with hard labels is sr_bce_logits_*): weight * mean f(x),
 * kind 1 (x - c)^2 'lsgan' (c = label); 2 c*x 'wgan' / hinge generator (c = -1 real, +1 fake); 3 softplus(c*x) 'wgan_softplus';
 * 4 relu(1 + c*x) 'hinge' discriminator; 5 softplus(x) - c*x = BCE-with-logits against a soft label c. */
int sr_gan_point_loss_fwd_f32(const float* x, int64_t n, int kind, float c, float weight, float* loss, void* ws, size_t ws_bytes,
                              void* stream);
int sr_gan_point_loss_bwd_f32(const float* x, int64_t n, int kind, float c, float weight, const float* gout, float* dx, void* stream);
/* GANLoss('vanilla') = BCEWithLogitsLoss (losses.py:379-380,438-461) on z = x - shift[0] (shift = device scalar, the
 * batch mean of the other logits in the relativistic form, esrgan_model.py:40-41,67,71; NULL = 0):
 *   loss[0] = weight*mean(softplus(-z)) for target real, weight*mean(softplus(z)) for target fake;
 *   dsum[0] (optional) = weight*mean(dBCE/dz), the gradient that flows into the subtracted mean.
 *   backward: dx = gout[0]*weight/n*dBCE/dz. */
int sr_bce_logits_fwd_f32(const float* x, const float* shift, int64_t n, int target_is_real, float weight, float* loss,
                          float* dsum, void* ws, size_t ws_bytes, void* stream);
int sr_bce_logits_bwd_f32(const float* x, const float* shift, int64_t n, int target_is_real, float weight,
                          const float* gout, float* dx, void* stream);
/* dx[i] = scale*gout[0]*s[0] for all i (gradient of a subtracted batch mean). */
int sr_fill_scaled_f32(const float* gout, const float* s, float scale, float* dx, int64_t n, void* stream);

/* torch.optim.Adam step (base_model.py:78-83; lr/betas from the yml) on flat fp32 arenas; step counts from 1;
 * grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).
 * abort_word (device int32, may be NULL; normally sr_abort_latch()): when it is non-zero at the time the kernel runs, a launch
 * behind this step's gradients timed out — parameters and moments are left untouched and *skipped (device int32, may be NULL)
 * is incremented, so that the host can take the skipped steps back out of its step count when it learns of the time-out. */
int sr_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr,
                     float beta1, float beta2, float eps, float weight_decay, float grad_scale, const int32_t* abort_word,
                     int32_t* skipped, void* stream);
/* dst = a*dst + b*src (EMA: base_model.py:50-57 with a = decay, b = 1-decay); abort_word as above: non-zero = dst stays. */
int sr_axpby_f32(float* dst, const float* src, float a, float b, int64_t n, const int32_t* abort_word, void* stream);

/* ------------------------------------------------------- whole generator ---- */

typedef struct sr_rrdbnet_cfg {
  int num_in_ch, num_out_ch, scale, num_feat, num_block, num_grow_ch; /* RRDBNet.__init__, rrdbnet_arch.py:87 */
} sr_rrdbnet_cfg;

/* Number of parameter tensors (= len(state_dict), 702 for the 23-block net) in
 * state_dict order: conv_first.{weight,bias}, body.{i}.rdb{1,2,3}.conv{1..5}.{weight,bias},
 * conv_body, conv_up1, conv_up2, conv_hr, conv_last (rrdbnet_arch.py:94-101). */
int sr_rrdbnet_num_params(const sr_rrdbnet_cfg* cfg);
/* Bytes of the packed-parameter blob and of the inference workspace for a
 * [n][num_in_ch][h][w] input (h, w = spatial size of the tensor given to forward). */
size_t sr_rrdbnet_packed_bytes(const sr_rrdbnet_cfg* cfg);
size_t sr_rrdbnet_workspace_bytes(const sr_rrdbnet_cfg* cfg, int n, int h, int w);
/* Pack all parameters.  host_params: HOST array of sr_rrdbnet_num_params() DEVICE pointers. */
int sr_rrdbnet_pack_f32(const sr_rrdbnet_cfg* cfg, const float* const* host_params, float* packed, void* stream);
/* y = RRDBNet.forward(x)  (rrdbnet_arch.py:105-119); x NCHW [n][num_in_ch][h][w] fp32,
 * y NCHW [n][num_out_ch][4h/s'][4w/s'] where s' = 1, 2, 4 for scale 4, 2, 1. */
int sr_rrdbnet_forward_f32(const sr_rrdbnet_cfg* cfg, const float* packed, const float* x, float* y, int n, int h,
                           int w, void* workspace, size_t workspace_bytes, void* stream);

/* Tuning knob (process-wide): the whole-network forwards can cut the batch into `groups` image groups (1..4) that run
 * the same launch sequence concurrently on internal side streams forked from / joined to `stream` with events (no host
 * synchronisation, graph-capturable, results bit-identical).  groups = 0 (default) chooses per path: the fp32 forward uses 1
 * (measured on MI355X the overlap buys < 0.5 %: its per-launch cost is LDS-DMA refill traffic and the output-store drain,
 * not idle CUs); the bf16 inference forward uses up to 4 while every group keeps >= 64 workgroups per launch (its 20-80 us launches
 * spend a third of their time in ramp-up, tail and epilogue drain, which another group's launches fill: +7 % at batch 16,
 * +13 % at batch 32 of 128x128 tiles; the training forward keeps 1, grouping measured no gain there). */
int sr_set_forward_groups(int groups);

/* Training: forward that KEEPS every activation the backward needs (one concat buffer per dense block,
 * 3*num_block+1 of them, plus the head maps) in `saved`, and the matching backward.
 *   sr_rrdbnet_forward_train_f32  == RRDBNet.forward under autograd (esrgan_model.py:18)
 *   sr_rrdbnet_backward_f32       == autograd's backward through it (esrgan_model.py:47): given dL/dy it writes
 *       dL/dparam for every parameter (host_dparams: HOST array of DEVICE pointers in state_dict order, each
 *       shaped like its parameter; NULL entries are skipped; accumulate=1 adds into them, which is how the
 *       gradient lands directly in a flat all-reduce / optimiser arena) and, if dx != NULL, dL/dx.
 * packed_dgrad holds the transposed/flipped weight images (sr_rrdbnet_pack_dgrad_f32). */
size_t sr_rrdbnet_saved_bytes(const sr_rrdbnet_cfg* cfg, int n, int h, int w);
size_t sr_rrdbnet_backward_workspace_bytes(const sr_rrdbnet_cfg* cfg, int n, int h, int w);
size_t sr_rrdbnet_packed_dgrad_bytes(const sr_rrdbnet_cfg* cfg);
int sr_rrdbnet_pack_dgrad_f32(const sr_rrdbnet_cfg* cfg, const float* const* host_params, float* packed_dgrad,
                              void* stream);
int sr_rrdbnet_forward_train_f32(const sr_rrdbnet_cfg* cfg, const float* packed, const float* x, float* y, int n, int h,
                                 int w, void* saved, size_t saved_bytes, void* stream);
int sr_rrdbnet_backward_f32(const sr_rrdbnet_cfg* cfg, const float* packed_dgrad, const void* saved, size_t saved_bytes,
                            const float* dy, int n, int h, int w, float* const* host_dparams, float* dx,
                            void* workspace, size_t workspace_bytes, int accumulate, void* stream);

/* ------------------------------------------------- bf16 path (extension) ---- */
/* The reference has no reduced precision; BASELINE configs 3-4 name bf16.  Activations and their
 * gradients are CB16: __bf16 feat[N][C/16][H][W][16] (32-byte pixels again); weights stay fp32 masters
 * (FlatAdam's arena) and are rounded into bf16 MFMA images per optimiser step; bias, accumulation
 * (v_mfma_f32_32x32x16_bf16), every epilogue, weight gradients and the optimiser are fp32.
 * The entry points mirror their *_f32 twins one to one (same reference counterparts); tensors are passed as void*,
 * descriptors are reused with pointers to bf16 data in the float* fields, *_img_stride in ELEMENTS of the tensor's
 * dtype, cin_pad a multiple of 16 (sr_conv3x3_cin_pad16).  sr_conv3x3_bf16 does not take accumulate / res_cbn /
 * mask_cb0; sr_conv3x3_wgrad_bf16 reads bf16 x and dy and writes fp32 dweight / dbias. */
int sr_nchw_to_cb16_bf16(const float* src, void* dst, int N, int C, int H, int W, int unshuffle, int dst_cblocks,
                         int64_t dst_img_stride, void* stream);
int sr_cb16_to_nchw_f32(const void* src, int64_t src_img_stride, float* dst, int N, int C, int H, int W, int shuffle,
                        void* stream);
int sr_conv3x3_cin_pad16(int cin, int first_seg, int seg);
size_t sr_conv3x3_packed_weight_elems_bf16(int cout, int cin, int first_seg, int seg, int mode);
int sr_conv3x3_pack_bf16(const float* weight, const float* bias, int cout, int cin, int first_seg, int seg, int mode,
                         void* wpacked, float* bpacked, void* stream);
int sr_conv3x3_bf16(const sr_conv3x3_desc* d, void* stream);
size_t sr_conv3x3_wgrad_slab_bytes_bf16(int n, int h, int w);
int sr_conv3x3_wgrad_bf16(const sr_conv3x3_wgrad_desc* d, void* stream);
/* The five weight gradients of one residual dense block (rrdbnet_arch.py:21-25) in one launch: cat = the block's concat
 * buffer [x|x1..x4], D = its gradient concat buffer [dY5|dY4|dY3|dY2|dY1] (both CB16, same image stride);
 * host_dparams[2k], [2k+1] = fp32 dweight / dbias of conv k+1 (NULL weight: skipped); conv5's gradient is scaled by
 * scale5.  Same results as five sr_conv3x3_wgrad_bf16 calls up to fp32 summation order. */
size_t sr_rdb_wgrad_slab_bytes_bf16(int n, int h, int w, int nf, int gc);
int sr_rdb_wgrad_bf16(const void* cat, const void* D, int64_t img_stride, int n, int h, int w, int nf, int gc,
                      float* const* host_dparams, float scale5, int accumulate, void* slab, size_t slab_bytes, void* stream);
int sr_upsample2x_bwd_bf16(const void* g, int64_t g_img_stride, void* dst, int64_t dst_img_stride, const void* mask,
                           int64_t mask_img_stride, float mask_slope, int n, int cblocks, int h, int w, void* stream);
int sr_cb16_axpby_bf16(void* dst, int64_t dst_img_stride, const void* src, int64_t src_img_stride, float a, float b, int n,
                       int cblocks, int h, int w, void* stream);
/* Discriminator-side helpers of the bf16 path (disc_bf16.hip; UNetDiscriminatorSN with compute_dtype = 'bf16').
 * A 4x4 / stride-2 / pad-1 convolution runs as a 3x3 convolution of the pixel-unshuffled input:
 *   sr_cb16_unshuffle2_bf16: [N][C/16][2h][2w][16] -> [N][4C/16][h][w][16], channel (2 ry + rx) C + c (C % 16 == 0;
 *                            cblocks = C/16; inverse = 1 is the adjoint, i.e. the data gradient's way back);
 *   sr_conv4x4s2_weight_as_3x3_f32: W[cout][cin][4][4] -> W'[cout][4 cin][3][3] (16 live taps of 36, the rest zero);
 *                            adjoint = 1 folds a 3x3 weight gradient back into w4.
 * sr_lrelu_bwd_bf16 / sr_bilinear2x_{fwd,bwd}_bf16 are the CB16 twins of the fp32 entry points above;
 * sr_bilinear2x_fwd_bf16 resamples bf16(src + src2) when src2 is given (a skip connection folded into the pass).
 * sr_cb16_add_bf16: out = a + b over n contiguous elements (the skip connections, `x4 = x4 + x2` ...).
 * sr_cb16_fork_bwd_bf16: gradient of an encoder activation x [N][C/16][2h][2w][16] that feeds a skip connection and,
 *   pixel-unshuffled, the next 4x4/s2 convolution: dz = lrelu'(mask) * (g_skip + unshuffle^-1(g_u)) in one pass
 *   (g_u [N][4C/16][h][w][16]; g_skip, mask optional; all tensors contiguous). */
int sr_cb16_unshuffle2_bf16(const void* src, int64_t src_img_stride, void* dst, int64_t dst_img_stride, int n, int cblocks,
                            int h, int w, int inverse, void* stream);
int sr_conv4x4s2_weight_as_3x3_f32(float* w4, float* w3, int cout, int cin, int adjoint, void* stream);
int sr_lrelu_bwd_bf16(const void* gy, const void* y, void* dz, float slope, int64_t n, void* stream);
int sr_cb16_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* CB16 twins of sr_lrelu_fwd_f32 / sr_maxpool2x2_{fwd,bwd}_f32 (VGGFeatureExtractor with compute_dtype = 'bf16'). */
int sr_lrelu_fwd_bf16(const void* x, void* y, float slope, int64_t n, void* stream);
int sr_maxpool2x2_fwd_bf16(const void* x, void* y, int n, int cblocks, int h, int w, void* stream);
int sr_maxpool2x2_bwd_bf16(const void* x, const void* dy, void* dx, int n, int cblocks, int h, int w, void* stream);
int sr_cb16_fork_bwd_bf16(const void* g_skip, const void* g_u, const void* mask, void* dz, float slope, int n, int cblocks, int h,
                          int w, void* stream);
int sr_bilinear2x_fwd_bf16(const void* src, int64_t src_img_stride, const void* src2, int64_t src2_img_stride, void* dst,
                           int64_t dst_img_stride, int n, int cblocks, int h, int w, void* stream);
int sr_bilinear2x_bwd_bf16(const void* g, int64_t g_img_stride, void* gsrc, int64_t gsrc_img_stride, int n, int cblocks, int h,
                           int w, void* stream);
/* The same resampling gradient when the resampled tensor was the LeakyReLU(slope) output `mask` of a layer (UNetDiscriminatorSN's
 * conv3 / conv4 / conv5): gsrc = lrelu'(mask) * gradient — that layer's stand-alone sr_lrelu_bwd_bf16 pass disappears — and, when
 * gplain is not null, gplain = the gradient itself (what a skip connection added before the resampling receives). */
int sr_bilinear2x_bwd_lrelu_bf16(const void* g, int64_t g_img_stride, void* gsrc, int64_t gsrc_img_stride, const void* mask,
                                 int64_t mask_img_stride, float slope, void* gplain, int64_t gplain_img_stride, int n, int cblocks,
                                 int h, int w, void* stream);
/* Readers of an activation that only exists pixel-unshuffled (sr_conv3x3_desc.out_unshuffle2; `u2` arguments are
 * [n][4 cblocks][h][w][16] tensors standing for a plain [n][cblocks][2h][2w][16] one):
 *   sr_cb16_add_u2_bf16          out = a + b_u2                       (a, out plain [2h][2w]; UNet: x6 + x0)
 *   sr_bilinear2x_fwd_u2_bf16    dst = bilinear_x2(src + src2_u2)     (src plain [2h][2w], dst [4h][4w]; UNet: up(x4 + x2), up(x5 + x1))
 *   sr_cb16_fork_bwd_u2_bf16     sr_cb16_fork_bwd_bf16 with the LeakyReLU mask read from the u2 tensor (same index as g_u) */
int sr_cb16_add_u2_bf16(const void* a, const void* b_u2, void* out, int n, int cblocks, int h, int w, void* stream);
int sr_bilinear2x_fwd_u2_bf16(const void* src, int64_t src_img_stride, const void* src2_u2, int64_t src2_img_stride, void* dst,
                              int64_t dst_img_stride, int n, int cblocks, int h, int w, void* stream);
int sr_cb16_fork_bwd_u2_bf16(const void* g_skip, const void* g_u, const void* mask_u2, void* dz, float slope, int n, int cblocks, int h,
                             int w, void* stream);
/* dz = gy * (xsum - x0 > 0 ? 1 : slope): LeakyReLU backward of a conv whose output was stored as xsum = act + x0 with
 * res1_keep_sign; xsum, gy, dz plain [n][cblocks][2h][2w][16], x0_u2 pixel-unshuffled [n][4 cblocks][h][w][16]. */
int sr_lrelu_bwd_diff_u2_bf16(const void* gy, const void* xsum, const void* x0_u2, void* dz, float slope, int n, int cblocks, int h,
                              int w, void* stream);
/* nn.BatchNorm2d + LeakyReLU of VGGStyleDiscriminator128 on CB16 activations (bf16 in / out; statistics, running
 * buffers, gamma / beta and their gradients fp32): twins of sr_bn_lrelu_{fwd,bwd}_f32, same arguments. */
int sr_bn_lrelu_fwd_bf16(const void* x, int64_t x_img_stride, void* y, int64_t y_img_stride, int n, int c, int h, int w,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, int train,
                         float momentum, float eps, float slope, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                         void* stream);
int sr_bn_lrelu_bwd_bf16(const void* x, int64_t x_img_stride, const void* dy, int64_t dy_img_stride, const void* y,
                         int64_t y_img_stride, void* dx, int64_t dx_img_stride, int n, int c, int h, int w, const float* gamma,
                         const float* save_mean, const float* save_invstd, int train, float slope, float* dgamma, float* dbeta,
                         void* ws, size_t ws_bytes, void* stream);
size_t sr_rrdbnet_packed_bytes_bf16(const sr_rrdbnet_cfg* cfg);
size_t sr_rrdbnet_workspace_bytes_bf16(const sr_rrdbnet_cfg* cfg, int n, int h, int w);
int sr_rrdbnet_pack_bf16(const sr_rrdbnet_cfg* cfg, const float* const* host_params, void* packed, void* stream);
/* y (fp32 NCHW) = RRDBNet.forward(x fp32 NCHW) computed in bf16. */
int sr_rrdbnet_forward_bf16(const sr_rrdbnet_cfg* cfg, const void* packed, const float* x, float* y, int n, int h, int w,
                            void* workspace, size_t workspace_bytes, void* stream);
/* Training twins of sr_rrdbnet_forward_train_f32 / sr_rrdbnet_backward_f32: x, y, dy, dx and the parameter
 * gradients (host_dparams) are fp32; saved activations and activation gradients are bf16. */
size_t sr_rrdbnet_saved_bytes_bf16(const sr_rrdbnet_cfg* cfg, int n, int h, int w);
size_t sr_rrdbnet_backward_workspace_bytes_bf16(const sr_rrdbnet_cfg* cfg, int n, int h, int w);
size_t sr_rrdbnet_packed_dgrad_bytes_bf16(const sr_rrdbnet_cfg* cfg);
int sr_rrdbnet_pack_dgrad_bf16(const sr_rrdbnet_cfg* cfg, const float* const* host_params, void* packed_dgrad, void* stream);
int sr_rrdbnet_forward_train_bf16(const sr_rrdbnet_cfg* cfg, const void* packed, const float* x, float* y, int n, int h,
                                  int w, void* saved, size_t saved_bytes, void* stream);
int sr_rrdbnet_backward_bf16(const sr_rrdbnet_cfg* cfg, const void* packed_dgrad, const void* saved, size_t saved_bytes,
                             const float* dy, int n, int h, int w, float* const* host_dparams, float* dx, void* workspace,
                             size_t workspace_bytes, int accumulate, void* stream);

/* -------------------------------------------------- whole VGG discriminator ---- */
/* VGGStyleDiscriminator128 / VGGStyleDiscriminator256 (discriminator_arch.py:6-72, 75-143) as whole-network drivers
 * shaped like the generator's above: one call issues every launch of a forward (:52-72) or of autograd's backward through
 * it (esrgan_model.py:47,68,72) on `stream`; `saved` and `workspace` are caller-owned (sized by the *_bytes helpers), no
 * allocation, no synchronisation, graph-capturable.  Each launch is the single-op entry point the per-layer host path
 * uses, with the same descriptor, in the same order: results are bit-identical to that path.
 *
 * host_params: HOST array of sr_vgg_num_params() DEVICE pointers (fp32 masters) in state_dict order — conv0_0.{weight,bias},
 *   conv0_1.weight, bn0_1.{weight,bias}, then per stage conv{i}_0.weight, bn{i}_0.{weight,bias}, conv{i}_1.weight,
 *   bn{i}_1.{weight,bias}, then linear1.{weight,bias}, linear2.{weight,bias} (33 tensors for the 128 network).
 * host_buffers: HOST array of 3 * sr_vgg_num_batchnorm() DEVICE pointers — running_mean, running_var (fp32) and
 *   num_batches_tracked (int64, may be NULL) of each BatchNorm in module order.
 * packed: forward and data-gradient weight images of the ten (twelve) convs, sr_vgg_pack_* — repack after every optimiser step.
 *
 * Train-mode BatchNorm does not read its running statistics, but every forward of the reference moves them
 * (nn.BatchNorm2d, momentum 0.1, unbiased variance).  A train forward therefore leaves its update as delta vectors
 * (momentum * batch statistic) in `saved` and applies them once:  running = (1 - momentum) * running + delta, the expression
 * of the fused update, num_batches_tracked += 1.  sr_vgg_apply_stats_* applies the deltas of a kept `saved` block `repeats`
 * more times: the ESRGAN step (esrgan_model.py:38-39,65-72) calls net_d(gt) twice and net_d(output) three times on unchanged
 * weights, the repeats are bit-identical forwards, so a host runs each distinct forward once, keeps its `saved` block for
 * every backward that needs it and replays the statistics of the repeats in the reference's order (host_buffers = NULL in
 * sr_vgg_forward_* skips the update of that call, for a host that orders all of them itself).  Not valid for a
 * spectral-norm network: one power iteration per forward changes the weights themselves.
 *
 * sr_vgg_forward_*:  x NCHW fp32 [n][num_in_ch][S][S] -> logits [n] (may be NULL: they also stay in `saved`);
 *                    train = 0 uses (and does not move) the running statistics.
 * sr_vgg_backward_*: dlogits [n] -> host_dparams (HOST array of DEVICE pointers shaped like the parameters; all NULL or
 *                    host_dparams = NULL for a frozen discriminator; accumulate = 1 adds, which is how the two backward calls of
 *                    the critic phase land in one gradient arena) and, if dx != NULL, dL/dx [n][num_in_ch][S][S].
 *                    `train` = the flag the forward that filled `saved` ran with.
 * The *_bf16 twins keep activations and their gradients as CB16 bf16; parameters, statistics, the linear head and every
 * gradient that leaves the call stay fp32 (num_feat % 16 == 0). */
typedef struct sr_vgg_cfg {
  int num_in_ch, num_feat; /* VGGStyleDiscriminator128.__init__, discriminator_arch.py:18 */
  int input_size;          /* 128 or 256 */
} sr_vgg_cfg;
int sr_vgg_num_params(const sr_vgg_cfg* cfg);
int sr_vgg_num_batchnorm(const sr_vgg_cfg* cfg);
size_t sr_vgg_packed_bytes(const sr_vgg_cfg* cfg);
size_t sr_vgg_saved_bytes(const sr_vgg_cfg* cfg, int n);
size_t sr_vgg_workspace_bytes(const sr_vgg_cfg* cfg, int n);
int sr_vgg_pack_f32(const sr_vgg_cfg* cfg, const float* const* host_params, void* packed, void* stream);
int sr_vgg_forward_f32(const sr_vgg_cfg* cfg, const void* packed, const float* const* host_params, void* const* host_buffers,
                       const float* x, float* logits, int n, int train, void* saved, size_t saved_bytes, void* workspace,
                       size_t workspace_bytes, void* stream);
int sr_vgg_apply_stats_f32(const sr_vgg_cfg* cfg, const void* saved, size_t saved_bytes, int n, void* const* host_buffers,
                           int repeats, void* stream);
int sr_vgg_backward_f32(const sr_vgg_cfg* cfg, const void* packed, const float* const* host_params, const void* saved,
                        size_t saved_bytes, const float* dlogits, int n, int train, float* const* host_dparams, int accumulate,
                        float* dx, void* workspace, size_t workspace_bytes, void* stream);
size_t sr_vgg_packed_bytes_bf16(const sr_vgg_cfg* cfg);
size_t sr_vgg_saved_bytes_bf16(const sr_vgg_cfg* cfg, int n);
size_t sr_vgg_workspace_bytes_bf16(const sr_vgg_cfg* cfg, int n);
int sr_vgg_pack_bf16(const sr_vgg_cfg* cfg, const float* const* host_params, void* packed, void* stream);
int sr_vgg_forward_bf16(const sr_vgg_cfg* cfg, const void* packed, const float* const* host_params, void* const* host_buffers,
                        const float* x, float* logits, int n, int train, void* saved, size_t saved_bytes, void* workspace,
                        size_t workspace_bytes, void* stream);
int sr_vgg_apply_stats_bf16(const sr_vgg_cfg* cfg, const void* saved, size_t saved_bytes, int n, void* const* host_buffers,
                            int repeats, void* stream);
int sr_vgg_backward_bf16(const sr_vgg_cfg* cfg, const void* packed, const float* const* host_params, const void* saved,
                         size_t saved_bytes, const float* dlogits, int n, int train, float* const* host_dparams, int accumulate,
                         float* dx, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------- whole U-Net discriminator (bf16) ---- */
/* UNetDiscriminatorSN with compute_dtype = 'bf16' (the discriminator BASELINE configs 3-4 name; absent from the reference, SURVEY.md
 * section 0 D2: the published architecture) as whole-network drivers like the VGG ones above: one call issues every launch of the forward /
 * of autograd's backward with the descriptors of the per-layer host path, in its order — bit-identical to it.
 * Spectral normalisation stays with the host: host_params are the EFFECTIVE weights of this forward, fp32 device tensors in the order
 * conv0.weight, conv0.bias, conv1 .. conv8 (normalised weights, 4x4 for conv1-3, 3x3 for conv4-8), conv9.weight, conv9.bias (12);
 * host_dparams receives the gradients wrt those (all NULL / host_dparams NULL: a frozen discriminator), written, not accumulated.
 * One power iteration per train-mode forward changes the effective weights: `packed` and `saved` belong to ONE forward.
 * x [n][num_in_ch][h][w] fp32 NCHW (h, w multiples of 8) -> logits [n][1][h][w] fp32; num_feat % 16 == 0. */
typedef struct sr_unet_cfg {
  int num_in_ch, num_feat, skip_connection;
} sr_unet_cfg;
int sr_unet_num_params(const sr_unet_cfg* cfg);
size_t sr_unet_packed_bytes_bf16(const sr_unet_cfg* cfg);
size_t sr_unet_saved_bytes_bf16(const sr_unet_cfg* cfg, int n, int h, int w);
size_t sr_unet_workspace_bytes_bf16(const sr_unet_cfg* cfg, int n, int h, int w);
int sr_unet_pack_bf16(const sr_unet_cfg* cfg, const float* const* host_params, void* packed, void* stream);
int sr_unet_forward_bf16(const sr_unet_cfg* cfg, const void* packed, const float* x, float* logits, int n, int h, int w, void* saved,
                         size_t saved_bytes, void* stream);
int sr_unet_backward_bf16(const sr_unet_cfg* cfg, const void* packed, const void* saved, size_t saved_bytes, const float* dlogits, int n,
                          int h, int w, float* const* host_dparams, float* dx, void* workspace, size_t workspace_bytes, void* stream);

/* Weight gradients of the generator under the discriminator phase (bf16 backward).  The generator's weight gradients feed only its
 * optimiser step, and the discriminator phase of an ESRGAN step (esrgan_model.py:51-73) reads self.output and D's weights only, so
 * nothing between G's backward and G's optimiser step needs them.  With sr_set_backward_wgrad_deferred(1), sr_rrdbnet_backward_bf16
 * issues them on its second lane and RETURNS WITHOUT WAITING for the lane (its workspace grows: one gradient concat buffer per
 * dense block instead of a ring of four — query sr_rrdbnet_backward_workspace_bytes_bf16 with the switch in the state the call
 * will see); dx and the caller's stream are complete as usual.  The caller then owes three things: `saved` and `workspace` stay
 * untouched, and no other backward of this network is issued, until sr_backward_lane_join(stream) has been called — it makes
 * `stream` wait for the pending lane work (stream-ordered, no host wait) — and the parameter gradients are consumed on `stream`
 * after that call only.  Same kernels, same order per weight gradient: results are bit-identical to the undeferred call. */
int sr_set_backward_wgrad_deferred(int on);
int sr_backward_lane_join(void* stream);

/* ------------------------------------------------------------ measurement ---- */

/* Opt-in per-launch timing used by bench.py's roofline line: between sr_profile_start and
 * sr_profile_stop every conv / wgrad launch issued by the process (any thread: autograd runs
 * backward on its own) is bracketed by HIP events recorded on the launch's own stream.
 * sr_profile_stop synchronises on those events only.  One recording session at a time. */
typedef struct sr_launch_record {
  int32_t kernel_id;          /* index for sr_kernel_name */
  int32_t cin, cout, n, h, w; /* real channels, batch, OUTPUT spatial size */
  double flops;               /* algorithmic: 2*9*cin*cout*n*h*w */
  double bytes;               /* algorithmic HBM bytes: source read once + destination written once
                                 + residual / accumulate / mask reads */
  float ms;                   /* event-to-event duration of this launch */
} sr_launch_record;
int sr_profile_start(int max_records);
int sr_profile_stop(sr_launch_record* out, int capacity, int* count);
/* Device symbol substring of a kernel id (matches the rocprofv3 kernel-trace name). */
const char* sr_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* SR_HIP_H */
