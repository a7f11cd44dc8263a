"""ctypes binding of libsr_hip.so (C ABI declared in include/sr_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails the
caller gets an exception.  The product path never routes through ``oracle/`` or
through PyTorch CPU/ATen convolutions.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SR_HIP_LIB_PATH') or os.path.join(_HERE, 'lib', 'libsr_hip.so')  # override: A/B builds of the kernels

SR_ABI_VERSION = 3


class SrHipError(RuntimeError):
    """Raised when a libsr_hip.so entry point returns a non-zero status."""


class ConvDesc(C.Structure):
    """struct sr_conv3x3_desc (include/sr_hip.h)."""
    _fields_ = [
        ('in_', C.c_void_p), ('in_img_stride', C.c_int64), ('cin_pad', C.c_int), ('cin_real', C.c_int), ('in_h', C.c_int), ('in_w', C.c_int),
        ('upsample', C.c_int), ('wpacked', C.c_void_p), ('bpacked', C.c_void_p), ('cout', C.c_int),
        ('out', C.c_void_p), ('out_img_stride', C.c_int64), ('out_nchw', C.c_int), ('n', C.c_int),
        ('act_slope', C.c_float), ('alpha', C.c_float),
        ('res1', C.c_void_p), ('res1_img_stride', C.c_int64), ('beta1', C.c_float),
        ('res2', C.c_void_p), ('res2_img_stride', C.c_int64), ('beta2', C.c_float), ('res_cbn', C.c_int),
        ('out_h', C.c_int), ('out_w', C.c_int), ('accumulate', C.c_int), ('mask_src', C.c_void_p), ('mask_img_stride', C.c_int64),
        ('mask_cb0', C.c_int), ('mask_cbn', C.c_int), ('mask_slope', C.c_float), ('s2_channels', C.c_int), ('s2_side', C.c_int),
        ('out_unshuffle2', C.c_int), ('res1_u2', C.c_int), ('res1_keep_sign', C.c_int),
    ]


class WgradDesc(C.Structure):
    """struct sr_conv3x3_wgrad_desc (include/sr_hip.h)."""
    _fields_ = [
        ('x', C.c_void_p), ('x_img_stride', C.c_int64), ('cin_pad', C.c_int), ('in_h', C.c_int), ('in_w', C.c_int),
        ('upsample', C.c_int), ('dy', C.c_void_p), ('dy_img_stride', C.c_int64), ('cout', C.c_int), ('cin', C.c_int),
        ('first_seg', C.c_int), ('seg', C.c_int), ('n', C.c_int), ('scale', C.c_float), ('dweight', C.c_void_p),
        ('dbias', C.c_void_p), ('accumulate', C.c_int), ('slab', C.c_void_p), ('slab_bytes', C.c_size_t),
    ]


class SnLayer(C.Structure):
    """struct sr_sn_layer (include/sr_hip.h)."""
    _fields_ = [('w_orig', C.c_void_p), ('u', C.c_void_p), ('v', C.c_void_p), ('rows', C.c_int), ('cols', C.c_int),
                ('w_sn', C.c_void_p), ('sigma', C.c_void_p)]


class RRDBNetCfg(C.Structure):
    """struct sr_rrdbnet_cfg (include/sr_hip.h)."""
    _fields_ = [('num_in_ch', C.c_int), ('num_out_ch', C.c_int), ('scale', C.c_int), ('num_feat', C.c_int),
                ('num_block', C.c_int), ('num_grow_ch', C.c_int)]


class VGGCfg(C.Structure):
    """struct sr_vgg_cfg (include/sr_hip.h)."""
    _fields_ = [('num_in_ch', C.c_int), ('num_feat', C.c_int), ('input_size', C.c_int)]


class UNetCfg(C.Structure):
    """struct sr_unet_cfg (include/sr_hip.h)."""
    _fields_ = [('num_in_ch', C.c_int), ('num_feat', C.c_int), ('skip_connection', C.c_int)]


# name -> (restype, argtypes); every symbol include/sr_hip.h declares
SIGNATURES = {
    'sr_version': (C.c_int, []),
    'sr_last_error': (C.c_char_p, []),
    'sr_nchw_to_cb8_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int64, C.c_void_p]),
    'sr_cb8_to_nchw_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p]),
    'sr_upsample2x_bwd_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_float,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_cb8_axpby_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_conv3x3_packed_weight_floats': (C.c_size_t, [C.c_int, C.c_int]),
    'sr_conv3x3_packed_bias_floats': (C.c_size_t, [C.c_int]),
    'sr_conv3x3_cin_pad': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'sr_conv3x3_pack_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    'sr_conv3x3_f32': (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    'sr_conv4x4s2_packed_weight_floats': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'sr_conv4x4s2_pack_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'sr_conv4x4s2_f32': (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    'sr_conv4x4s2_dgrad_f32': (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    'sr_conv3x3_wgrad_slab_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'sr_conv3x3_wgrad_f32': (C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    'sr_conv4x4s2_wgrad_f32': (C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    'sr_reduce_workspace_bytes': (C.c_size_t, [C.c_int]),
    'sr_bn_lrelu_fwd_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_bn_lrelu_bwd_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_lrelu_bwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    'sr_linear_fwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_float, C.c_void_p]),
    'sr_linear_bwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'sr_bilinear2x_fwd_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p]),
    'sr_bilinear2x_bwd_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p]),
    'sr_spectral_norm_fwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_spectral_norm_bwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_spectral_norm_fwd_batch_f32': (C.c_int, [C.POINTER(SnLayer), C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_add_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'sr_gram_fwd_f32': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    'sr_gram_bwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    'sr_psnr_sse_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    'sr_maxpool2x2_fwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_maxpool2x2_bwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_channel_affine_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    'sr_lrelu_fwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    'sr_ssim_sum_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    'sr_mean_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_l1_loss_fwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    'sr_l1_loss_bwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'sr_pixel_loss_fwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p]),
    'sr_gan_point_loss_fwd_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.c_void_p]),
    'sr_gan_point_loss_bwd_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'sr_pixel_loss_bwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    'sr_bce_logits_fwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_bce_logits_bwd_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    'sr_fill_scaled_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    'sr_adam_step_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float,
                                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'sr_axpby_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int64, C.c_void_p, C.c_void_p]),
    'sr_abort_latch': (C.c_void_p, []),
    'sr_set_backward_wgrad_deferred': (C.c_int, [C.c_int]),
    'sr_backward_lane_join': (C.c_int, [C.c_void_p]),
    'sr_abort_latch_clear': (C.c_int, [C.c_void_p]),
    'sr_rrdbnet_num_params': (C.c_int, [C.POINTER(RRDBNetCfg)]),
    'sr_rrdbnet_packed_bytes': (C.c_size_t, [C.POINTER(RRDBNetCfg)]),
    'sr_rrdbnet_workspace_bytes': (C.c_size_t, [C.POINTER(RRDBNetCfg), C.c_int, C.c_int, C.c_int]),
    'sr_rrdbnet_pack_f32': (C.c_int, [C.POINTER(RRDBNetCfg), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    'sr_rrdbnet_forward_f32': (C.c_int, [C.POINTER(RRDBNetCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_nchw_to_cb16_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                       C.c_void_p]),
    'sr_cb16_to_nchw_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    'sr_conv3x3_cin_pad16': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'sr_conv3x3_packed_weight_elems_bf16': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'sr_conv3x3_pack_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    'sr_conv3x3_bf16': (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    'sr_conv3x3_wgrad_slab_bytes_bf16': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'sr_conv3x3_wgrad_bf16': (C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    'sr_rdb_wgrad_slab_bytes_bf16': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'sr_rdb_wgrad_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_void_p), C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_upsample2x_bwd_bf16': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_float,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_cb16_axpby_bf16': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p]),
    'sr_cb16_unshuffle2_bf16': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p]),
    'sr_conv4x4s2_weight_as_3x3_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_lrelu_bwd_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    'sr_bilinear2x_fwd_bf16': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_void_p]),
    'sr_cb16_add_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'sr_lrelu_fwd_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    'sr_maxpool2x2_fwd_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_maxpool2x2_bwd_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_cb16_fork_bwd_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p]),
    'sr_bilinear2x_bwd_bf16': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p]),
    'sr_bilinear2x_bwd_lrelu_bf16': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_float, C.c_void_p,
                                               C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_cb16_add_u2_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'sr_bilinear2x_fwd_u2_bf16': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p]),
    'sr_cb16_fork_bwd_u2_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p]),
    'sr_lrelu_bwd_diff_u2_bf16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_void_p]),
    'sr_bn_lrelu_fwd_bf16': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_bn_lrelu_bwd_bf16': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_rrdbnet_packed_bytes_bf16': (C.c_size_t, [C.POINTER(RRDBNetCfg)]),
    'sr_rrdbnet_workspace_bytes_bf16': (C.c_size_t, [C.POINTER(RRDBNetCfg), C.c_int, C.c_int, C.c_int]),
    'sr_rrdbnet_pack_bf16': (C.c_int, [C.POINTER(RRDBNetCfg), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    'sr_rrdbnet_forward_bf16': (C.c_int, [C.POINTER(RRDBNetCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_rrdbnet_saved_bytes_bf16': (C.c_size_t, [C.POINTER(RRDBNetCfg), C.c_int, C.c_int, C.c_int]),
    'sr_rrdbnet_backward_workspace_bytes_bf16': (C.c_size_t, [C.POINTER(RRDBNetCfg), C.c_int, C.c_int, C.c_int]),
    'sr_rrdbnet_packed_dgrad_bytes_bf16': (C.c_size_t, [C.POINTER(RRDBNetCfg)]),
    'sr_rrdbnet_pack_dgrad_bf16': (C.c_int, [C.POINTER(RRDBNetCfg), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    'sr_rrdbnet_forward_train_bf16': (C.c_int, [C.POINTER(RRDBNetCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_rrdbnet_backward_bf16': (C.c_int, [C.POINTER(RRDBNetCfg), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                           C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_int, C.c_void_p]),
    'sr_set_forward_groups': (C.c_int, [C.c_int]),
    'sr_rrdbnet_saved_bytes': (C.c_size_t, [C.POINTER(RRDBNetCfg), C.c_int, C.c_int, C.c_int]),
    'sr_rrdbnet_backward_workspace_bytes': (C.c_size_t, [C.POINTER(RRDBNetCfg), C.c_int, C.c_int, C.c_int]),
    'sr_rrdbnet_packed_dgrad_bytes': (C.c_size_t, [C.POINTER(RRDBNetCfg)]),
    'sr_rrdbnet_pack_dgrad_f32': (C.c_int, [C.POINTER(RRDBNetCfg), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    'sr_rrdbnet_forward_train_f32': (C.c_int, [C.POINTER(RRDBNetCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'sr_rrdbnet_backward_f32': (C.c_int, [C.POINTER(RRDBNetCfg), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_int, C.c_void_p]),
}



class LaunchRecord(C.Structure):
    """struct sr_launch_record (include/sr_hip.h)."""
    _fields_ = [('kernel_id', C.c_int32), ('cin', C.c_int32), ('cout', C.c_int32), ('n', C.c_int32), ('h', C.c_int32),
                ('w', C.c_int32), ('flops', C.c_double), ('bytes', C.c_double), ('ms', C.c_float)]


SIGNATURES.update({
    'sr_conv3x3_chain_sync_ints': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'sr_conv3x3_chain_bf16': (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'sr_set_conv_chain': (C.c_int, [C.c_int]),
    'sr_chain_watchdog': (C.c_int, []),
    'sr_conv3x3_chain_f32': (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'sr_set_conv_chain_f32': (C.c_int, [C.c_int]),
    'sr_patch_augment_u8_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                          C.POINTER(C.c_float), C.c_void_p]),
    'sr_profile_start': (C.c_int, [C.c_int]),
    'sr_profile_stop': (C.c_int, [C.POINTER(LaunchRecord), C.c_int, C.POINTER(C.c_int)]),
    'sr_kernel_name': (C.c_char_p, [C.c_int]),
})

for _suf, _plain in (('_f32', ''), ('_bf16', '_bf16')):
    SIGNATURES.update({
        'sr_vgg_packed_bytes' + _plain: (C.c_size_t, [C.POINTER(VGGCfg)]),
        'sr_vgg_saved_bytes' + _plain: (C.c_size_t, [C.POINTER(VGGCfg), C.c_int]),
        'sr_vgg_workspace_bytes' + _plain: (C.c_size_t, [C.POINTER(VGGCfg), C.c_int]),
        'sr_vgg_pack' + _suf: (C.c_int, [C.POINTER(VGGCfg), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
        'sr_vgg_forward' + _suf: (C.c_int, [C.POINTER(VGGCfg), C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p,
                                            C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
        'sr_vgg_apply_stats' + _suf: (C.c_int, [C.POINTER(VGGCfg), C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.c_int,
                                                C.c_void_p]),
        'sr_vgg_backward' + _suf: (C.c_int, [C.POINTER(VGGCfg), C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p,
                                             C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                             C.c_void_p]),
    })
SIGNATURES.update({
    'sr_unet_num_params': (C.c_int, [C.POINTER(UNetCfg)]),
    'sr_unet_packed_bytes_bf16': (C.c_size_t, [C.POINTER(UNetCfg)]),
    'sr_unet_saved_bytes_bf16': (C.c_size_t, [C.POINTER(UNetCfg), C.c_int, C.c_int, C.c_int]),
    'sr_unet_workspace_bytes_bf16': (C.c_size_t, [C.POINTER(UNetCfg), C.c_int, C.c_int, C.c_int]),
    'sr_unet_pack_bf16': (C.c_int, [C.POINTER(UNetCfg), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    'sr_unet_forward_bf16': (C.c_int, [C.POINTER(UNetCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    'sr_unet_backward_bf16': (C.c_int, [C.POINTER(UNetCfg), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
})
SIGNATURES.update({'sr_vgg_num_params': (C.c_int, [C.POINTER(VGGCfg)]), 'sr_vgg_num_batchnorm': (C.c_int, [C.POINTER(VGGCfg)])})

_lib = None


def load():
    """Loads libsr_hip.so (once).  Raises if it is missing or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SrHipError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                         '(or `make -C image_restoration_amd/csrc`). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    v = lib.sr_version()
    if v != SR_ABI_VERSION:
        raise SrHipError(f'libsr_hip.so ABI {v} != expected {SR_ABI_VERSION}: rebuild')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().sr_last_error().decode('utf-8', 'replace')
        raise SrHipError(f'{what} failed (status {rc}): {msg}')
