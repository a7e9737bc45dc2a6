"""ctypes binding of liburesnet_hip.so (the C ABI declared in include/uresnet_hip.h).

The library is the product path: if it is missing, loading fails loudly -- there is no
CPU or ATen fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("URESNET_HIP_LIB") or os.path.join(_HERE, "lib", "liburesnet_hip.so")   # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "uresnet_hip.h")

UR_F32, UR_BF16 = 0, 1
PACK_FWD, PACK_DGRAD, PACK_CONVT_FWD, PACK_CONVT_DGRAD = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_LEAKYRELU, ACT_GELU, ACT_HSWISH = range(6)


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dtype", "n", "h", "w", "cin", "oh", "ow", "cout", "kh", "kw", "stride",
                                       "pad", "in_ld", "out_ld")]


class UResNetHipError(RuntimeError):
    pass


_vp, _i, _ll, _f, _sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t
_dp = C.POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol the header declares
SIGNATURES = {
    "ur_version": (_i, []),
    "ur_last_error": (C.c_char_p, []),
    "ur_conv2d_packed_elems": (_sz, [_dp, _i]),
    "ur_conv2d_pack_weight": (_i, [_vp, _vp, _dp, _i, _vp]),
    "ur_conv2d_stats_rows": (_i, [_dp]),
    "ur_conv2d_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _dp, _vp]),
    "ur_conv2d_dgrad": (_i, [_vp, _vp, _vp, _dp, _vp]),
    "ur_conv1x1_pw_supported": (_i, [_dp, _i]),
    "ur_conv1x1_pw_fwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _dp, _vp]),
    "ur_conv1x1_pw_dgrad": (_i, [_vp, _vp, _i, _i, _vp, _dp, _vp]),
    "ur_conv1x1_pw_wgrad_supported": (_i, [_dp]),
    "ur_conv1x1_pw_wgrad_workspace": (_sz, [_dp]),
    "ur_conv1x1_pw_wgrad": (_i, [_vp, _vp, _vp, _vp, _sz, _dp, _vp]),
    "ur_conv1x1_pw_fwd_affine": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _dp, _vp]),
    "ur_conv1x1_pw_wgrad_bn_workspace": (_sz, [_dp]),
    "ur_conv1x1_pw_wgrad_bn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _dp, _vp]),
    "ur_conv1x1_pw_dgrad_bn_pair": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _dp, _vp]),
    "ur_conv1x1_pw_wgrad_bn_pair": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp,
                                         _vp, _sz, _dp, _vp]),
    "ur_conv1x1_pw_wgrad_affine": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _sz, _dp, _vp]),
    "ur_conv1x1_pw_fwd_affine_pair": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _dp, _vp]),
    "ur_conv1x1_pw_dgrad_pair": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _dp, _vp]),
    "ur_conv1x1_pw_wgrad_affine_pair": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _dp, _vp]),
    "ur_conv2d_wgrad_workspace": (_sz, [_dp]),
    "ur_conv2d_wgrad": (_i, [_vp, _vp, _vp, _vp, _sz, _dp, _vp]),
    "ur_conv3x3_tile_supported": (_i, [_dp, _i]),
    "ur_conv3x3_tile_packed_elems": (_sz, [_dp]),
    "ur_conv3x3_tile_pack_weight": (_i, [_vp, _vp, _dp, _i, _vp]),
    "ur_conv3x3_tile_pack_weights_multi": (_i, [_vp, _i, _ll, _vp]),
    "ur_conv3x3_tile_pack_blocks": (_ll, [_i, _i, _i]),
    "ur_conv3x3_tile_stats_rows": (_i, [_dp]),
    "ur_conv3x3_tile_fwd": (_i, [_vp, _vp, _vp, _vp, _dp, _vp]),
    "ur_conv3x3_tile_dgrad": (_i, [_vp, _vp, _vp, _dp, _vp]),
    "ur_conv3x3_tile_dgrad_add": (_i, [_vp, _vp, _vp, _vp, _i, _dp, _vp]),
    "ur_conv3x3_tile_dgrad_stats_rows": (_i, [_dp]),
    "ur_conv3x3_wide_supported": (_i, [_dp, _i]),
    "ur_conv3x3_wide_preferred": (_i, [_dp, _i]),
    "ur_conv3x3_wide_packed_elems": (_sz, [_dp]),
    "ur_conv3x3_wide_pack_weight": (_i, [_vp, _vp, _dp, _i, _vp]),
    "ur_conv3x3_wide_stats_rows": (_i, [_dp]),
    "ur_conv3x3_wide_fwd": (_i, [_vp, _vp, _vp, _vp, _dp, _vp]),
    "ur_conv3x3_wide_dgrad": (_i, [_vp, _vp, _vp, _dp, _vp]),
    "ur_conv3x3_wide_dgrad_add": (_i, [_vp, _vp, _vp, _vp, _i, _dp, _vp]),
    "ur_conv3x3_wide_dgrad_bnsums": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _dp, _vp]),
    "ur_conv3x3_tile_dgrad_bnsums": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _dp, _vp]),
    "ur_conv3x3_tile_fwd_affine": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _dp, _vp]),
    "ur_wgrad3x3_tile_supported": (_i, [_dp]),
    "ur_wgrad3x3_tile_workspace": (_sz, [_dp]),
    "ur_wgrad3x3_tile": (_i, [_vp, _vp, _vp, _vp, _sz, _dp, _vp]),
    "ur_wgrad3x3_tile_marked": (_i, [_vp, _vp, _vp, _vp, _sz, _dp, _vp, _vp]),
    "ur_convT2x2_fwd": (_i, [_vp, _vp, _vp, _vp, _dp, _vp]),
    "ur_convT2x2_dgrad": (_i, [_vp, _vp, _vp, _dp, _vp]),
    "ur_convT2x2_wgrad_workspace": (_sz, [_dp]),
    "ur_convT2x2_wgrad": (_i, [_vp, _vp, _vp, _vp, _sz, _dp, _vp]),
    "ur_colsum_workspace": (_sz, [_ll, _i]),
    "ur_colsum": (_i, [_vp, _i, _ll, _i, _i, _vp, _vp, _sz, _vp]),
    "ur_gconv3x3_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ur_gconv3x3_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ur_gconv3x3_stats_rows": (_i, [_i, _i, _i, _i, _i, _i]),
    "ur_gconv3x3_fwd_stats": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ur_gconv3x3_wgrad_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "ur_gconv3x3_wgrad": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp]),
    "ur_bn_stats_rows": (_sz, [_ll, _i]),
    "ur_bn_stats": (_i, [_vp, _i, _ll, _i, _vp, _vp]),
    "ur_bn_finalize": (_i, [_vp, _i, _ll, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ur_bn_finalize_affine": (_i, [_vp, _i, _ll, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ur_bn_eval_prepare": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp]),
    "ur_bn_eval_affine": (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp]),
    "ur_bn_act_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _ll, _i, _i, _vp]),
    "ur_bn_act_bwd_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _i, _i, _vp, _vp]),
    "ur_maxpool2x2_bwd_bnsums_supported": (_i, [_i, _i, _i, _i, _i]),
    "ur_maxpool2x2_bwd_bnsums": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ur_bn_bwd_finalize": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "ur_bn_act_fwd_pool_supported": (_i, [_i, _i, _i, _i, _i]),
    "ur_bn_act_fwd_pool": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ur_bn_act_fwd_stats_supported": (_i, [_i, _ll, _i]),
    "ur_bn_act_fwd_stats_rows": (_sz, [_ll, _i]),
    "ur_bn_act_fwd_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _ll, _i, _vp, _vp]),
    "ur_bn_act_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _ll, _i, _i, _vp]),
    "ur_maxpool2x2_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ur_maxpool2x2_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ur_upsample_bilinear_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ur_upsample_bilinear_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ur_avgpool_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ur_avgpool_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ur_copy_channels": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ur_pad_channels": (_i, [_vp, _i, _vp, _i, _ll, _i, _vp]),
    "ur_channel_shuffle": (_i, [_vp, _vp, _i, _ll, _i, _i, _vp]),
    "ur_act_fwd": (_i, [_vp, _vp, _i, _ll, _i, _f, _vp]),
    "ur_act_bwd": (_i, [_vp, _vp, _vp, _i, _ll, _i, _f, _vp]),
    "ur_prelu_fwd": (_i, [_vp, _vp, _i, _vp, _i, _ll, _i, _vp]),
    "ur_prelu_bwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _ll, _i, _vp]),
    "ur_add_inplace": (_i, [_vp, _vp, _i, _ll, _vp]),
    "ur_spin": (_i, [_i, _vp]),
    "ur_nchw_to_nhwc": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "ur_pack_image_u8": (_i, [_vp, _vp, _i, _ll, _i, _i, _vp, _vp, _i, _vp]),
    "ur_remap_labels_u8": (_i, [_vp, _vp, _ll, _vp, _vp]),
    "ur_nhwc_to_nchw": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "ur_loss_workspace": (_sz, [_ll]),
    "ur_ce_fwd": (_i, [_vp, _i, _vp, _ll, _i, _i, _ll, _vp, _vp, _vp, _sz, _vp]),
    "ur_ce_fwd_noalloc_max_rows": (_ll, []),
    "ur_ce_fwd_noalloc_prepare": (_i, []),
    "ur_ce_fwd_noalloc": (_i, [_vp, _i, _vp, _ll, _i, _i, _ll, _vp, _vp, _vp]),
    "ur_ce_bwd": (_i, [_vp, _i, _vp, _ll, _i, _i, _ll, _vp, _vp, _vp, _vp, _i, _vp]),
    "ur_ce_bwd_padded_workspace": (_sz, [_ll]),
    "ur_ce_bwd_padded_supported": (_i, [_i, _i, _i]),
    "ur_ce_bwd_padded": (_i, [_vp, _i, _vp, _ll, _i, _ll, _vp, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "ur_bce_fwd": (_i, [_vp, _i, _vp, _ll, _vp, _vp, _sz, _vp]),
    "ur_bce_bwd": (_i, [_vp, _i, _vp, _ll, _vp, _vp, _vp]),
    "ur_argmax_channels": (_i, [_vp, _i, _ll, _i, _i, _vp, _vp]),
    "ur_sigmoid_mask": (_i, [_vp, _i, _ll, _vp, _vp]),
    "ur_confusion_matrix": (_i, [_vp, _i, _vp, _ll, _i, _i, _ll, _vp, _vp]),
    "ur_seg_metrics_accumulate": (_i, [_vp, _i, _vp, _vp, _vp]),
    "ur_adam_multi": (_i, [_vp, _vp, _i, _ll, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "ur_adam_multi_dev": (_i, [_vp, _vp, _i, _ll, _f, _f, _f, _f, _f, _f, _vp]),
    "ur_sgd_multi": (_i, [_vp, _vp, _i, _ll, _f, _f, _f, _vp]),
}

_lib = None


def load():
    """Load the shared library once; raises UResNetHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UResNetHipError(
            "liburesnet_hip.so not found at %s -- build it with `python __graft_entry__.py build` "
            "(u-resnet_amd/csrc/build.sh); there is no fallback path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = library/header mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise UResNetHipError("%s failed (%d): %s" % (name, rc, lib.ur_last_error().decode()))


def query(name, *args):
    """Invoke a size/count-returning entry point."""
    return getattr(load(), name)(*args)
