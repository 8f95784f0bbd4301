"""ctypes binding of libdmcnet_hip.so (the C ABI declared in include/dmcnet_hip.h).

There is no CPU fallback: if the shared library is missing or an entry point fails, the
caller gets an exception.
"""
import ctypes
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMC_HIP_LIB", os.path.join(PKG, "libdmcnet_hip.so"))   # override: A/B builds

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_Z = ctypes.c_size_t
_L = ctypes.c_long

#: name -> (restype, argtypes); must list every symbol include/dmcnet_hip.h declares
SIGNATURES = {
    "dmc_version": (_I, []),
    "dmc_last_error": (ctypes.c_char_p, []),
    "dmc_profile_mark": (_I, [_P]),
    "dmc_set_option": (_I, [ctypes.c_char_p, _I]),
    "dmc_get_option": (_I, [ctypes.c_char_p]),
    "dmc_gen_tiny_workspace_bytes": (_Z, []),
    "dmc_gen_tiny_saved_bytes": (_Z, [_I, _I, _I]),
    "dmc_gen_tiny_gbuf_bytes": (_Z, [_I, _I, _I]),
    "dmc_gen_tiny_partials_bytes": (_Z, [_I, _I, _I]),
    "dmc_gen_tiny_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dmc_gen_tiny_mse_partials_bytes": (_Z, []),
    "dmc_gen_tiny_fwd_mse": (_I, [_P] * 10 + [_I, _I, _I, _I, _P]),
    "dmc_gen_tiny_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dmc_flow_mse_partials_bytes": (_Z, []),
    "dmc_flow_mse_fwd": (_I, [_P, _P, _P, _P, _Z, _P]),
    "dmc_flow_mse_bwd": (_I, [_P, _P, _P, _P, _Z, _P]),
    "dmc_consensus_ce_fwd_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dmc_disc_tail_stats_bytes": (_Z, [_I]),
    "dmc_disc_tail_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "dmc_disc_tail_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dmc_prepare_inputs_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "dmc_prepare_inputs": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "dmc_prepare_crop_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "dmc_prepare_inputs_crop": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dmc_bn_act_supported": (_I, [_I, _I]),
    "dmc_bn_act_stats_bytes": (_Z, [_I]),
    "dmc_bn_act_scratch_bytes": (_Z, [_I]),
    "dmc_bn_act_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P]),
    "dmc_bn_act_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dmc_bn_apply_nhwc": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "dmc_bn_apply_act_nhwc": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dmc_bn_bwd_act_nhwc": (_I, [_P] * 10 + [_I, _F, _I, _I, _P]),
    "dmc_channel_sum_nhwc": (_I, [_P, _P, _P, _I, _I, _P]),
    "dmc_bn_relu_pool_supported": (_I, [_I, _I, _I, _I]),
    "dmc_bn_relu_pool_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P]),
    "dmc_bn_relu_pool_codes_bytes": (_Z, [_I, _I, _I, _I]),
    "dmc_bn_relu_pool_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "dmc_conv_nhwc_supported": (_I, [_I] * 9),
    "dmc_conv_nhwc_stat_blocks": (_I, [_I] * 8),
    "dmc_conv_nhwc_fwd": (_I, [_P] * 7 + [_I] * 11 + [_P]),
    "dmc_conv_nhwc_stats_final": (_I, [_P, _I, _I, ctypes.c_long, _P, _P, _P, _F, _F, _P]),
    "dmc_conv_nhwc_wt_bytes": (_Z, [_I] * 4),
    "dmc_conv_nhwc_presplit_supported": (_I, [_I, _I]),
    "dmc_conv_nhwc_split": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dmc_conv_nhwc_dgrad": (_I, [_P] * 4 + [_I] * 9 + [_P]),
    "dmc_conv_nhwc_dgrad_add": (_I, [_P] * 5 + [_I] * 9 + [_P]),
    "dmc_conv_nhwc_wgrad_bytes": (_Z, [_I] * 9),
    "dmc_conv_nhwc_wgrad": (_I, [_P] * 4 + [_I] * 9 + [_P]),
    "dmc_x3s_slices_bytes": (_Z, [_L, _I]),
    "dmc_x3s_split": (_I, [_P, _P, _L, _I, _P]),
    "dmc_x3s_merge": (_I, [_P, _P, _L, _I, _P]),
    "dmc_x3s_wpack_bytes": (_Z, [_I, _I]),
    "dmc_x3s_pack_weights": (_I, [_P, _P, _P, _I, _I, _P]),
    "dmc_x3s_conv_supported": (_I, [_I] * 5),
    "dmc_x3s_conv_stat_blocks": (_I, [_I] * 4),
    "dmc_x3s_conv_fwd": (_I, [_P] * 4 + [_I] * 6 + [_P]),
    "dmc_x3s_conv_dgrad": (_I, [_P] * 4 + [_I] * 5 + [_P]),
    "dmc_x3s_conv_wgrad_supported": (_I, [_I] * 5),
    "dmc_x3s_conv_wgrad_bytes": (_Z, [_I] * 5),
    "dmc_x3s_conv_wgrad": (_I, [_P] * 4 + [_I] * 5 + [_P]),
    "dmc_x3s_conv_dgrad_s2_supported": (_I, [_I] * 5),
    "dmc_x3s_pack_weights_s2": (_I, [_P, _P, _I, _I, _P]),
    "dmc_x3s_conv_dgrad_s2": (_I, [_P] * 3 + [_I] * 5 + [_P]),
    "dmc_x3q_wpack_bytes": (_Z, [_I, _I]),
    "dmc_x3q_supported": (_I, [_I] * 5),
    "dmc_x3q_stat_blocks": (_I, [_I] * 4),
    "dmc_x3q_split": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dmc_x3q_merge": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dmc_x3q_pack_weights": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "dmc_x3q_conv_fwd": (_I, [_P] * 6 + [_I] * 6 + [_P]),
    "dmc_x3q_conv_dgrad": (_I, [_P] * 4 + [_I] * 5 + [_P]),
    "dmc_bn_apply_act_x3q": (_I, [_P] * 8 + [_I] * 5 + [_P]),
    "dmc_x3q_conv_wgrad_supported": (_I, [_I] * 5),
    "dmc_x3q_conv_wgrad_bytes": (_Z, [_I] * 5),
    "dmc_x3q_conv_wgrad": (_I, [_P] * 6 + [_I] * 5 + [_P]),
    "dmc_bn_apply_act_x3s": (_I, [_P] * 8 + [_I, _I, _I, _P]),
    "dmc_bn_act_bwd_x3s": (_I, [_P] * 13 + [_I, _I, _I, _P]),
    "dmc_x3s_conv_dgrad_bnb": (_I, [_P] * 9 + [_I] + [_P] + [_I] + [_P] * 2 + [_I] * 5 + [_P]),
    "dmc_bn_act_bwd_x3s_apply": (_I, [_P] * 12 + [_I, _I, _I, _P]),
    "dmc_bn_relu_pool_fwd_x3s": (_I, [_P] * 9 + [_I, _I, _I, _I, _I, _F, _F, _P]),
    "dmc_bn_relu_pool_fwd_arg": (_I, [_P] * 11 + [_I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "dmc_bn_relu_pool_bwd_arg": (_I, [_P] * 11 + [_I, _I, _I, _I, _P]),
    "dmc_disc_first_supported": (_I, [_I]),
    "dmc_disc_first_fwd": (_I, [_P] * 5 + [_I] * 5 + [_P]),
    "dmc_disc_first_dgrad": (_I, [_P] * 3 + [_I] * 4 + [_P]),
    "dmc_disc_first_wgrad_bytes": (_Z, [_I]),
    "dmc_disc_first_wgrad": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "dmc_stem_wgrad_supported": (_I, [_I, _I]),
    "dmc_stem_wgrad_partials_bytes": (_Z, [_I, _I, _I]),
    "dmc_stem_wgrad": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "dmc_stem_fwd": (_I, [_P, _P, _L, _L, _L, _L, _P, _I, _I, _I, _P]),
    "dmc_stem_fwd_x3_workspace_bytes": (_Z, [_I, _I, _I]),
    "dmc_stem_fwd_x3": (_I, [_P, _P, _L, _L, _L, _L, _P, _P, _I, _I, _I, _P]),
    "dmc_stem_fwd_x3_stat_blocks": (_I, [_I, _I, _I]),
    "dmc_stem_dgrad_workspace_bytes": (_Z, []),
    "dmc_stem_dgrad_supported": (_I, [_I, _I]),
    "dmc_stem_dgrad": (_I, [_P, _P, _L, _L, _L, _L, _P, _P, _I, _I, _I, _P]),
    "dmc_stem_fwd_x3_stats": (_I, [_P, _P, _L, _L, _L, _L, _P, _P, _P, _I, _I, _I, _P]),
    "dmc_conv3d_bf16_supported": (_I, [_I] * 9),
    "dmc_conv3d_bf16_wpack_bytes": (_Z, [_I] * 5),
    "dmc_conv3d_bf16_pack": (_I, [_P, _L, _L, _L, _P, _P] + [_I] * 5 + [_P]),
    "dmc_conv3d_bf16_stat_blocks": (_I, [_I] * 5),
    "dmc_conv3d_bf16_stat_blocks_k": (_I, [_I] * 9),
    "dmc_conv3d_bf16_fwd": (_I, [_P, _P, _L, _L, _L, _P, _P, _P] + [_I] * 9 + [_P]),
    "dmc_conv3d_bf16_dgrad": (_I, [_P, _P, _L, _L, _L, _P, _P] + [_I] * 9 + [_P]),
    "dmc_conv3d_bf16_wgrad_bytes": (_Z, [_I] * 9),
    "dmc_conv3d_bf16_wgrad": (_I, [_P] * 4 + [_I] * 9 + [_P]),
    "dmc_stem3d_bf16_workspace_bytes": (_Z, [_I] * 4),
    "dmc_stem3d_bf16_stat_blocks": (_I, [_I] * 4),
    "dmc_stem3d_bf16_fwd": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "dmc_stem3d_bf16_wgrad_workspace_bytes": (_Z, [_I] * 4),
    "dmc_stem3d_bf16_wgrad": (_I, [_P] * 4 + [_I] * 4 + [_P]),
    "dmc_stem3d_bf16_dgrad_workspace_bytes": (_Z, []),
    "dmc_stem3d_bf16_dgrad": (_I, [_P] * 4 + [_I] * 4 + [_P]),
    "dmc_bn3d_bf16_supported": (_I, [_L, _I]),
    "dmc_bn3d_bf16_scratch_bytes": (_Z, [_I]),
    "dmc_bn3d_bf16_fwd": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _F, _P]),
    "dmc_add4_bf16": (_I, [_P] * 5 + [_L, _P]),
    "dmc_bn3d_bf16_fwd_ld": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _F, _F, _P]),
    "dmc_bn3d_bf16_bwd": (_I, [_P, _L] + [_P] * 8 + [_L, _I, _I, _P]),
    "dmc_unit3d_bf16_fwd_workspace_bytes": (_Z, [_I] * 9),
    "dmc_unit3d_bf16_bwd_workspace_bytes": (_Z, [_I] * 9),
    "dmc_unit3d_bf16_fwd": (_I, [_P] * 9 + [_I] * 10 + [_F, _F, _P]),
    "dmc_unit3d_bf16_fwd_into": (_I, [_P] * 9 + [_L] + [_I] * 10 + [_F, _F, _P]),
    "dmc_unit3d_bf16_bwd": (_I, [_P, _L] + [_P] * 11 + [_I] * 10 + [_P]),
    "dmc_maxpool3d_tf_out_shape": (_I, [_I] * 10 + [_P] * 3),
    "dmc_maxpool3d_tf_bf16_fwd": (_I, [_P] * 3 + [_I] * 11 + [_P]),
    "dmc_maxpool3d_tf_bf16_bwd": (_I, [_P] * 3 + [_I] * 11 + [_P]),
    "dmc_mv_owner_bytes": (_Z, [_I, _I, _I]),
    "dmc_mv_accu_init": (_I, [_P, _I, _I, _P]),
    "dmc_mv_rasterise": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _P]),
    "dmc_mv_accumulate": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _I, _P]),
    "dmc_mv_from_accu": (_I, [_P, _P, _I, _I, _P]),
    "dmc_residual": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "dmc_mv_gop_batch": (_I, [_P, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
}

_lib = None


class DmcHipError(RuntimeError):
    pass


def load():
    """Load the library once; raises if it has not been built (``python -m dmcnet_amd.build``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DmcHipError(
                "%s is missing: the DMC-Net hot path has no CPU fallback. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'`." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)     # AttributeError if the .so does not export it
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise DmcHipError("%s failed (%d): %s" % (what, rc, load().dmc_last_error().decode()))


def ptr(t):
    return _P(t.data_ptr()) if t is not None else _P(0)


def ptr_array(tensors):
    return (_P * len(tensors))(*[t.data_ptr() for t in tensors])
