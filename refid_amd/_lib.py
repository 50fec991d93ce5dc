"""ctypes binding of librefid_hip.so (the C ABI declared in include/refid_hip.h).

There is NO fallback: if the shared library is missing or an entry point fails, the
caller gets an exception.  Build it with ``python -m refid_amd.build``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librefid_hip.so")


class RefidHipError(RuntimeError):
    pass


class PwExtras(C.Structure):
    _fields_ = [
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float), ("ln_out", C.c_void_p),
        ("ld_ln_out", C.c_int),
        ("pool", C.c_void_p), ("pool_parts", C.c_int), ("inv_hw", C.c_float), ("hw", C.c_int), ("se_c", C.c_int),
        ("se_w1", C.c_void_p), ("se_b1", C.c_void_p), ("se_w2", C.c_void_p), ("se_b2", C.c_void_p),
        ("se_m", C.c_void_p), ("se_z1", C.c_void_p), ("se_s", C.c_void_p),
        ("xs_out", C.c_void_p), ("ld_xs_out", C.c_int),
        ("res2", C.c_void_p), ("ld_res2", C.c_int),
        ("out2", C.c_void_p), ("ld_out2", C.c_int),
    ]


class ConvDesc(C.Structure):
    _fields_ = [
        ("in_a", C.c_void_p), ("in_b", C.c_void_p),
        ("ld_a", C.c_int), ("ld_b", C.c_int),
        ("c_a", C.c_int), ("c_b", C.c_int),
        ("w_packed", C.c_void_p), ("bias", C.c_void_p),
        ("out", C.c_void_p), ("ld_out", C.c_int),
        ("res", C.c_void_p), ("ld_res", C.c_int),
        ("mask", C.c_void_p), ("ld_mask", C.c_int),
        ("n", C.c_int), ("h", C.c_int), ("w", C.c_int),
        ("ho", C.c_int), ("wo", C.c_int),
        ("cout", C.c_int), ("cout_pad", C.c_int), ("co_base", C.c_int),
        ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("mode", C.c_int),
        ("slope_pre", C.c_float), ("slope_post", C.c_float), ("slope_mask", C.c_float),
        ("algo", C.c_int),
        ("split_k", C.c_int), ("wino_tile", C.c_int),
        ("pw", C.POINTER(PwExtras)),
        ("mfma_terms", C.c_int),
        ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
        ("add2", C.c_void_p), ("ld_add2", C.c_int), ("out2", C.c_void_p), ("ld_out2", C.c_int),
        ("mask_mode", C.c_int),
    ]


WGRAD_MAX_GROUPS = 24    # == REFID_WGRAD_MAX_GROUPS in include/refid_hip.h


class WgradDesc(C.Structure):
    _fields_ = [
        ("g", C.c_void_p), ("ld_g", C.c_int), ("c_o", C.c_int),
        ("in_a", C.c_void_p), ("in_b", C.c_void_p),
        ("ld_a", C.c_int), ("ld_b", C.c_int),
        ("c_a", C.c_int), ("c_b", C.c_int),
        ("dw", C.c_void_p), ("db", C.c_void_p), ("slabs", C.c_void_p),
        ("n", C.c_int), ("h", C.c_int), ("w", C.c_int), ("ho", C.c_int), ("wo", C.c_int),
        ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("i_base", C.c_int), ("i_total", C.c_int), ("o_real", C.c_int), ("algo", C.c_int),
        ("phase", C.c_int),
        ("groups", C.c_int),
        ("g_more", C.c_void_p * (WGRAD_MAX_GROUPS - 1)), ("in_a_more", C.c_void_p * (WGRAD_MAX_GROUPS - 1)),
        ("in_b_more", C.c_void_p * (WGRAD_MAX_GROUPS - 1)),
    ]


ABI_VERSION = 9          # == REFID_ABI_VERSION in include/refid_hip.h
_lib = None


def lib():
    """The loaded library; raises RefidHipError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RefidHipError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU/eager fallback). "
            "Run `python -m refid_amd.build`.")
    L = C.CDLL(LIB_PATH)
    L.refid_last_error.restype = C.c_char_p
    L.refid_abi_version.restype = C.c_int
    L.refid_device_cu_count.restype = C.c_int
    L.refid_conv2d.argtypes = [C.POINTER(ConvDesc), C.c_void_p]
    L.refid_conv_workspace_bytes.argtypes = [C.POINTER(ConvDesc)]
    L.refid_conv_workspace_bytes.restype = C.c_size_t
    L.refid_conv_kc.argtypes = [C.c_int] * 4
    L.refid_conv_bn.argtypes = [C.c_int] * 5
    L.refid_conv_tile_name.argtypes = [C.c_int] * 5
    L.refid_conv_tile_name.restype = C.c_char_p
    L.refid_wgrad_workspace_bytes.argtypes = [C.POINTER(WgradDesc)]
    L.refid_wgrad_workspace_bytes.restype = C.c_size_t
    L.refid_conv2d_wgrad.argtypes = [C.POINTER(WgradDesc), C.c_void_p]
    L.refid_wgrad_finish_flush.argtypes = [C.c_void_p]
    L.refid_rows_sum_defer.argtypes = [C.c_int]
    L.refid_rows_sum_flush.argtypes = [C.c_void_p]
    L.refid_packed_weight_floats.argtypes = [C.c_int] * 7
    L.refid_packed_weight_floats.restype = C.c_size_t
    L.refid_packed_weight_split_bytes.restype = C.c_size_t
    L.refid_packed_weight_split_bytes.argtypes = [C.c_int] * 7
    L.refid_pack_conv_weights_split.argtypes = [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p]
    L.refid_pack_conv_weights.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]
    L.refid_nchw_to_nhwc.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    L.refid_nchw_tsum_to_nhwc.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_void_p] + [C.c_int] * 5 + \
        [C.c_void_p]
    L.refid_nhwc_to_nchw.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong] + [C.c_int] * 4 + [C.c_void_p]
    L.refid_nchw_to_nhwc_tb.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
    L.refid_nhwc_to_nchw_tb.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_longlong] + [C.c_int] * 5 + [C.c_void_p]
    L.refid_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
    L.refid_sum_n.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
    L.refid_act_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_longlong, C.c_void_p]
    _bind_extra(L)
    if L.refid_abi_version() != ABI_VERSION:
        raise RefidHipError("librefid_hip.so ABI version mismatch")
    _lib = L
    return L


def _bind_extra(L):
    """EGACA / train-step entry points (a stale .so without them fails loudly here)."""
    vp, i, f, ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
    L.refid_pack_conv_weights_scaled.argtypes = [vp, vp, vp] + [i] * 7 + [vp]
    L.refid_pack_conv_weights_bf16.argtypes = [vp, vp, vp] + [i] * 7 + [vp]
    L.refid_packed_weight_wino6_bytes.argtypes = [i] * 4
    L.refid_packed_weight_wino6_bytes.restype = C.c_size_t
    L.refid_pack_conv_weights_wino6.argtypes = [vp, vp, vp] + [i] * 4 + [vp]
    L.refid_packed_weight_wino3h_bytes.argtypes = [i] * 4
    L.refid_packed_weight_wino3h_bytes.restype = C.c_size_t
    L.refid_pack_conv_weights_wino3h.argtypes = [vp, vp, vp] + [i] * 4 + [vp]
    L.refid_pack_batch_prepass.argtypes = [vp, i, vp]
    L.refid_packed_weight_split_f16_bytes.argtypes = [i] * 6
    L.refid_packed_weight_split_f16_bytes.restype = C.c_size_t
    L.refid_pack_conv_weights_split_f16.argtypes = [vp, vp, vp] + [i] * 6 + [vp]
    L.refid_mul_vec.argtypes = [vp, vp, vp, i, vp]
    L.refid_pack_entry_bytes.restype = C.c_size_t
    L.refid_pack_entry_bytes.argtypes = []
    L.refid_pack_entry_fill.argtypes = [vp, i, vp, vp, vp, i, i, i, i, i, i, i, i, i]
    L.refid_pack_batch.argtypes = [vp, i, i, vp]
    L.refid_pack_table_check.argtypes = [vp, i]
    L.refid_fold_back.argtypes = [vp] * 8 + [i, i, vp]
    L.refid_layernorm2d_fwd.argtypes = [vp, i, vp, vp, vp, i, ll, i, f, vp]
    L.refid_layernorm2d_bwd.argtypes = [vp, i, vp, i, vp, vp, i, vp, i, vp, vp, vp, ll, i, f, vp]
    L.refid_layernorm2d_bwd_parts.argtypes = [ll, i]
    L.refid_colsum_parts.argtypes = [ll, i]
    L.refid_egaca_gs_reduce_parts.argtypes = [i, i]
    L.refid_dwconv3x3_gelu_fwd.argtypes = [vp, i, vp, vp, vp, vp, vp, i, i, i, i, vp]
    L.refid_dwconv3x3_bwd.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, i, i, i, i, vp]
    L.refid_dwconv3x3_bwd_parts.argtypes = [i, i, i]
    L.refid_se_fwd.argtypes = [vp, i, f, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
    L.refid_dwconv_pool_parts.argtypes = [i, i, i]
    L.refid_se_bwd.argtypes = [vp] * 12 + [i, i, vp]
    L.refid_scale_cat.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    L.refid_egaca_gs_reduce.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp]
    L.refid_egaca_bwd_elem.argtypes = [vp, vp, vp, f, vp, vp, vp, i, i, i, i, vp]
    L.refid_gelu_fwd.argtypes = [vp, vp, ll, vp]
    L.refid_gelu_bwd.argtypes = [vp, vp, vp, ll, vp]
    L.refid_colsum.argtypes = [vp, i, vp, vp, ll, i, vp]
    L.refid_charbonnier_parts.argtypes = [ll]
    L.refid_charbonnier.argtypes = [vp, vp, vp, vp, vp, ll, f, f, vp]
    L.refid_grad_sqnorm.argtypes = [vp, vp, ll, vp]
    L.refid_psnr_loss_parts.argtypes = [i, ll]
    L.refid_psnr_loss.argtypes = [vp, vp, vp, vp, vp, vp, i, ll, f, vp]
    L.refid_clip_adamw.argtypes = [vp, vp, vp, vp, vp, f, f, f, f, f, f, f, i, ll, vp]
    L.refid_clip_adamw_dev.argtypes = [vp, vp, vp, vp, vp, f, f, vp, f, f, f, f, ll, vp]
    d = C.c_double
    L.refid_events_to_voxel.argtypes = [vp, vp, vp, vp, ll, i, i, i, d, d, vp, vp]
    L.refid_sqerr_u8_parts.argtypes = [i, ll]
    L.refid_sqerr_u8.argtypes = [vp, vp, i, ll, vp, vp, vp]
    L.refid_ssim3d_u8_parts.argtypes = [i, i, i]
    L.refid_ssim3d_u8.argtypes = [vp, vp, i, i, i, vp, vp, vp]
    L.refid_tile_add.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, vp]
    L.refid_tile_normalize.argtypes = [vp, vp, i, i, i, vp]
    L.refid_hin_parts.argtypes = [i]
    L.refid_hin_lrelu_fwd.argtypes = [vp, i, vp, vp, vp, i, vp, vp, i, i, i, i, f, f, vp]
    L.refid_hin_lrelu_bwd.argtypes = [vp, i, vp, i, vp, i, vp, vp, vp, i, vp, vp, vp, vp, i, i, i, i, f, vp]
    L.refid_fac_fwd.argtypes = [vp, i, vp, i, vp, i, ll, i, vp]
    L.refid_fac_bwd.argtypes = [vp, i, vp, i, vp, i, vp, i, vp, i, ll, i, vp]


def check(rc, what):
    if rc != 0:
        msg = lib().refid_last_error().decode("utf-8", "replace")
        raise RefidHipError(f"{what} failed (rc={rc}): {msg}")
