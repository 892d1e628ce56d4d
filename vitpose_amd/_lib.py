"""ctypes binding of libvitpose_hip.so (the C ABI declared in include/vitpose_hip.h).

The product path has no CPU fallback: if the shared library is missing or does not
export a symbol, importing this module's `lib()` raises.
"""
import ctypes
import os

import torch  # noqa: F401  (first: the library must bind to the HIP runtime torch has loaded)
from ctypes import POINTER, Structure, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 4   # VP_ABI_VERSION of include/vitpose_hip.h this binding was written for
LIB_PATH = os.environ.get('VITPOSE_HIP_LIB') or os.path.join(_HERE, 'lib', 'libvitpose_hip.so')

VP_EPI_BF16, VP_EPI_GELU, VP_EPI_RESID_F32, VP_EPI_DGELU = 0, 1, 2, 3
VP_EPI_SPLITK_F32, VP_EPI_F32, VP_EPI_SCALE_RELU, VP_EPI_DRELU, VP_EPI_PAIR = 4, 5, 6, 7, 8
VP_F32, VP_BF16 = 0, 1


class GemmArgs(Structure):
    _fields_ = [
        ('M', c_int), ('N', c_int), ('K', c_int),
        ('x', c_void_p), ('ldx', c_int), ('x_kmajor', c_int),
        ('y', c_void_p), ('ldy', c_int), ('y_kmajor', c_int),
        ('epilogue', c_int),
        ('out', c_void_p), ('ldo', c_int),
        ('out2', c_void_p),
        ('bias', c_void_p), ('colscale', c_void_p), ('resid', c_void_p),
        ('rowscale', c_void_p), ('rows_per_scale', c_int),
        ('aux', c_void_p), ('ldaux', c_int),
        ('pos', c_void_p), ('pos_rows', c_int),
        ('split_k', c_int), ('split_stride', c_long),
        ('out_row_div', c_int), ('out_row_hi', c_int), ('out_row_lo', c_int), ('out_row_off', c_int),
        ('tail_ws', c_void_p), ('tail_ws_bytes', c_size_t), ('tail_force', c_int), ('tile_mode', c_int),
        ('route_expert', c_void_p), ('route_delta', c_void_p), ('route_bias_delta', c_void_p), ('route_split', c_int),
        ('gather16_H', c_int), ('gather16_W', c_int), ('gather16_C', c_int), ('gather16_zero', c_void_p),
        ('split3', c_int), ('pair_act', c_int),
        ('ln_stats_out', c_void_p), ('ln_stats', c_void_p), ('ln_slots', c_int), ('ln_eps', c_float),
    ]


P = c_void_p
# name -> argtypes (the trailing stream argument included); every function returns int
class GroupEntry(Structure):
    _fields_ = [('x', c_void_p), ('y', c_void_p), ('out', c_void_p),
                ('ldx', c_int), ('ldy', c_int), ('ldo', c_int), ('M', c_int), ('N', c_int),
                ('tiles_n', c_int), ('ntiles', c_int), ('tile0', c_int), ('rowsum', c_void_p),
                ('image_mask', c_void_p), ('tiles_per_image', c_int)]


class LnPartial(Structure):
    _fields_ = [('partial', c_void_p), ('dgamma', c_void_p), ('dbeta', c_void_p), ('dcolsum', c_void_p),
                ('blocks', c_int), ('overwrite', c_int)]


SIGNATURES = {
    'vp_gemm_bf16': [POINTER(GemmArgs), P],
    'vp_ln_fold_weights': [P, P, P, P, P, P, P, c_int, c_int, P],
    'vp_deconv_fwd_bf16': [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P],
    'vp_deconv_weight_prep': [P, P, P, c_int, c_int, P],
    'vp_deconv_dgrad_bf16': [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P],
    'vp_wgrad_group_table': [POINTER(GroupEntry), c_int, c_int, POINTER(c_int)],
    'vp_wgrad_grouped_bf16': [P, c_int, c_int, c_int, c_int, P],
    'vp_wgrad_grouped_kmap_bf16': [P, c_int, c_int, c_int, P],
    'vp_splitk_reduce': [P, c_int, c_long, c_size_t, P, c_int, P],
    'vp_layernorm_fwd': [P, P, P, P, P, P, P, c_int, c_int, c_float, P],
    'vp_layernorm_bwd': [P, P, P, P, P, P, P, P, P, c_int, P, P, P, c_int, c_int, P, c_size_t, P],
    'vp_layernorm_bwd_finalize': [P, c_int, c_int, P],
    'vp_layernorm_bwd_static': [P, P, P, P, P, P, P, P, P, c_int, P, P, P, c_int, c_int, P],
    'vp_layernorm_bwd_blocks': [c_int, c_int, c_size_t],       # host-side helper: no stream argument
    'vp_gemm_ln_slots': [c_int, c_int],                        # host-side helper: no stream argument
    'vp_attention_fwd': [P, P, P, c_int, c_int, c_int, c_int, c_float, P],
    'vp_attention_bwd': [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P],
    'vp_patch_im2col': [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    'vp_strided_cast': [P, c_int, P, c_int, POINTER(c_int * 4), POINTER(c_long * 4), P],
    'vp_colsum_bf16': [P, P, c_int, c_int, c_int, P],
    'vp_batch_sum_f32': [P, P, c_int, c_size_t, P],
    'vp_deconv_phase_im2col': [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P],
    'vp_deconv_out_gather16': [P, P, c_int, c_int, c_int, c_int, P],
    'vp_cast_f32_bf16': [P, P, c_size_t, P],
    'vp_bn_stats': [P, P, c_int, c_int, P],
    'vp_bn_finalize': [P, c_int, c_int, P, P, P, P, c_float, c_float, P, P, P, P, P],
    'vp_bn_apply_relu': [P, P, P, P, c_int, c_int, P],
    'vp_bn_bwd': [P, P, P, P, P, P, P, P, c_int, c_int, P],
    'vp_im2col3x3_bf16': [P, P, c_int, c_int, c_int, c_int, P],
    'vp_col2im3x3_bf16': [P, P, P, c_int, c_int, c_int, c_int, P],
    'vp_resize_to_cl_bf16': [P, P, P, P, P, P] + [c_int] * 8 + [P],
    'vp_resize_to_cl_bwd': [P, P, P, P, P, P] + [c_int] * 10 + [P],
    'vp_deconv_weight_pad4': [P, P, c_size_t, c_int, P],
    'vp_bn_bwd_eval': [P, P, P, P, P, P, P, P, c_int, c_int, P],
    'vp_upconv_fwd': [P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    'vp_upconv_bwd': [P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    'vp_plane_sum_f32': [P, P, c_int, c_int, c_int, P],
    'vp_relu_bf16': [P, P, c_size_t, P],
    'vp_nchw_to_cl_bf16': [P, P, c_int, c_int, c_int, c_int, P],
    'vp_warp_affine_normalize': [P, P, P, P, P, c_int, c_int, c_int, P, P, P],
    'vp_generate_target': [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P],
    'vp_flip_back_avg': [P, P, P, c_int, P, c_int, c_int, c_int, c_int, P],
    'vp_decode_heatmaps': [P, P, P, c_int, c_int, c_int, c_int, P, c_int, P],
    'vp_decode_blurred': [P, P, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_float, P],
    'vp_split_pair_f32': [P, P, c_size_t, P],
    'vp_layernorm_fwd_pair': [P, P, P, P, P, c_int, c_int, c_int, c_float, P],
    'vp_attention_split': [P, P, c_int, c_int, c_int, c_int, c_float, P],
    'vp_patch_im2col_pair': [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    'vp_deconv_fwd_split': [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P],
    'vp_heatmap_loss': [P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_float, P],
    'vp_mse_loss_nchw': [P, P, P, P, P, c_int, c_int, c_int, c_float, P],
    'vp_argmax_rows': [P, c_int, c_int, P, P, P],
    'vp_sumsq_f32': [P, c_size_t, P, P],
    'vp_sumsq_f32_det': [P, c_size_t, P, P, c_int, P],
    'vp_sumsq_chunks_det': [P, P, P, c_int, P, P, c_int, P],
    'vp_drop_path_scales': [P, P, c_int, c_int, ctypes.c_ulonglong, ctypes.c_ulonglong, P],
    'vp_scale_bf16': [P, c_size_t, P, P],
    'vp_identity_norm_bwd': [P, P, c_int, P, P, c_int, c_int, P],
    'vp_scale_f32': [P, c_size_t, P, P],
    'vp_sum_bf16': [P, c_int, P, c_size_t, P],
    'vp_mask_rows_f32': [P, P, c_int, P, c_int, c_size_t, P],
    'vp_zero': [P, c_size_t, P],
    'vp_zero_regions': [P, P, c_int, P],
    'vp_pos_combine': [P, P, c_int, c_int, P],
    'vp_pos_embed_grad': [P, P, c_int, c_int, c_int, P],
    'vp_adamw_step': [P, P, P, P, P, P, P, P, P, c_int, c_float, c_float, c_float, c_float, c_int, P,
                      c_float, c_float, P],
}

_lib = None


def lib():
    """Load the HIP library once; raise loudly when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: build it with `make -C vitpose_amd/csrc` (or '
            '`python -c "import __graft_entry__ as g; g.build()"`). There is no CPU fallback.')
    L = ctypes.CDLL(LIB_PATH)
    L.vp_abi_version.restype = c_int
    if L.vp_abi_version() != ABI_VERSION:
        raise RuntimeError(f'{LIB_PATH} has ABI version {L.vp_abi_version()}, this binding needs {ABI_VERSION}: rebuild it')
    L.vp_build_arch.restype = ctypes.c_char_p
    for name, argtypes in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = c_int
    _lib = L
    return L


def exported_symbols():
    return ['vp_abi_version', 'vp_build_arch'] + list(SIGNATURES)
