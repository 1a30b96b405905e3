"""ctypes binding of librenderih_amd.so (the C ABI declared in include/renderih_amd.h).

The product path has no fallback: if the library is missing or a kernel launch fails, it raises.
"""
import ctypes as C
import os

from . import _build

c_f = C.c_void_p          # device float*
c_i = C.c_int
c_l = C.c_int64
c_u64 = C.c_uint64
c_fl = C.c_float


class GemmDesc(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('C', C.c_void_p), ('bias', C.c_void_p), ('R', C.c_void_p),
                ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
                ('lda', C.c_int32), ('ldb', C.c_int32), ('ldc', C.c_int32), ('ldr', C.c_int32),
                ('a_mode', C.c_int32), ('b_mode', C.c_int32),
                ('nb1', C.c_int32), ('nb2', C.c_int32),
                ('sA1', C.c_int64), ('sA2', C.c_int64), ('sB1', C.c_int64), ('sB2', C.c_int64),
                ('sC1', C.c_int64), ('sC2', C.c_int64),
                ('splitk', C.c_int32), ('kchunk', C.c_int32), ('sCsplit', C.c_int64),
                ('alpha', C.c_float), ('relu', C.c_int32),
                ('H', C.c_int32), ('W', C.c_int32), ('Cin', C.c_int32), ('Ho', C.c_int32), ('Wo', C.c_int32),
                ('KH', C.c_int32), ('KW', C.c_int32), ('strideA', C.c_int32), ('upS', C.c_int32),
                ('padH', C.c_int32), ('padW', C.c_int32), ('tile', C.c_int32), ('engine', C.c_int32),
                ('cS', C.c_int32), ('cOH', C.c_int32), ('cOW', C.c_int32), ('cH', C.c_int32), ('cW', C.c_int32), ('ones_row', C.c_int32),
                ('reserved0', C.c_int32), ('sBias1', C.c_int64), ('sR1', C.c_int64), ('stats', C.c_void_p),
                ('drop_p', C.c_float), ('reserved1', C.c_int32), ('drop_seed', C.c_uint64), ('drop_seed_dev', C.c_void_p),
                ('amax_a', C.c_void_p), ('amax_b', C.c_void_p),
                ('a_seg', C.c_void_p * 3), ('lda_seg', C.c_int32 * 3), ('k_seg', C.c_int32 * 3)]


class ReduceDesc(C.Structure):
    _fields_ = [('P', C.c_void_p), ('dst', C.c_void_p), ('db', C.c_void_p)] + \
               [(n, C.c_int32) for n in ('S', 'Mp', 'M', 'N', 'Cin', 'taps', 'CinValid', 'accumulate', 'CinPitch', 'reserved')]


class Conv3Desc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w_h2', C.c_void_p), ('y', C.c_void_p), ('stats', C.c_void_p), ('amax_x', C.c_void_p),
                ('amax_w', C.c_void_p)] + \
               [(n, C.c_int32) for n in ('imgs', 'H', 'W', 'C', 'N', 'ldx', 'ldy', 'Kpad', 'relu', 'ldr')] + [('r', C.c_void_p)]


class PanelDesc(C.Structure):
    _fields_ = [('a', C.c_void_p), ('w_h2', C.c_void_p), ('c', C.c_void_p), ('r', C.c_void_p), ('stats', C.c_void_p),
                ('amax_a', C.c_void_p), ('amax_w', C.c_void_p)] + \
               [(n, C.c_int32) for n in ('M', 'N', 'K', 'lda', 'ldc', 'ldr', 'relu')]


class H2Desc(C.Structure):
    _fields_ = [('w', C.c_void_p), ('dst', C.c_void_p), ('amax', C.c_void_p)] + \
               [(n, C.c_int32) for n in ('Cout', 'Cin', 'KH', 'KW', 'CinPad', 'for_dgrad', 'Kpad')]


class AbsmaxDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('out', C.c_void_p), ('n', C.c_int64)]


BOUND_FLOATS = 2048     # = RIH_BOUND_FLOATS: a bound block (64 partial maxima, one per 128-byte line)


class LnFinalDesc(C.Structure):
    _fields_ = [('ws', C.c_void_p), ('dg', C.c_void_p), ('db', C.c_void_p), ('D', C.c_int32), ('nblk', C.c_int32)]


class PackDesc(C.Structure):
    _fields_ = [('w', C.c_void_p), ('dst', C.c_void_p)] + \
               [(n, C.c_int32) for n in ('Cout', 'Cin', 'KH', 'KW', 'CinPad', 'mode', 'kh0', 'kw0', 'step', 'Th', 'Tw',
                                         'reserved')]


class HConvDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('zero', C.c_void_p), ('bias', C.c_void_p),
                ('post_scale', C.c_void_p), ('post_shift', C.c_void_p), ('res', C.c_void_p), ('y', C.c_void_p)] + \
               [(n, C.c_int32) for n in ('N', 'H', 'W', 'Cin', 'Cout', 'KH', 'KW', 'stride', 'pad', 'Ho', 'Wo',
                                         'ldx', 'ldr', 'ldy', 'Kpad', 'relu', 'out_f32')]


class ManoModel(C.Structure):
    _fields_ = [('comps', C.c_void_p), ('hands_mean', C.c_void_p), ('shapedirs', C.c_void_p),
                ('posedirs', C.c_void_p), ('v_template', C.c_void_p), ('J_reg', C.c_void_p),
                ('weights', C.c_void_p), ('parent', C.c_int32 * 16)]


class MeshTopo(C.Structure):
    _fields_ = [('faces', C.c_void_p), ('vptr', C.c_void_p), ('vlist', C.c_void_p), ('J', C.c_void_p),
                ('perm', C.c_void_p), ('V', C.c_int32), ('F', C.c_int32), ('NJ', C.c_int32), ('Vc', C.c_int32),
                ('pool', C.c_int32)]


# name -> (restype, argtypes); must list every symbol include/renderih_amd.h declares
SIGNATURES = {
    'rih_cdev': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_fl, c_f, C.c_void_p]),
    'rih_attention_bwd_dkv_fused': (c_i, [c_f, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f, c_f, c_i,
                                          C.c_void_p]),
    'rih_sdf': (c_i, [c_f, C.c_void_p, c_f, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_prepare_images': (c_i, [C.c_void_p, c_i, c_i, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_f, c_f, C.c_void_p,
                                 C.c_void_p]),
    'rih_prepare_labels': (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_f, C.c_void_p, c_fl, c_i, c_fl, c_f, c_f, c_f, C.c_void_p]),
    'rih_hconv': (c_i, [C.POINTER(HConvDesc), C.c_void_p]),
    'rih_hpack_conv_weight': (c_i, [c_f, c_f, C.c_void_p, c_i, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_hbn_fold': (c_i, [c_f, c_f, c_f, c_f, c_f, c_fl, c_f, c_f, c_i, C.c_void_p]),
    'rih_himage_nchw_to_nhwc8': (c_i, [c_f, C.c_void_p, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_hmaxpool3x3s2': (c_i, [C.c_void_p, C.c_void_p, c_i, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_hupsample2x': (c_i, [C.c_void_p, C.c_void_p, c_i, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_havgpool': (c_i, [C.c_void_p, c_f, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_attention_bwd_dq_fused': (c_i, [c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_fl, c_u64, c_f, c_f, c_f,
                                         c_i, c_f, c_i, C.c_void_p]),
    'rih_attention_fwd_fused': (c_i, [c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_fl, c_u64, c_f, c_f, c_f, c_i,
                                      c_f, c_i, C.c_void_p]),
    'rih_conv3x3_ok': (c_i, [C.POINTER(Conv3Desc)]),
    'rih_conv3x3_stats_rows': (c_i, [C.POINTER(Conv3Desc)]),
    'rih_conv3x3': (c_i, [C.POINTER(Conv3Desc), C.c_void_p]),
    'rih_h2_multi': (c_i, [C.POINTER(H2Desc), c_i, C.c_void_p]),
    'rih_panel_ok': (c_i, [C.POINTER(PanelDesc)]),
    'rih_panel_stats_rows': (c_i, [C.POINTER(PanelDesc)]),
    'rih_panel': (c_i, [C.POINTER(PanelDesc), C.c_void_p]),
    'rih_stem_ok': (c_i, [C.POINTER(Conv3Desc)]),
    'rih_stem': (c_i, [C.POINTER(Conv3Desc), C.c_void_p]),
    'rih_rows_ok': (c_i, [C.POINTER(PanelDesc)]),
    'rih_rows_stats_rows': (c_i, [C.POINTER(PanelDesc)]),
    'rih_rows': (c_i, [C.POINTER(PanelDesc), C.c_void_p]),
    'rih_hardswish_fwd': (c_i, [c_f, c_f, c_l, C.c_void_p]),
    'rih_hardswish_bwd': (c_i, [c_f, c_f, c_f, c_l, C.c_void_p]),
    'rih_tanh_scale_fwd': (c_i, [c_f, c_f, c_l, c_fl, C.c_void_p]),
    'rih_tanh_scale_bwd': (c_i, [c_f, c_f, c_f, c_l, c_fl, C.c_void_p]),
    'rih_rot6d_fwd': (c_i, [c_f, c_f, c_f, c_i, C.c_void_p]),
    'rih_rot6d_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, C.c_void_p]),
    'rih_rodrigues_fwd': (c_i, [c_f, c_f, c_i, C.c_void_p]),
    'rih_rodrigues_bwd': (c_i, [c_f, c_f, c_f, c_i, C.c_void_p]),
    'rih_center_scale_fwd': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_f, c_f, C.c_void_p]),
    'rih_center_scale_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_f, c_f, C.c_void_p]),
    'rih_hand_metrics': (c_i, [c_f] * 5 + [c_i] * 6 + [c_f] * 6 + [C.c_void_p]),
    'rih_gemm': (c_i, [C.POINTER(GemmDesc), C.c_void_p]),
    'rih_splitk_reduce': (c_i, [c_f, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_splitk_reduce_bias_batched': (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_f, c_i, c_l, c_l, c_l,
                                             C.c_void_p]),
    'rih_adam_multi': (c_i, [C.c_void_p, C.c_void_p, C.c_void_p, c_i, c_fl, c_fl, c_fl, c_fl, c_fl, c_i, c_i, C.c_void_p]),
    'rih_adam_chunk': (c_i, []),
    'rih_ln_param_final_multi': (c_i, [C.POINTER(LnFinalDesc), c_i, C.c_void_p]),
    'rih_pack_conv_weight_multi': (c_i, [C.POINTER(PackDesc), c_i, C.c_void_p]),
    'rih_splitk_reduce_multi': (c_i, [C.POINTER(ReduceDesc), c_i, C.c_void_p]),
    'rih_splitk_reduce_bias': (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_f, C.c_void_p]),
    'rih_splitk_finish': (c_i, [c_f, c_i, c_i, c_i, c_f, c_i, c_f, c_f, c_i, c_fl, c_i, C.c_void_p]),
    'rih_pack_conv_weight': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_pack_conv_weight_sub': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_nchw_to_nhwc': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_nhwc_to_nchw': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_maxpool3x3s2_fwd': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_maxpool3x3s2_bwd': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_avgpool_fwd': (c_i, [c_f, c_f, c_i, c_i, c_i, C.c_void_p]),
    'rih_avgpool_bwd': (c_i, [c_f, c_f, c_i, c_i, c_i, C.c_void_p]),
    'rih_upsample2x_fwd': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_upsample2x_bwd': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_upsample_bilinear_fwd': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_upsample_bilinear_bwd': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_nearest_up_add_fwd': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_nearest_up_bwd': (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_bn_ws_floats': (c_l, [c_i, c_i]),
    'rih_bn_stats': (c_i, [c_f, c_i, c_i, c_fl, c_fl, c_f, c_f, c_f, c_f, c_f, C.c_void_p]),
    'rih_bn_eval_stats': (c_i, [c_f, c_f, c_i, c_fl, c_f, c_f, C.c_void_p]),
    'rih_bn_apply': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, C.c_void_p, c_f, C.c_void_p]),
    'rih_bn_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, C.c_void_p, c_f, C.c_void_p]),
    'rih_layernorm_fwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_fl, c_i, C.c_void_p]),
    'rih_ln_nblk': (c_i, [c_i]),
    'rih_layernorm_fwd_grouped': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_l, c_l, c_fl, c_i, C.c_void_p]),
    'rih_layernorm_bwd_grouped': (c_i, [c_f] * 11 + [c_i, c_i, c_i, c_l, c_i, c_f, C.c_void_p]),
    'rih_layernorm_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f, C.c_void_p]),
    'rih_softmax_fwd': (c_i, [c_f, c_f, c_f, c_l, c_i, c_i, c_fl, c_u64, C.c_void_p, C.c_void_p]),
    'rih_softmax_bwd': (c_i, [c_f, c_f, c_l, c_i, c_i, c_fl, c_u64, C.c_void_p, c_fl, C.c_void_p]),
    'rih_add_dropout': (c_i, [c_f, c_f, c_f, c_l, c_i, c_i, c_fl, c_u64, C.c_void_p, C.c_void_p]),
    'rih_dropout_bwd': (c_i, [c_f, c_f, c_l, c_fl, c_u64, C.c_void_p, C.c_void_p]),
    'rih_relu_fwd': (c_i, [c_f, c_f, c_l, C.c_void_p]),
    'rih_relu_bwd': (c_i, [c_f, c_f, c_f, c_l, C.c_void_p]),
    'rih_colsum_ws_floats': (c_l, [c_i, c_i]),
    'rih_colsum': (c_i, [c_f, c_i, c_i, c_i, c_f, c_i, c_f, C.c_void_p]),
    'rih_gather_rows': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_scatter_rows_add': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, C.c_void_p]),
    'rih_project_fwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_fl, C.c_void_p]),
    'rih_project_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_fl, C.c_void_p]),
    'rih_cheby_fwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, C.c_void_p]),
    'rih_cheby_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, C.c_void_p]),
    'rih_mano_ws_floats': (c_l, [c_i]),
    'rih_mano_bwd_ws_floats': (c_l, [c_i]),
    'rih_mano_pack_floats': (c_l, []),
    'rih_mano_debug_stamps': (c_i, [C.c_void_p]),
    'rih_mano_pack': (c_i, [C.POINTER(ManoModel), c_f, C.c_void_p]),
    'rih_mano_fwd': (c_i, [C.POINTER(ManoModel), c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_i, c_i,
                           C.c_void_p]),
    'rih_mano_bwd': (c_i, [C.POINTER(ManoModel), c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f,
                           c_f, c_f, c_f, c_f, c_f, c_f, c_i, C.c_void_p]),
    'rih_mesh_loss': (c_i, [C.POINTER(MeshTopo), c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_fl,
                            c_f, c_f, c_f, c_f, c_f, c_i, C.c_void_p]),
    'rih_mesh_loss_final': (c_i, [c_f, c_f, c_i, c_f, c_f, c_f, C.c_void_p]),
    'rih_gemm_stats_rows': (c_i, [C.POINTER(GemmDesc)]),
    'rih_gemm_dropout_ok': (c_i, [C.POINTER(GemmDesc)]),
    'rih_gemm_engine': (c_i, [C.POINTER(GemmDesc)]),
    'rih_absmax': (c_i, [c_f, c_l, c_f, C.c_void_p]),
    'rih_absmax_multi': (c_i, [C.POINTER(AbsmaxDesc), c_i, C.c_void_p]),
    'rih_gemm_multi_variant': (c_i, [C.POINTER(GemmDesc)]),
    'rih_gemm_multi_table_bytes': (c_l, [C.POINTER(GemmDesc), c_i]),
    'rih_gemm_multi_pack': (c_i, [C.POINTER(GemmDesc), c_i, C.c_void_p, C.POINTER(C.c_int32)]),
    'rih_gemm_multi_launch': (c_i, [C.c_void_p, c_i, c_i, C.c_void_p]),
    'rih_bn_stats_from_blocks': (c_i, [c_f, c_i, c_i, c_i, c_i, c_fl, c_fl, c_f, c_f, c_f, c_f, C.c_void_p]),
    'rih_flash_attention_fwd': (c_i, [c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_fl, c_u64, C.c_void_p, c_f,
                                      c_i, c_f, C.c_void_p]),
    'rih_flash_attention_bwd': (c_i, [c_f, c_i, c_f, c_i, c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fl, c_fl,
                                      c_u64, C.c_void_p, c_f, c_f, c_f, c_i, c_f, c_f, c_i, C.c_void_p]),
    'rih_version': (c_i, []),
    'rih_abi_sizes': (c_i, [C.POINTER(C.c_int32)]),
    'rih_arch': (C.c_char_p, []),
}


ABI_VERSION = 19     # = RIH_ABI_VERSION of include/renderih_amd.h

_lib = None


def lib_path():
    return _build.LIB


def load():
    """Load (building if needed and possible) the shared library; raise if that fails."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the HIP runtime already in the process when our
    # library (same soname) is loaded, otherwise two runtimes disagree about the device (hipErrorNoDevice).
    import torch  # noqa: F401
    path = _build.LIB
    ab = os.environ.get('RIH_AB_LIB')       # kernel A/B experiments on one GPU box: load this build instead (ABI-checked below)
    if ab:
        path = ab
    elif not os.path.exists(path) or _build.needs_build():
        try:
            _build.build(verbose=False)
        except Exception as e:  # noqa: BLE001
            # never fall back silently to an older binary: its struct layouts / signatures may differ from this binding
            raise RuntimeError('librenderih_amd.so is %s and could not be rebuilt with hipcc (%s); run '
                               '`python -m renderih_amd._build`' % ('stale' if os.path.exists(path) else 'missing', e))
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError => a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # every by-pointer struct of the header, in rih_abi_sizes' order; the last one (rih_adam_entry: four pointers + int64) is
    # built by hand as int64 rows in renderih_amd/optim.py
    mine = [C.sizeof(GemmDesc), C.sizeof(ManoModel), C.sizeof(MeshTopo), C.sizeof(HConvDesc),
            C.sizeof(ReduceDesc), C.sizeof(PackDesc), C.sizeof(LnFinalDesc), 5 * 8, C.sizeof(AbsmaxDesc),
            C.sizeof(Conv3Desc), C.sizeof(H2Desc), C.sizeof(PanelDesc)]
    if lib.rih_version() != ABI_VERSION:
        raise RuntimeError('librenderih_amd.so does not match this binding (ABI %d vs %d): rebuild with '
                           '`python -m renderih_amd._build`' % (lib.rih_version(), ABI_VERSION))
    sizes = (C.c_int32 * len(mine))()
    if lib.rih_abi_sizes(sizes) != 0 or list(sizes) != mine:
        raise RuntimeError('librenderih_amd.so does not match this binding (struct sizes %s vs %s): rebuild with '
                           '`python -m renderih_amd._build`' % (list(sizes), mine))
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        raise RuntimeError('renderih_amd: %s failed with code %d' % (what, code))
