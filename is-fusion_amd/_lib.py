"""ctypes binding of libisf_hip.so (C ABI declared in include/isf_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C is-fusion_amd/csrc``.  There is NO
CPU or PyTorch fallback: if the shared library is missing or a call fails, the ops raise.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libisf_hip.so")

c_int = ctypes.c_int
c_void_p = ctypes.c_void_p
c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)

ISF_OK = 0
REDUCE = {"sum": 0, "mean": 1, "max": 2}
CONV_SUBM, CONV_SPARSE = 0, 1


class ConvLayer(ctypes.Structure):
    """struct isf_conv_layer (include/isf_hip.h)."""
    _fields_ = [
        ("conv_type", c_int),
        ("ksize", c_int * 3), ("stride", c_int * 3), ("padding", c_int * 3),
        ("c_in", c_int), ("c_out", c_int),
        ("packed", c_void_p), ("packed16", c_void_p), ("scale", c_void_p), ("shift", c_void_p),
        ("relu", c_int), ("residual_from", c_int),
    ]


class EncoderStats(ctypes.Structure):
    """struct isf_encoder_stats."""
    _fields_ = [
        ("num_layers", c_int),
        ("num_in", c_int * 32), ("num_out", c_int * 32),
        ("pairs", ctypes.c_longlong * 32),
        ("ms", ctypes.c_float * 32),
        ("precision", c_int),
    ]


class EncoderOptions(ctypes.Structure):
    """struct isf_encoder_options: per-call precision (0 auto f16x3 / 1 fp32 MFMA / 2 single-pass f16) and timing
    diagnostic of the conv kernels (0 off)."""
    _fields_ = [("precision", c_int), ("diagnostic", c_int), ("stage_rows", c_int), ("stage_mask", c_int),
                ("bev_format", c_int)]


def encoder_options(precision=0, diagnostic=0, stage_rows=0, stage_mask=0, bev_format=0):
    """-> byref(isf_encoder_options) or None for the defaults.  stage_rows: LDS-staged input rows per conv tile
    (0 = library default, -1 = off); stage_mask: layers that run staged when stage_rows > 0 (0 = all); bev_format 1: the
    BEV map as split-format token matrices (one per 256-channel group) instead of fp32 [B, C*D, H, W]."""
    if not precision and not diagnostic and not stage_rows and not stage_mask and not bev_format:
        return None
    return ctypes.byref(EncoderOptions(int(precision), int(diagnostic), int(stage_rows), int(stage_mask), int(bev_format)))


class ConvCuPlan(ctypes.Structure):
    """struct isf_conv_cu_plan: unit plan of the one-workgroup-per-CU sparse-conv kernel (pointers into plan_buf)."""
    _fields_ = [("group_masks", c_void_p), ("units", c_void_p), ("num_units", c_void_p), ("max_units", c_int),
                ("num_out", c_int), ("variant", c_int), ("cap", c_int)]


class VfeParams(ctypes.Structure):
    """struct isf_vfe_params."""
    _fields_ = [
        ("in_channels", c_int), ("c1", c_int), ("c2", c_int),
        ("w1", c_void_p), ("scale1", c_void_p), ("shift1", c_void_p),
        ("w2", c_void_p), ("scale2", c_void_p), ("shift2", c_void_p),
        ("voxel_size", ctypes.c_float * 3), ("coors_range", ctypes.c_float * 6),
    ]


class Sweep(ctypes.Structure):
    """isf_sweep_t."""
    _fields_ = [
        ("first_point", ctypes.c_int64), ("num_points", ctypes.c_int32), ("sample", ctypes.c_int32),
        ("is_sweep", ctypes.c_int32), ("remove_close", ctypes.c_int32), ("close_radius", ctypes.c_float),
        ("time_lag", ctypes.c_float), ("rotation", ctypes.c_double * 9), ("translation", ctypes.c_double * 3),
    ]


class PointAug(ctypes.Structure):
    """isf_point_aug_t."""
    _fields_ = [
        ("enabled", ctypes.c_int32), ("rot_mat_T", ctypes.c_float * 9), ("translation", ctypes.c_float * 3),
        ("scale", ctypes.c_float), ("flip_horizontal", ctypes.c_int32), ("flip_vertical", ctypes.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/isf_hip.h declares
_F3 = ctypes.c_float * 3
_F6 = ctypes.c_float * 6
_I3 = c_int * 3
_I4 = c_int * 4
SIGNATURES = {
    "isf_version": (c_int, []),
    "isf_last_error": (ctypes.c_char_p, []),
    "isf_device_count": (c_int, [c_int_p]),
    "isf_release_workspace": (c_int, []),
    "isf_workspace_bytes": (c_int, [ctypes.POINTER(ctypes.c_size_t)]),
    "isf_debug_workspace_blocks": (c_int, [ctypes.POINTER(ctypes.c_ulonglong), c_int, c_int_p]),
    "isf_dynamic_voxelize": (c_int, [c_void_p, c_int, c_int, _F3, _F6, c_void_p, c_void_p]),
    "isf_dynamic_voxelize_batched": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int64), c_int, c_int, _F3,
                                             _F6, c_void_p, c_void_p]),
    "isf_hard_voxelize": (c_int, [c_void_p, c_int, c_int, _F3, _F6, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_int_p, c_void_p]),
    "isf_hard_voxelize_device": (c_int, [c_void_p, c_int, c_int, _F3, _F6, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "isf_hard_voxelize_batched_device": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int64), c_int, c_int, _F3, _F6, c_int,
                                                 c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "isf_dynamic_point_to_voxel_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                                   c_void_p, c_void_p, c_void_p, c_int_p, c_void_p]),
    "isf_dynamic_point_to_voxel_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                    c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "isf_dynamic_vfe_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, _F3, _F6, c_void_p,
                                        c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                        c_void_p, c_void_p, c_void_p, c_int_p, c_void_p]),
    "isf_hard_simple_vfe": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isf_nbr_stride": (c_int, [c_int]),
    "isf_conv_out_shape": (c_int, [_I3, _I3, _I3, _I3, _I3]),
    "isf_build_rulebook": (c_int, [c_void_p, c_int, c_int, _I3, _I3, _I3, _I3, c_int, c_void_p, c_int,
                                   c_void_p, c_int, c_int_p, c_void_p]),
    "isf_rulebook_to_indice_pairs": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                             c_void_p]),
    "isf_indice_pairs_to_rulebook": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                             c_void_p]),
    "isf_sparse_conv_forward": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                        c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "isf_packed_filter_elems": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "isf_pack_filters": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isf_sparse_conv_forward_packed": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                                               c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                               c_void_p]),
    "isf_packed_filter16_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "isf_pack_filters_f16x3": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isf_pack_filters_f16x3_transposed": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isf_f32_to_split": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, c_void_p]),
    "isf_split_to_f32": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, c_void_p]),
    "isf_f32_to_half": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, c_void_p]),
    "isf_half_to_f32": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, c_void_p]),
    "isf_sparse_conv_forward_f16x3": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                              c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "isf_sparse_conv_tile_order": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                           ctypes.POINTER(c_int), c_void_p]),
    "isf_sparse_conv_forward_f16x3_ordered": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                                      c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                                      c_void_p, c_void_p]),
    "isf_sparse_conv_forward_dma": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                            c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "isf_sparse_conv_tile_table": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                           ctypes.POINTER(c_int), c_void_p]),
    "isf_sparse_conv_forward_f16x3_tiled": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                                    c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                                    c_void_p, c_void_p]),
    "isf_sparse_conv_tile_table_host": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                                ctypes.POINTER(c_int)]),
    "isf_sparse_conv_trace": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                                      c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                      ctypes.POINTER(c_int), c_void_p]),
    "isf_sparse_conv_phase_trace": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                            c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                            ctypes.c_size_t, c_int_p, c_int_p, c_int_p, c_void_p]),
    "isf_rulebook_to_lines": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "isf_lines_to_rulebook": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isf_sparse_conv_forward_dma_lines": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                                  c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                                  c_void_p]),
    "isf_sparse_conv_cu_plan_ints": (c_int, [c_int, ctypes.POINTER(ctypes.c_size_t)]),
    "isf_sparse_conv_cu_plan": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, ctypes.POINTER(ConvCuPlan), c_void_p]),
    "isf_sparse_conv_forward_cu": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                           c_void_p, c_void_p, c_void_p, c_int, c_void_p, ctypes.POINTER(ConvCuPlan),
                                           c_void_p]),
    "isf_sparse_conv_cu_plan_host": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, ctypes.POINTER(c_int)]),
    "isf_sparse_conv_cu_max_units": (c_int, [c_int, c_int]),
    "isf_stage_unit_rows": (c_int, []),
    "isf_stage_unit_cap": (c_int, []),
    "isf_rulebook_stage_tables": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "isf_sparse_conv_forward_staged": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                               c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                               c_int, c_int, c_void_p]),
    "isf_ms_deform_attn_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                           c_int, c_int, c_int, c_void_p, c_void_p]),
    "isf_ms_deform_attn_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                            c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "isf_ingroup_indices": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "isf_packed_window_block_bytes": (ctypes.c_size_t, [c_int]),
    "isf_pack_window_block": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "isf_window_block_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p]),
    "isf_p2g_backward": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                 c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "isf_sparse_to_dense_bev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p]),
    "isf_sparse_encoder_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, _I3,
                                           ctypes.POINTER(ConvLayer), c_int, c_void_p, _I4,
                                           ctypes.POINTER(EncoderStats), c_int, ctypes.POINTER(EncoderOptions),
                                           c_void_p]),
    "isf_lidar_branch_forward": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_int64), c_int,
                                         ctypes.POINTER(VfeParams), _I3, ctypes.POINTER(ConvLayer), c_int,
                                         c_void_p, _I4, ctypes.POINTER(EncoderStats), c_int,
                                         ctypes.POINTER(EncoderOptions), c_void_p]),
    "isf_packed_linear_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "isf_pack_linear": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "isf_pack_linear_transposed": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "isf_linear_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                   c_int, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_int, c_int, c_int,
                                   c_int, c_void_p]),
    "isf_window_attention_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                             c_void_p]),
    "isf_attention_forward": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_void_p, c_int, c_void_p]),
    "isf_attention_forward_dropout": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                              c_int, ctypes.c_float, ctypes.c_uint64, c_void_p, c_int, c_void_p]),
    "isf_channel_attention_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "isf_p2g_forward": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isf_p2g_forward_split": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                      c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isf_instance_topk": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "isf_msda_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_void_p, c_void_p]),
    "isf_decode_boxes": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_float)]
                         + [c_void_p] * 5),
    "isf_sparse_conv_dma_trace": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                         c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                         ctypes.POINTER(c_int), c_void_p]),
    "isf_instance_gather": (c_int, [c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 10),
    "isf_head_query_init": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 14),
    "isf_head_scatter_predictions": (c_int, [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int),
                                            ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_void_p),
                                            c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "isf_assemble_points": (c_int, [c_void_p, ctypes.POINTER(Sweep), c_int, c_int, ctypes.POINTER(PointAug),
                                    ctypes.POINTER(ctypes.c_float), c_void_p, c_void_p,
                                    ctypes.POINTER(ctypes.c_int32), c_void_p]),
    "isf_transpose_rulebook": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "isf_sparse_conv_backward_input": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int,
                                               c_void_p, c_void_p]),
    "isf_sparse_conv_backward_filter": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                                c_int, c_void_p, c_void_p]),
    "isf_bn1d_stats": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "isf_bn1d_apply": (c_int, [c_void_p, c_int, c_int, c_void_p, ctypes.c_float, c_void_p, c_void_p, ctypes.c_float,
                               ctypes.c_float, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "isf_bn1d_stats_pivot": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "isf_bn1d_apply_pivot": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p,
                                     ctypes.c_float, ctypes.c_float, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                     c_void_p, c_void_p]),
    "isf_bn1d_apply_pivot_counted": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p,
                                             ctypes.c_float, ctypes.c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int, c_void_p, c_void_p, c_void_p]),
    "isf_bn1d_backward_sums": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "isf_bn1d_backward_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "isf_pair_list_capacity": (c_int, [c_int, c_int]),
    "isf_rulebook_pair_lists": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "isf_grad_to_split": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, c_void_p, c_void_p]),
    "isf_grad_rescale": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, c_void_p, c_void_p]),
    "isf_split_to_f32_scaled": (c_int, [c_void_p, ctypes.c_size_t, c_void_p, c_void_p, c_void_p]),
    "isf_sparse_conv_backward_filter_f16x3": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                                      c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "isf_msda_backward": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p] * 4),
    "isf_attention_backward": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                       c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                       c_void_p]),
    "isf_attention_backward_dropout": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                               c_int, c_int, c_int, c_int, ctypes.c_float, ctypes.c_uint64, c_void_p, c_int,
                                               c_void_p, c_void_p, c_int, c_void_p]),
    "isf_window_attention_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                              c_void_p]),
    "isf_dense_grid_rulebook": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                        c_int * 2, c_void_p]),
    "isf_nchw_to_split": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "isf_split_to_nchw": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
}

_lib = None


class IsfError(RuntimeError):
    pass


def load():
    """Load libisf_hip.so; raises (loudly) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IsfError(
            f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C is-fusion_amd/csrc).  isfusion_amd has no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != ISF_OK:
        msg = load().isf_last_error().decode("utf-8", "replace")
        raise IsfError(f"{what or 'libisf_hip'} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def stream():
    """hipStream_t of torch's current stream on the current device, as an int.  Through the two C accessors torch's own
    compiled-graph runtime uses: torch.cuda.current_stream().cuda_stream builds a Stream object and resolves the device
    index in Python -- 5-10 us a call, ~800 calls per training step, which is paced by the host (tools/train_host_profile.py)."""
    global _raw_stream
    if _raw_stream is None:
        import torch
        get, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        if get is not None and dev is not None:
            _raw_stream = lambda: get(dev())
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream
    return _raw_stream()


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise IsfError("isfusion_amd ops run on the GPU only (HIP kernels); got a CPU tensor. "
                           "There is deliberately no CPU fallback.")


def f3(v):
    return _F3(*[float(x) for x in v])


def f6(v):
    return _F6(*[float(x) for x in v])


def i3(v):
    return _I3(*[int(x) for x in v])


def grad_rescale(g):
    """(g * s, scale float32 [2] = {s, 1 / s}) through isf_grad_rescale: the power-of-two scale of pow2_rescale (largest
    finite |g| into [2^9, 2^10)) in two launches, no host sync"""
    import torch
    g = g.contiguous().float()
    out = torch.empty_like(g)
    sc = torch.empty(2, dtype=torch.float32, device=g.device)
    check(load().isf_grad_rescale(ptr(g), g.numel(), ptr(out), ptr(sc), stream()), "isf_grad_rescale")
    return out, sc


def pow2_rescale(g):
    """(g * s, s) with s a power of two (a device scalar, no host sync) that brings max|g| to about 2^10.  The f16x3
    kernels carry fp32 operands as f16 hi + lo halves: exact down to 2^-24 of f16's normal range, so operands must sit
    inside it -- activations of a normalised network do, gradients (1e-5 and below) do not and would fall into f16's
    subnormals.  Scaling by a power of two is exact and commutes with the (linear) backward ops."""
    import torch
    a = g.detach().abs()
    # the scale comes from the largest FINITE entry: an inf / NaN in one row must stay in that row (amax = inf would give
    # s = 0 and turn the whole tensor into NaN through g * 0 / 0); non-finite entries pass through the scaling unchanged
    amax = torch.where(torch.isfinite(a), a, torch.zeros_like(a)).amax().clamp_min(1e-30)
    s = torch.exp2(10.0 - torch.ceil(torch.log2(amax)))
    return g * s, s
