"""ctypes binding of ``libococc_hip.so`` (C ABI declared in ``include/ococc_hip.h``).

The library is the product: there is no CPU or PyTorch fallback behind these
calls.  Importing this module without the built library raises, and every op
refuses tensors that do not live on a ROCm device.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('OCOCC_LIB_PATH') or os.path.join(_HERE, 'libococc_hip.so')   # (the override: diagnostic builds, tools/probe)

F32, BF16 = 0, 1
REDUCE = {'sum': 0, 'mean': 1, 'avg': 1, 'max': 2}

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
_F3 = ctypes.c_float * 3
_F6 = ctypes.c_float * 6
_I3 = ctypes.c_int32 * 3
_I4 = ctypes.c_int32 * 4

# name -> (restype, argtypes); one entry per function declared in ococc_hip.h
SIGNATURES = {
    'ococc_last_error': (ctypes.c_char_p, []),
    'ococc_version': (c_i32, []),
    'ococc_arch': (ctypes.c_char_p, []),
    'ococc_dynamic_voxelize_f32': (c_i32, [c_vp, c_i64, c_i32, _F3, _F6, c_vp, c_vp]),
    'ococc_hard_voxelize_workspace_bytes': (c_i64, [c_i64, _F3, _F6]),
    'ococc_hard_voxelize_f32': (c_i32, [c_vp, c_i64, c_i32, _F3, _F6, c_i32, c_i32, c_vp, c_vp,
                                        c_vp, c_vp, c_vp, c_i64, c_vp]),
    'ococc_grid_unique_workspace_bytes': (c_i64, [c_i32, _I4]),
    'ococc_grid_unique_workspace_layout': (c_i32, [c_i32, _I4, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    'ococc_grid_unique_i32': (c_i32, [c_vp, c_i64, c_i32, _I4, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp,
                                      c_vp, c_i64, c_vp]),
    'ococc_voxelize_scatter_workspace_bytes': (c_i64, [c_i64, c_i32, _I3]),
    'ococc_object_grid_geometry_workspace_bytes': (c_i64, [c_i64, c_i32, _I3, c_i32]),
    'ococc_object_grid_geometry_f32': (c_i32, [c_vp, c_i32, c_vp, c_i64, c_vp, c_i32, _F3, _F6, c_i32, _I3, c_i32,
                                               c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                               c_vp, c_i64, c_vp]),
    'ococc_object_grid_geometry_order_f32': (c_i32, [c_vp, c_i32, c_vp, c_i64, c_vp, c_i32, _F3, _F6, c_i32, _I3, c_i32,
                                                     c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                     c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    'ococc_voxelize_scatter_mean_f32': (c_i32, [c_vp, c_i32, c_vp, c_i64, c_vp, c_i32, _F3, _F6, c_i32, _I3,
                                                c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'ococc_occ_visibility_f64': (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_i32,
                                         c_vp, c_vp, c_vp, c_vp]),
    'ococc_segment_count_i32': (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp]),
    'ococc_segment_reduce_f32': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64,
                                         c_vp]),
    'ococc_segment_reduce_bwd_f32': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp,
                                             c_i64, c_vp]),
    'ococc_subm_rulebook_workspace_bytes': (c_i64, [c_i64, c_i32, _I3, _I3]),
    'ococc_subm_rulebook_build': (c_i32, [c_vp, c_i64, c_i32, _I3, _I3, _I3, c_vp, c_vp, c_vp, c_vp,
                                          c_vp, c_i64, c_vp]),
    'ococc_subm_rulebook_build_sorted': (c_i32, [c_vp, c_i64, c_i32, _I3, _I3, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                 c_i32, c_vp, c_i64, c_vp]),
    'ococc_conv_rulebook_workspace_bytes': (c_i64, [c_i64, c_i32, _I3, _I3]),
    'ococc_conv_rulebook_build': (c_i32, [c_vp, c_i64, c_i32, _I3, _I3, _I3, _I3, _I3, c_i32, c_vp, c_i64, c_vp,
                                          c_vp, c_vp, c_vp, c_i64, c_vp]),
    'ococc_rulebook_pairs_to_table': (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i32, c_i64, c_vp, c_vp,
                                              c_vp]),
    'ococc_sparse_conv_wgrad_reduce_multi': (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_indice_maxpool': (c_i32, [c_vp, c_i32, c_i64, c_i32, c_vp, c_i32, c_i64, c_vp, c_vp]),
    'ococc_indice_maxpool_backward': (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_i32, c_i64, c_vp, c_vp]),
    'ococc_sparse_conv_wgrad_multi_bf16': (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_backward_param_reduce_multi': (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_sparse_conv_tile_bf16': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_i64, c_vp, c_vp,
                                            c_i32, c_vp]),
    'ococc_sparse_conv_tile_ln_bf16': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_i64, c_vp, c_vp,
                                               c_f32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'ococc_sparse_conv_tile_lnbwd_partial_rows': (c_i64, [c_i64, c_i32, c_i32]),
    'ococc_sparse_conv_tile_lnbwd_bf16': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_i64, c_vp, c_vp,
                                                  c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_vp]),
    'ococc_sparse_conv_gather_gemm_bf16': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp,
                                                   c_vp, c_i64, c_vp, c_vp, c_i32, c_vp]),
    'ococc_subm_row_order_scratch_bytes': (c_i64, [c_i64]),
    'ococc_subm_row_order_counter_bytes': (c_i64, []),
    'ococc_subm_row_order_place': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'ococc_subm_row_order': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_sparse_conv_sorted_bf16': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64,
                                              c_vp, c_vp, c_i32, c_vp]),
    'ococc_sparse_conv_sorted_ln_bf16': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp,
                                                 c_f32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'ococc_sparse_conv_sorted_lnbwd_partial_rows': (c_i64, [c_i64]),
    'ococc_sparse_conv_sorted_lnbwd_bf16': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp,
                                                    c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_vp]),
    'ococc_sparse_conv_gather_gemm_ln_bf16': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_i64,
                                                      c_vp, c_vp, c_f32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'ococc_weight_prepare_multi_bf16': (c_i32, [c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                                ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), ctypes.POINTER(c_vp), c_vp]),
    'ococc_weight_prepare_bf16': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'ococc_sparse_conv_wgrad_workspace_bytes': (c_i64, [c_i32, c_i64, c_i32, c_i32]),
    'ococc_sparse_conv_wgrad_bf16': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp,
                                             c_i32, c_i64, c_vp, c_vp, c_i64, c_vp]),
    'ococc_layernorm_act_fwd': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_f32, c_i32, c_vp, c_vp,
                                        c_i32, c_vp]),
    'ococc_layernorm_act_bwd_workspace_bytes': (c_i64, [c_i64, c_i32]),
    'ococc_layernorm_act_bwd': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp,
                                        c_vp, c_vp, c_i32, c_vp, c_i64, c_vp]),
    'ococc_layernorm_act_dropout_fwd_bf16': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_f32, c_i32, ctypes.c_uint32,
                                                     ctypes.c_uint64, c_vp, c_vp, c_vp]),
    'ococc_layernorm_act_dropout_bwd_bf16': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i32, ctypes.c_uint32,
                                                     ctypes.c_uint64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'ococc_layernorm_act_bwd_partial_rows': (c_i32, [c_i64, c_i32, c_i32]),
    'ococc_layernorm_param_reduce_multi': (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_point_pool_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'ococc_dynamic_point_pool_mixed': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, _F3, c_i32, c_i64, c_vp,
                                               c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'ococc_aligned_iou3d_f32': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    'ococc_group_rank_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'ococc_group_rank_i32': (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'ococc_window_attn_fwd_bf16': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32,
                                           c_f32, c_vp, c_i64, c_vp, c_vp]),
    'ococc_window_attn_bwd_bf16': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64,
                                           c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]),
    'ococc_window_attn_fwd_gather_bf16': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32,
                                                  c_i32, c_f32, c_vp, c_i64, c_vp, c_vp]),
    'ococc_window_attn_bwd_gather_bf16': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp,
                                                  c_vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_i64, c_i64,
                                                  c_i64, c_vp]),
    'ococc_linear_fragments_bf16': (c_i32, [c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64),
                                            ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_vp), c_vp]),
    'ococc_window_tile_plan_workspace_bytes': (c_i64, [c_i64]),
    'ococc_window_block_partial_rows': (c_i64, [c_i64]),
    'ococc_window_tile_plan': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'ococc_window_attn_block_fwd_bf16': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp,
                                                 c_vp, c_vp, c_f32, c_vp, c_vp]),
    'ococc_window_attn_block_bwd_bf16': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp,
                                                 c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_window_attn_block_train_fwd_bf16': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp,
                                                       c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp]),
    'ococc_window_attn_block_bwd_saved_bf16': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp,
                                                       c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_token_ffn_block_fwd_bf16': (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32,
                                               c_i32, c_vp, c_vp]),
    'ococc_token_ffn_block_bwd_bf16': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32,
                                               c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_token_wgrad_slabs': (c_i64, [c_i64]),
    'ococc_token_wgrad_bf16': (c_i32, [c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_i64, c_i64, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), c_vp]),
    'ococc_partial_rows_sum_f32': (c_i32, [c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_vp), c_vp]),
    'ococc_point_mlp_fragment_floats': (c_i64, [c_i32, c_i32]),
    'ococc_point_mlp_tiles': (c_i64, [c_i64]),
    'ococc_point_mlp_pack_f32': (c_i32, [c_vp, c_i32, c_i32, c_i64, c_i64, c_vp, c_vp]),
    'ococc_point_mlp_pack_multi_f32': (c_i32, [c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_vp), c_vp]),
    'ococc_point_mlp_fwd_f32': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp, c_i32, c_vp,
                                        c_i64, c_vp, c_i32, c_vp, c_vp, c_f32, c_i32, c_vp, c_vp, c_i64, c_vp]),
    'ococc_point_mlp_segment_argmax': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp]),
    'ococc_point_mlp_wgrad_slices': (c_i32, [c_i64]),
    'ococc_point_mlp_force_tile': (c_i32, [c_i32]),
    'ococc_sir_layer_fwd_floats': (c_i64, [c_vp, c_i64, c_i64]),
    'ococc_sir_layer_fwd_f32': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'ococc_sir_layer_bwd_layout': (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_sir_layer_bwd_f32': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'ococc_rotate_z_f32': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    'ococc_points_box_to_box_f32': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    'ococc_roi_box_targets_f32': (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp]),
    'ococc_sir_rel_chain_fwd_floats': (c_i64, [c_vp, c_i64]),
    'ococc_sir_rel_chain_bwd_layout': (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_sir_rel_chains_fwd_f32': (c_i32, [c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'ococc_sir_rel_chains_bwd_f32': (c_i32, [c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_sir_layer_set_fused': (c_i32, [c_i32]),
    'ococc_sir_layer_fused_status': (c_i32, [c_vp, ctypes.POINTER(c_i32)]),
    'ococc_split3_bf16': (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i32, c_vp, c_i32, c_vp]),
    'ococc_temporal_attention_fwd_f32': (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                                 c_i32, c_f32, c_f32, ctypes.c_uint64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'ococc_temporal_attention_bwd_f32': (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32,
                                                 c_f32, ctypes.c_uint64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp,
                                                 c_i64, c_vp, c_i64, c_vp]),
    'ococc_sir_layer_fused_check': (c_i32, []),
    'ococc_sir_layer_fused_debug': (c_i32, [c_i32, c_i32]),
    'ococc_point_mlp_wgrad_f32': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    'ococc_point_mlp_wgrad_multi_f32': (c_i32, [c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    'ococc_point_mlp_bwd_f32': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_f32, c_vp, c_i32, c_vp,
                                        c_i64, c_vp, c_vp, c_i32, c_vp, c_vp, c_f32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_pos_encode_bf16': (c_i32, [c_vp, c_i64, ctypes.POINTER(c_f32), c_i32, c_vp, c_i32, c_vp]),
    'ococc_linear_fragments32_bf16': (c_i32, [c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_vp), c_vp]),
    'ococc_mlp_layer_fwd_bf16': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32,
                                         ctypes.c_uint32, ctypes.c_uint64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'ococc_occ_mlp_fwd_bf16': (c_i32, [c_vp, c_i64, c_vp, c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                       c_f32, c_vp, c_vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), c_vp, c_vp, c_vp,
                                       c_vp]),
    'ococc_occ_mlp_train_fwd_bf16': (c_i32, [c_vp, c_i64, c_vp, c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                             ctypes.POINTER(c_vp), c_f32, c_vp, c_vp, ctypes.c_uint32,
                                             ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                             ctypes.POINTER(c_vp), c_vp, c_vp]),
    'ococc_occ_mlp_bwd_workgroups': (c_i64, [c_i64]),
    'ococc_occ_mlp_bwd_partial_cols': (c_i64, []),
    'ococc_occ_mlp_bwd_scratch_bytes': (c_i64, [c_i64]),
    'ococc_occ_mlp_bwd_bf16': (c_i32, [c_vp, c_i64, c_vp, c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                       ctypes.POINTER(c_vp), c_f32, c_vp, c_vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64),
                                       ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                       c_vp, c_vp, c_i64, c_vp]),
    'ococc_sst_window_coors_i64': (c_i32, [c_vp, c_i64, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), c_vp, c_vp, c_vp]),
    'ococc_sst_drop_level_i64': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64),
                                         ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), c_vp, c_vp, c_vp]),
    'ococc_sst_pos_embed': (c_i32, [c_vp, c_i64, ctypes.POINTER(c_i32), c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp]),
    'ococc_cast_f32_to_bf16': (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    'ococc_cast_bf16_to_f32': (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    'ococc_adamw_f32': (c_i32, [c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), c_f32, c_f32, c_f32, c_f32, c_f32,
                                c_vp, c_i32, c_vp]),
    'ococc_adamw_lr_dev_f32': (c_i32, [c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                       ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), c_vp, c_f32, c_f32, c_f32, c_f32,
                                       c_vp, c_i32, c_vp]),
    'ococc_adamw_operands_f32': (c_i32, [c_i32, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp),
                                         ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), c_f32, c_vp, c_f32, c_f32, c_f32, c_f32,
                                         c_vp, c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                         ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                         ctypes.POINTER(c_vp), c_vp]),
    'ococc_timer_create': (c_i32, [ctypes.POINTER(c_vp)]),
    'ococc_timer_record': (c_i32, [c_vp, c_i32, c_vp]),
    'ococc_timer_elapsed_ms': (c_i32, [c_vp, c_vp, ctypes.POINTER(c_f32)]),
    'ococc_timer_destroy': (c_i32, [c_vp]),
}


class Timer(object):
    """One HIP event behind the C ABI (ococc_timer_*), recorded on torch's current stream."""

    def __init__(self):
        h = c_vp()
        check(lib.ococc_timer_create(ctypes.byref(h)), 'timer_create')
        self.h = h

    def record(self, in_graph=False):
        check(lib.ococc_timer_record(self.h, int(in_graph), stream()), 'timer_record')

    def elapsed_ms(self, stop):
        ms = c_f32()
        check(lib.ococc_timer_elapsed_ms(self.h, stop.h, ctypes.byref(ms)), 'timer_elapsed')
        return float(ms.value)

    def __del__(self):
        try:
            lib.ococc_timer_destroy(self.h)
        except Exception:  # interpreter shutdown
            pass


class OcoccError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build the HIP kernels first '
            '(python -c "import __graft_entry__ as g; g.build()" or '
            '`make -C objectcentricocccompletion_amd/csrc`). There is no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc, what=''):
    if rc != 0:
        msg = lib.ococc_last_error().decode(errors='replace')
        raise OcoccError(f'{what} failed with code {rc}: {msg}')


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise OcoccError('ococc ops run on a ROCm device only; got a CPU tensor '
                             '(there is no CPU fallback)')


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """hipStream_t of torch's current stream on the current device (the raw accessor: building a torch.cuda.Stream
    object per kernel launch cost ~11 us of host time, 1.4 ms per OcOccNet step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def f3(v):
    return _F3(*[float(x) for x in v])


def f6(v):
    return _F6(*[float(x) for x in v])


def i3(v):
    return _I3(*[int(x) for x in v])


def i4(v):
    v = list(v) + [1] * (4 - len(v))
    return _I4(*[int(x) for x in v])


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise OcoccError(f'unsupported dtype {dt}; the kernels take float32 or bfloat16')


_CONSTS = {}


def const_tensor(values, device, dtype=torch.float32):
    """A small read-only constant on `device`, uploaded once per (values, device, dtype): building it per call
    (torch.tensor(list, device=...)) is a pageable host-to-device copy, i.e. a host synchronisation, every time."""
    key = (tuple(float(v) for v in values), str(device), dtype)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(key[0], dtype=dtype, device=device)
    return t


_plan = None  # the BufferPlan in force (None: plain torch.empty)


def empty(shape, dtype, device):
    """torch.empty, or the next buffer of the BufferPlan in force."""
    if _plan is not None:
        return _plan.take(tuple(int(v) for v in shape), dtype, device)
    return torch.empty(shape, dtype=dtype, device=device)


class BufferPlan(object):
    """Caller-owned output memory for a chain of ops whose wrappers allocate through L.empty / L.workspace.

    The first ``with plan:`` block records every buffer the wrapped calls allocate; every later block hands the same
    tensors back in the same order (the calls must repeat with the same shapes and dtypes -- checked).  With two plans
    a captured HIP graph can write the geometry of the NEXT batch into one set of buffers while the training kernels
    of the same graph read the other set (graph.PipelinedStep)."""

    def __init__(self):
        self.bufs, self.pos, self.recorded = [], 0, False

    def take(self, shape, dtype, device):
        if not self.recorded:
            t = torch.empty(shape, dtype=dtype, device=device)
            self.bufs.append(t)
            return t
        if self.pos >= len(self.bufs):
            raise OcoccError('BufferPlan: more allocations than in the recorded run')
        t = self.bufs[self.pos]
        self.pos += 1
        if tuple(t.shape) != shape or t.dtype != dtype or t.device != torch.device(device):
            raise OcoccError(f'BufferPlan: allocation {self.pos - 1} was {tuple(t.shape)} {t.dtype}, now {shape} {dtype}')
        return t

    def __enter__(self):
        global _plan
        if _plan is not None:
            raise OcoccError('BufferPlan: plans do not nest')
        self.pos = 0
        _plan = self
        return self

    def __exit__(self, *exc):
        global _plan
        _plan = None
        if exc[0] is None and self.recorded and self.pos != len(self.bufs):
            raise OcoccError('BufferPlan: fewer allocations than in the recorded run')
        self.recorded = True
        return False


def workspace(nbytes, device):
    """Caller-owned scratch buffer (the C ABI never allocates)."""
    return empty((max(int(nbytes), 1),), torch.uint8, device)


_logged = set()


def log_once(key, msg, level='warning'):
    """one line on the package logger the first time ``key`` is seen: a fused kernel that does not cover a module's shape
    falls back to the operator-by-operator path -- correct, several times slower, and otherwise silent"""
    if key not in _logged:
        _logged.add(key)
        import logging
        getattr(logging.getLogger('objectcentricocccompletion_amd'), level)(msg)
