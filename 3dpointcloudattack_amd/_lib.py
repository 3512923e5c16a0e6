"""ctypes binding of libpc3d_hip.so (the C-ABI declared in include/pc3d.h).

The product path has NO fallback: if the shared library is missing or a call fails, this module raises.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C 3dpointcloudattack_amd/csrc``.
"""
import ctypes
import os

# torch MUST be imported before libpc3d_hip.so is dlopen'ed: the torch wheel bundles its own libamdhip64.so
# (SONAME libamdhip64.so.7). Loaded first, it satisfies our library's NEEDED entry, so both share ONE HIP
# runtime (streams and device pointers are interchangeable). The other order loads a second runtime.
import torch  # noqa: F401,E402

from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PC3D_LIB_PATH") or os.path.join(_HERE, "libpc3d_hip.so")     # (the override: A/B builds in tools/)

_P = c_void_p
_I = c_int
_L = c_int64
_F = c_float
_D = ctypes.c_double
_PTS = [_P, _L, _L, _L]  # pointer + (batch, point, channel) element strides

# name -> argtypes; every entry returns int (0 ok) except the two noted below.
SIGNATURES = {
    "pc3d_nn_f32": _PTS + _PTS + [_I, _I, _I, _P, _P, _P],
    "pc3d_nn_i64_f32": _PTS + _PTS + [_I, _I, _I, _P, _P, _P, _P],
    "pc3d_nn_bidir_f32": _PTS + _PTS + [_I, _I, _I, _P, _P, _P, _P, _P],
    "pc3d_nn_bidir_shared_ws_bytes": [_I, _I, _I],
    "pc3d_nn_bidir_shared_f32": _PTS + _PTS + [_I, _I, _I, _P, _P, _P, _P, _P, _L, _P],
    "pc3d_gemm_nt_f32": [_P, _L, _P, _P, _P, _L, _F, _I, _I, _I, _I, _F, _P, _L, _P],
    "pc3d_gemm_nt_tiled_f32": [_P, _L, _P, _P, _P, _L, _F, _I, _I, _I, _I, _F, _P, _L, _I, _P],
    "pc3d_gemm_nt_res_f32": [_P, _L, _P, _P, _P, _L, _I, _I, _I, _I, _F, _P, _L, _I, _P],
    "pc3d_gate_f32": [_P, _P, _L, _F, _P, _P],
    "pc3d_gather_max_rows_f32": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "pc3d_gather_max_rows_bwd_f32": [_P, _P, _I, _I, _I, _I, _P, _I, _P],
    "pc3d_rev_index_i32": [_P, _I, _I, _I, _I, _P, _P, _P, _P],
    "pc3d_rev_gather_sum_f32": [_P, _L, _P, _L, _F, _P, _P, _I, _I, _I, _I, _P, _L, _P],
    "pc3d_scatter_rows_det_f32": [_P, _P, _L, _P, _L, _F, _I, _I, _I, _I, _P, _L, _I, _I, _P],
    "pc3d_att_scale_f32": [_P, _P, _L, _I, _P, _P, _P],
    "pc3d_att_scale_bwd_f32": [_P, _P, _P, _P, _L, _I, _P, _P],
    "pc3d_topk_desc_f32": [_P, _I, _I, _I, _P, _P],
    "pc3d_curve_attn_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P],
    "pc3d_curve_attn_bwd_ws_floats": [_I, _I, _I, _I, _I],
    "pc3d_curve_attn_bwd_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P],
    "pc3d_lpfa_prep_f32": [_P, _P, _P, _P, _P, _L, _I, _P, _P, _P],
    "pc3d_lpfa_prep_bwd_f32": [_P, _P, _P, _P, _L, _I, _P, _P, _P],
    "pc3d_kappa_f32": _PTS + _PTS + [_P, _I, _I, _I, _P, _P],
    "pc3d_kappa_bwd_f32": _PTS + _PTS + [_P, _P, _I, _I, _I, _P, _I, _P],
    "pc3d_kappa_gather_f32": _PTS + _PTS + [_I, _P, _P, _I, _I, _I, _P, _P, _P],
    "pc3d_geoa3_record_f32": [_P, _I, _I, _I, _P, _I, _P, _P, _I, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P],
    "pc3d_group_act_bwd_mask_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _I, _P],
    "pc3d_knn_graph_i32": _PTS + [_I, _I, _I, _P, _P, _P, _I, _P],
    "pc3d_knn_graph_hint_i32": _PTS + [_I, _I, _I, _P, _P, _P, _I, _P, _P],
    "pc3d_fps_threads_f32": [_I] + _PTS + [_I, _I, _I, _P, _P, _P],
    "pc3d_fps_pruned_f32": _PTS + [_I, _I, _I, _P, _P, _P],
    "pc3d_affine3_f32": _PTS + [_I, _I, _P, _P, _F, _I, _P, _L, _P],
    "pc3d_affine3_bwd_f32": [_P, _L, _I, _I, _I, _P, _F] + _PTS + _PTS + [_P],
    "pc3d_scatter_points_det_f32": [_P, _P, _I, _I, _I] + _PTS + [_P],
    "pc3d_rows_max_f32": [_P, _L, _I, _I, _P, _P, _P],
    "pc3d_gemm_nt_groupsum_f32": [_P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "pc3d_gemm_nt_groupsum_packed_f32": [_P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "pc3d_group_act_bwd_points_f32": [_P] * 6 + [_I] * 5 + [_F, _P, _P],
    "pc3d_group_max_linear_bwd_ks_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P],
    "pc3d_group_max_linear_bwd_sparse_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "pc3d_sa_chain_table_unit": [_I, _I, _I, _I, _I],
    "pc3d_sa_blocks_i32": [_P, _I, _I, _I, _I, _P, _P, _P, _P],
    "pc3d_sa_chain_tb_f32": [_P, _L, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P],
    "pc3d_sa_chain_f32": [_P, _L, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P],
    "pc3d_gemm_nt_gather_f32": [_P, _L, _P, _P, _I, _I, _I, _I, _F, _P, _P, _I, _I, _I, _F, _P, _L, _P, _P, _P],
    "pc3d_group_max_linear_bwd_mask_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "pc3d_gemm_nt_poolbwd_f32": [_P, _L, _P, _P, _I, _I, _F, _P, _I, _I, _P, _L, _P],
    "pc3d_group_reverse_list_len": [_I, _I],
    "pc3d_group_reverse_i32": [_P, _I, _I, _I, _I, _P, _P, _P, _P],
    "pc3d_group_act_bwd_rev_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "pc3d_geoa3_terms_f32": [_P] * 7 + [_I, _I, _I, _F, _F, _F, _P, _P, _P],
    "pc3d_geoa3_terms_bwd_f32": [_P] * 6 + [_I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P],
    "pc3d_lpfa_fused_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P],
    "pc3d_lpfa_fused_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P],
    "pc3d_estimate_normal_f32": _PTS + [_P, _I, _I, _I] + _PTS + [_P],
    "pc3d_group_act_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P],
    "pc3d_group_act_bwd_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _I, _P],
    "pc3d_rowreduce_f32": [_P, _I, _I, _I, _I, _P, _P],
    "pc3d_nn_bwd_f32": _PTS + _PTS + [_I, _I, _I]
    + [_P, _P, _L, _L, _F] + [_P, _P, _L, _L, _F]
    + _PTS + _PTS + [_I, _P],
    "pc3d_knn_f32": _PTS + _PTS + [_I, _I, _I, _I, _P, _P, _P],
    "pc3d_knn_hint_f32": _PTS + _PTS + [_I, _I, _I, _I, _P, _P, _P, _P],
    "pc3d_knn_bwd_f32": _PTS + _PTS + [_I, _I, _I, _I, _P, _P] + _PTS + _PTS + [_I, _P, _P],
    "pc3d_knn_self_bwd_f32": _PTS + [_I, _I, _I, _P, _P, _P] + _PTS + [_I, _P, _P],
    "pc3d_knn_outlier_loss_f32": [_P, _I, _I, _I, _F, _P, _P, _P],
    "pc3d_fps_f32": _PTS + [_I, _I, _I, _P, _P, _P],
    "pc3d_ball_query_f32": _PTS + _PTS + [_I, _I, _I, _F, _I, _P, _P],
    "pc3d_ball_query_kernel_f32": [_I] + _PTS + _PTS + [_I, _I, _I, _F, _I, _P, _P],
    "pc3d_group_gather_f32": _PTS + [_P, _I, _P] + _PTS + [_I, _I, _I, _I, _P, _P],
    "pc3d_group_gather_bwd_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "pc3d_knn_feat_f32": [_P, _I, _I, _I, _I, _P, _P],
    "pc3d_gather_max_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "pc3d_gather_max_bwd_f32": [_P, _P, _I, _I, _I, _P, _I, _P],
    "pc3d_graph_laplacian_f32": _PTS + [_P, _I, _I, _I, _P, _P],
    "pc3d_spectral_reproject_f32": [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "pc3d_rowdot3_f32": [_P, _P, _I, _I, _I, _I, _P, _P, _P],
    "pc3d_clip_f32": _PTS + _PTS + _PTS + [_I, _I, _I, _F] + _PTS + [_P],
    "pc3d_adam_clip_step_f32": _PTS + _PTS + _PTS + [_P, _P] + _PTS + _PTS + [_I, _I, _D, _D, _D, _D, _F, _P, _I, _P],
    "pc3d_i32_add": [_P, _I, _P],
    "pc3d_group_linear_max_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "pc3d_group_linear_max_kernel_f32": [_I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "pc3d_act_pool_f32": [_P, _I, _I, _I, _F, _P, _P, _P],
    "pc3d_act_pool_bwd_f32": [_P, _P, _P, _I, _I, _I, _F, _P, _P],
    "pc3d_curve_walk_fwd_f32": [_P] * 7 + [_I] * 6 + [_P] * 5 + [_P],
    "pc3d_curve_walk_bwd_f32": [_P] * 7 + [_I] * 6 + [_P] * 8 + [_I, _P],
    "pc3d_curve_walk_bwd_ws_floats": [_I] * 6,
    "pc3d_curve_agg_lds_bytes": [_I] * 5,
    "pc3d_curve_agg_kv_f32": [_P] * 9 + [_I] * 5 + [_P, _P, _P],
    "pc3d_curve_agg_kv_bwd_f32": [_P] * 11 + [_I] * 5 + [_P, _P],
    "pc3d_edge_act_f32": [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P],
    "pc3d_edge_act_bwd_f32": [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _I, _P, _P, _P],
    "pc3d_act_mean_f32": [_P, _I, _I, _I, _I, _F, _P, _P],
    "pc3d_act_mean_bwd_f32": [_P, _P, _I, _I, _I, _I, _F, _P, _P],
    "pc3d_edge_max_f32": [_P, _P, _I, _I, _I, _I, _F, _P, _P, _P],
    "pc3d_edge_max_cat_f32": [_P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _L, _P],
    "pc3d_edge_max_bwd_f32": [_P, _L, _P, _P, _I, _I, _I, _F, _P, _I, _P],
    "pc3d_edge_max_bwd_sum_f32": [_P, _L, _P, _L, _P, _P, _I, _I, _I, _F, _P, _P],
    "pc3d_edge_max_bwd_slice_f32": [_P, _L, _P, _P, _I, _I, _I, _F, _P, _I, _P],
    "pc3d_group_max_linear_bwd_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "pc3d_cls_tail_f32": [_P, _I, _I, _P, _P, _I, _P, _I, _F, _F, _P, _P, _P, _P, _P, _P],
    "pc3d_cw_update_f32": _PTS + _PTS + [_I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P] + _PTS + [_P, _P]
    + [_D, _D, _D, _D, _F, _P, _I, _I, _P, _P, _P],
    "pc3d_cw_bookkeep_f32": _PTS + _PTS + [_I, _I, _P, _P, _I, _P, _P, _P, _P] + _PTS + _PTS + [_P, _P, _P],
    "pc3d_cw_step_f32": _PTS + _PTS + [_P, _P] + _PTS + [_I, _I, _D, _D, _D, _D, _F, _P, _I, _I, _P, _P, _P, _P],
    "pc3d_pairwise_f32": _PTS + _PTS + [_I, _I, _I, _I, _P, _P],
    "pc3d_pointmlp3_tile_points": [],
    "pc3d_pointmlp3_max_fwd_f32": _PTS + [_I, _I] + [_P] * 7 + [_I, _I, _I, _I] + [_P] * 6 + [_P],
    "pc3d_pointmlp3_max_fwd_th_f32": _PTS + [_I, _I] + [_P, _P, _P, _I, _P] + [_P] * 6 + [_I, _I, _I, _I] + [_P] * 6 + [_P],
    "pc3d_linear_pre_f32": [_P, _I, _I, _I, _P, _P, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P],
    "pc3d_pointmlp3_max_bwd_f32": _PTS + [_I, _I] + [_P] * 7 + [_I, _I, _I] + [_P, _P, _P, _P] + _PTS + [_P, _I, _P],
    "pc3d_pointmlp3_bwd_tile_points": [],
    "pc3d_linear_f32": [_P, _I, _I, _I, _I, _P, _P, _I, _I, _F, _P, _I, _F, _P, _I, _P],
    "pc3d_cls_loss_f32": [_P, _I, _I, _I, _P, _I, _F, _F, _P, _P, _P, _P, _P],
}

# entry points that do not return a status code
RESTYPES = {"pc3d_nn_bidir_shared_ws_bytes": c_int64, "pc3d_curve_attn_bwd_ws_floats": c_int64,
            "pc3d_group_reverse_list_len": c_int64, "pc3d_curve_agg_lds_bytes": c_int64,
            "pc3d_curve_walk_bwd_ws_floats": c_int64}

_lib = None


class Pc3dError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises Pc3dError when the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Pc3dError(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback — run "
            "`make -C 3dpointcloudattack_amd/csrc` (or __graft_entry__.build())."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.pc3d_version.restype = c_int
    lib.pc3d_version.argtypes = []
    lib.pc3d_last_error.restype = c_char_p
    lib.pc3d_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = RESTYPES.get(name, c_int)
        fn.argtypes = argtypes
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an int-returning entry point and raise Pc3dError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.pc3d_last_error().decode("utf-8", "replace")
        raise Pc3dError(f"{name} failed (rc={rc}): {msg}")
    return rc
