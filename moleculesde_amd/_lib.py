"""ctypes binding of libmsde_hip.so (include/msde_hip.h).  No torch types cross this boundary: only
raw device pointers, sizes and the HIP stream handle.  Fails loudly if the library is missing --
there is no CPU fallback anywhere in moleculesde_amd."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmsde_hip.so")

P = ctypes.c_void_p
I = ctypes.c_int
F = ctypes.c_float
LL = ctypes.c_longlong
ULL = ctypes.c_ulonglong

# name -> argtypes (restype is int unless listed in _RESTYPE)
SIGNATURES = {
    "msde_abi_version": [],
    "msde_target_arch": [],
    "msde_radius_count": [P, P, P, I, F, I, P, P],
    "msde_exclusive_scan_i32": [P, P, I, P],
    "msde_radius_fill": [P, P, P, I, F, I, P, P, P, P, I, P],
    "msde_segment_sum_rows": [P, I, P, P, I, I, F, P, I, P],
    "msde_segment_sum_rows2": [P, I, P, P, P, P, I, I, F, P, I, P],
    "msde_pair_gather_add": [P, P, I, P, P, I, I, P, P],
    "msde_pair_gather_add_stats": [P, P, I, P, P, I, I, P, P, P, P],
    "msde_segment_sum_rows_bn": [P, P, I, P, P, I, I, P, P, P, P, I, P],
    "msde_pair_strip": [],
    "msde_pair_bn_dgrad_stats": [P, I, P, P, P, P, P, I, I, I, P, P, P, P],
    "msde_pair_bn_scatter": [P, P, I, P, P, P, P, P, I, P, P, P, P, P],
    "msde_pair_gather_cat": [P, I, P, I, P, P, I, I, I, P, P],
    "msde_mlp_head_fwd": [P, I, P, P, I, I, I, P, P],
    "msde_mlp_head_bwd_slabs": [I, I],
    "msde_mlp_head_bwd": [P, I, P, P, I, I, I, P, P, P, P, P],
    "msde_mlp_head_mix_fwd": [P, I, P, P, I, P, P, I, P, P, P, P],
    "msde_mlp_head_mix_bwd": [P, I, P, P, P, P, P, I, I, P, P, P, P],
    "msde_gather_rows": [P, P, I, I, P, P],
    "msde_embedding_sum_fwd": [P, P, I, I, I, P, P],
    "msde_radius_transpose": [P, P, P, P, I, I, P, P, P, P],
    "msde_radius_transpose_mol": [P, I, I, P, P, I, I, P, P, P],
    "msde_embedding_sum_bwd_workspace_floats": [I, I, I],
    "msde_embedding_sum_bwd": [P, P, P, I, I, I, P, P, P],
    "msde_gin_aggregate_fwd": [P, P, P, P, P, P, I, I, P, P],
    "msde_gin_aggregate_bwd_x": [P, P, P, P, P, P, P, P, I, I, P, P],
    "msde_gin_aggregate_bn_fwd": [P, P, P, I, P, P, P, P, P, I, I, P, P, P],
    "msde_gin_aggregate_bwd_x_stats": [P, P, P, P, P, P, P, P, I, P, I, P, P, I, P, P, P],
    "msde_gin_aggregate_bwd_tab_workspace_floats": [I, I, I, I],
    "msde_gin_aggregate_bwd_tab": [P, P, P, P, P, P, I, I, I, I, P, P, P, P, P, P],
    "msde_gin_aggregate_bwd_tab_multi": [P, P, P, P, I, P, P, P, I, I, I, I, P, P, P],
    "msde_rbf_cutoff_fwd": [P, P, I, I, P, F, F, P, P, P],
    "msde_cfconv_aggregate_fwd": [P, P, P, P, P, I, I, P, P],
    "msde_cfconv_aggregate_bwd_w": [P, P, P, P, P, I, I, I, P, P],
    "msde_cfconv_aggregate_bwd_x": [P, P, P, P, P, P, I, I, P, P],
    "msde_cfconv_fused_fwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, F, I, P, P, P],
    "msde_cfconv_fused_bwd_w_workspace_floats": [I, I, I],
    "msde_cfconv_fused_bwd_w_slabs": [I, I],
    "msde_gin_aggregate_bwd_tab_slabs": [I, I],
    "msde_cfconv_fused_bwd_w": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, F, I, P, P, P, P, P, P],
    "msde_pair_build": [P, P, I, F, P, P, P, P, I, P, P],
    "msde_cfconv_pair_filter": [P, P, P, P, P, P, P, I, I, I, F, F, I, P, P],
    "msde_cfconv_pair_filter_multi": [P, P, P, P, P, P, P, I, I, I, I, F, F, I, P, P],
    "msde_cfconv_pair_bwd_w_multi_slabs": [I, I, I],
    "msde_cfconv_pair_bwd_w_multi": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, F, I, P, P],
    "msde_cfconv_pair_aggregate": [P, P, P, P, P, I, I, I, P, P],
    "msde_cfconv_pair_bwd_w": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, F, I, P, P, P, P, P, P],
    "msde_edge_geometry_fwd": [P, P, P, I, P, P, I, P, P, P, P, P, P],
    "msde_edge_geometry_fwd_ld": [P, P, P, I, P, P, I, P, P, P, I, P, I, I, P, P],
    "msde_edge_attention_fwd": [P, P, P, P, I, P, I, P, P, I, I, I, F, ULL, P, P, P, P],
    "msde_edge_attention_bwd": [P, P, P, P, I, P, I, P, I, P, P, P, I, I, I, F, ULL, P, P, P, P, P, I, P],
    "msde_frame_mix_mean_fwd": [P, P, P, I, P, P],
    "msde_frame_mix_mean_add_fwd": [P, P, P, I, P, P, P],
    "msde_frame_mix_mean_bwd": [P, P, P, I, I, P, P],
    "msde_linear_fwd": [P, P, P, I, I, I, P, P],
    "msde_linear_bwd_x": [P, P, I, I, I, P, P],
    "msde_linear_bwd_w_workspace_bytes": [I, I, I],
    "msde_linear_bwd_w_splits": [I, I, I],
    "msde_linear_bwd_w_describe": [P, P, I, I, I, I, P, P, P],
    "msde_linear_bwd_w_describe_ld": [P, I, P, I, I, I, I, I, P, P, P],
    "msde_linear_bwd_w_grouped": [P, P, I, I, P],
    "msde_linear_bwd_w_grouped_ex": [P, P, I, I, I, P],
    "msde_step_counters": [P, P, P],
    "msde_linear_bwd_w_partial": [P, P, I, I, I, I, P, P, P],
    "msde_reduce_slabs_multi": [P, P, I, I, P],
    "msde_reduce_slabs_chunks": [LL, I],
    "msde_linear_bwd_w": [P, P, I, I, I, P, P, P, P, P],
    "msde_gemm_ex": [P, P],
    "msde_gemm_rs": [P, P],
    "msde_gemm_t2": [P, P],
    "msde_gemm_t2_supported": [I, I, I, I],
    "msde_gemm_t2_geometry": [I, I, I, P, P],
    "msde_transpose_multi": [P, P, I, I, P],
    "msde_transpose": [P, P, I, I, P],
    "msde_relayout": [P, I, P, I, I, I, I, P],
    "msde_affine_cols": [P, I, I, P, P, I, P, P],
    "msde_bn_bwd_colstats": [P, P, P, P, I, P, I, P, P],
    "msde_bn_bwd_cols": [P, I, P, I, P, P, P, P, P, I, P, I, P, I, P],
    "msde_bn_bwd_fin_cols": [P, I, P, P, P, P, I, P, I, P, P, I, P, I, P, I, P, P, P],
    "msde_gemm_rs_geometry": [I, I, I, P, P],
    "msde_bn_fin_fwd": [P, I, I, I, P, I, P, P, F, F, P, P, P, P, P, P, P],
    "msde_bn_fin_bwd": [P, I, I, P, I, P, P, P, P, P, P, P, P, P],
    "msde_dense_prepare": [P, P, P, P, P, P, P, P, I, I, F, I, F, F, P, P, I, ULL, P, I, I, P, P, P, P, P, P, P],
    "msde_dense_edge_layer_fwd": [P, P, P, I, I, I, I, P, P, P, P, I, I, P, P, P, P, P, P, P],
    "msde_dense_edge_layer_bwd": [P, P, P, P, I, I, I, I, P, P, P, P, I, I, P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P,
                                  P],
    "msde_dense_node_gcn_fwd": [P, P, P, P, P, P, I, I, P, I, P],
    "msde_dense_node_gcn_bwd": [P, I, P, I, P, P, P, P, I, I, P, P, P],
    "msde_dense_loss_fwd": [P, I, P, P, P, P, P, P, P, P, P, I, I, F, F, F, P, P, P, P, P, P],
    "msde_dense_loss_bwd": [P, P, P, P, P, I, P, P, P, P, P, I, I, F, F, F, P, P, P, P, P],
    "msde_plan_build": [P, I, P, P, P, P, P, P, P, I, I, I, I, I, I, I] + [P] * 23 + [P],
    "msde_plan_row_lists": [P, P, I, I, P, P, P, P],
    "msde_dd_unary": [P, P, LL, I, I, F, P, P],
    "msde_dd_unary_mul": [P, P, P, P, LL, I, I, F, P, P],
    "msde_dd_rbf": [P, P, P, I, I, F, I, P, P],
    "msde_dd_binary": [P, P, LL, I, F, P, P],
    "msde_dd_sum_n": [P, I, LL, P, P],
    "msde_dd_sum_rows_n": [P, P, I, I, I, P, P],
    "msde_dd_mul_rows": [P, P, I, I, P, P],
    "msde_dd_row_dot": [P, P, I, I, P, P],
    "msde_dd_edge_diff": [P, P, P, I, P, P],
    "msde_dd_edge_scatter": [P, P, P, P, I, P, P],
    "msde_dd_row_norm": [P, P, I, P, P],
    "msde_dd_seg_expand": [P, P, P, I, I, I, P, P],
    "msde_dd_broadcast_rows": [P, I, I, P, P],
    "msde_dd_transpose3": [P, I, I, P, P],
    "msde_dd_merge3": [P, P, P, I, P, P],
    "msde_debug_stamp": [P, P],
    "msde_combine_losses": [P, P, P, P, F, F, F, F, P, P],
    "msde_combine_losses_bwd": [P, F, F, F, F, P, P],
    "msde_combine_losses_ex": [P, P, P, P, F, F, F, F, P, P, P, P, I, P],
    "msde_cl_ebm_fwd": [P, P, P, P, I, I, F, P, P, P, P, P, P],
    "msde_cl_ebm_bwd": [P, P, P, P, P, P, P, P, I, I, F, P, P, P, P],
    "msde_bn_workspace_floats": [I, I],
    "msde_colsum": [P, I, I, P, P, P, P],
    "msde_res_layernorm_fwd": [P, P, P, P, I, I, F, P, P, P, P],
    "msde_res_layernorm_bwd": [P, P, P, P, P, I, I, P, P, P, P, P],
    "msde_bn_fwd": [P, I, I, P, P, F, F, P, P, I, P, P, P, P, P, P],
    "msde_bn_bwd": [P, P, P, P, P, P, I, I, I, P, P, P, P, P, P],
    "msde_adam_flat": [P, P, P, P, LL, P, P, P, I, F, F, F, F, F, P],
    "msde_ssp_fwd": [P, LL, P, P],
    "msde_ssp_bwd": [P, P, LL, P, P],
    "msde_silu_dropout_fwd": [P, LL, F, ULL, P, P, P],
    "msde_silu_dropout_bwd": [P, P, LL, F, ULL, P, P, P],
    "msde_mul_add_fwd": [P, P, P, LL, P, P],
    "msde_mul_add_bwd": [P, P, P, LL, P, P, P],
    "msde_pc_corrector": [P, P, P, P, P, ULL, I, F, F, P, P, P],
    "msde_pc_predictor": [P, P, P, P, P, ULL, I, P, P, P],
    "msde_l1_energy_force_loss": [P, P, I, P, P, I, F, F, F, P, P, P, P],
    "msde_randperm": [I, I, ULL, P, P, P, P],
    "msde_ve_perturb": [P, P, P, P, I, I, I, F, F, F, P, P, P],
    "msde_ve_perturb_rng": [P, P, I, I, I, F, F, F, ULL, P, P, P, P, P],
    "msde_ve_pos_loss_fwd": [P, P, P, F, P, I, I, P, P, P],
    "msde_ve_pos_loss_bwd": [P, P, P, F, P, P, I, I, P, P, P],
    "msde_gat_tail_blocks": [I],
    "msde_gat_tail_fwd": [P, P, P, P, P, P, P, P, P, P, I, I, F, F, F, ULL, P, I, P, P, P, P, P],
    "msde_gat_tail_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, F, F, F, ULL, P, I, P, P, P, P, P, P, P, P],
    "msde_escore_mol_saved_floats": [I],
    "msde_escore_mol_fwd": [P, P, P, I, P, P, I, P, P, P, I, I, I, I, I, I, F, F, ULL, P, F, F, P, P, P],
    "msde_escore_mol_score": [P, P, P, P, I, I, P, I, P, P, P, I, I, I, I, I, I, F, F, P, P, P],
    "msde_escore_mol_score_scratch_floats": [I],
    "msde_escore_mol_slab_floats": [],
    "msde_escore_mol_bwd": [P, P, P, I, P, P, I, P, P, P, P, P, I, I, I, I, I, I, F, F, ULL, P, F, F, P, P, P, P, I, P, P],
    "msde_chunk_elems": [],
    "msde_gather_chunks": [P, I, P, P],
    "msde_adam_chunks": [P, P, I, P, P, P, P, P, I, F, F, F, F, F, P],
}


class GemmDesc(ctypes.Structure):
    """msde_gemm_desc of include/msde_hip.h (field order and types must match)."""
    _fields_ = [("A", P), ("A2", P), ("B", P), ("B2", P), ("bias", P), ("bias2", P), ("C", P), ("Z", P), ("R", P),
                ("rowscale", P),
                ("a_gs", LL), ("b_gs", LL), ("bias_gs", LL), ("c_gs", LL), ("r_gs", LL), ("b_kblk_stride", LL),
                ("M", I), ("N", I), ("K1", I), ("K2", I),
                ("lda", I), ("lda2", I), ("ldb", I), ("ldb2", I), ("ldc", I), ("ldz", I), ("ldr", I),
                ("act", I), ("act_lo", I), ("act_hi", I), ("epi", I), ("flags", I), ("groups", I),
                ("b_kblk_log2", I), ("alpha", F)]


class RsDesc(ctypes.Structure):
    """msde_rs_desc of include/msde_hip.h (field order and types must match)."""
    _fields_ = [("A", P), ("A2", P), ("B", P), ("bias", P), ("C", P), ("Z", P), ("R", P), ("Res", P), ("A_out", P),
                ("xf0", P), ("xf1", P), ("xf2", P), ("xf3", P), ("xf4", P), ("stats", P), ("stats_z", P),
                ("stats_mean", P), ("m_valid", P),
                ("M", I), ("N", I), ("K", I),
                ("lda", I), ("lda2", I), ("ldb", I), ("ldc", I), ("ldz", I), ("ldr", I), ("ldres", I), ("lda_out", I),
                ("ld_sz", I),
                ("act", I), ("epi", I), ("flags", I), ("axf", I), ("stats_mode", I), ("rt", I), ("splits", I)]


RS_AXF_NONE, RS_AXF_AFFINE, RS_AXF_BNBWD, RS_AXF_RELU, RS_VEC_STORE = 0, 1, 2, 4, 8
RS_STATS_BNFWD, RS_STATS_BNBWD = 1, 2


class EdgeLayerParams(ctypes.Structure):
    """msde_edge_layer_params of include/msde_hip.h."""
    _fields_ = [(n, P) for n in ("bv", "mW0", "mb0", "mW1", "mb1", "mW2", "mb2", "cW0", "cb0", "cW1", "cb1")]


ACT = {None: 0, "none": 0, "tanh": 1, "silu": 2, "elu": 3, "ssp": 4, "relu": 5, "sspo": 6}
EPI_ACT, EPI_DACT = 0, 1
GEMM_B_KMAJOR, GEMM_ACCUMULATE = 1, 2
REDUCE_LONG = 64          # MSDE_REDUCE_LONG of include/msde_hip.h

_RESTYPE = {"msde_target_arch": ctypes.c_char_p, "msde_linear_bwd_w_workspace_bytes": ctypes.c_longlong,
            "msde_cfconv_fused_bwd_w_workspace_floats": ctypes.c_longlong,
            "msde_embedding_sum_bwd_workspace_floats": ctypes.c_longlong,
            "msde_gin_aggregate_bwd_tab_workspace_floats": ctypes.c_longlong,
            "msde_reduce_slabs_chunks": ctypes.c_longlong, "msde_escore_mol_saved_floats": ctypes.c_longlong,
            "msde_escore_mol_score_scratch_floats": ctypes.c_longlong,
            "msde_escore_mol_slab_floats": ctypes.c_longlong}

_lib = None


class MsdeHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be the one already mapped
    # when our library's NEEDED libamdhip64.so.7 is resolved, or two runtimes end up in one process
    # (symptom: hipErrorNoDevice / hangs).  Importing torch first guarantees a single runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise MsdeHipError(
            f"{LIB_PATH} not found: build it with `python -m moleculesde_amd.build` "
            "(hipcc --offload-arch=gfx950).  moleculesde_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, ctypes.c_int)
    _lib = lib
    return lib


def check(code, name):
    if code != 0:
        kind = "argument error" if code < 0 else "hipError_t"
        raise MsdeHipError(f"{name} failed: {kind} {code}")


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
