"""ctypes binding of libkpgnn_hip.so (include/kpgnn.h).  There is NO fallback: if the library is missing
or a call fails, an exception is raised - the product path never silently runs on the CPU."""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_PKG, "libkpgnn_hip.so")

MODE_GIN, MODE_GINPLUS, MODE_GCN, MODE_SUM = 0, 1, 2, 3

c_i32, c_i64, c_f32p, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p


MATH_AUTO, MATH_F32 = 0, 1     # include/kpgnn.h KPGNN_MATH_*
DENSE_MATH = MATH_AUTO          # what new dense descriptors ask for (ops_dense.set_dense_math)


class KpgnnError(RuntimeError):
    pass


class AggFwdDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i32), ("K", c_i32), ("D", c_i32), ("K_csr", c_i32), ("mode", c_i32),
        ("n_code0", c_i32), ("n_codek", c_i32), ("use_tables", c_i32),
        ("rowptr", c_vp), ("col", c_vp), ("code", c_vp), ("dis", c_vp),
        ("x", c_vp), ("x_sn", c_i64), ("x_sk", c_i64),
        ("table0", c_vp), ("tablek", c_vp),
        ("periph", c_vp), ("p_sn", c_i64), ("p_sk", c_i64),
        ("eps", c_vp),
        ("out", c_vp), ("o_sn", c_i64), ("o_sk", c_i64),
        ("pre", c_vp), ("theta", c_vp), ("hout", c_vp), ("xbias", c_vp),
        ("ptab", c_vp), ("uid", c_vp), ("uid_stride", c_i64),
        ("x_slot", c_vp * 16), ("n_dict", c_i32), ("alphas", c_vp), ("storage", c_i32),
        ("n_dyn", c_vp),
        ("graph_ptr", c_vp), ("num_graphs", c_i32), ("max_graph_nodes", c_i32), ("hinit", c_vp), ("hinit2", c_vp),
    ]


class AggBwdDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i32), ("K", c_i32), ("D", c_i32), ("K_csr", c_i32), ("mode", c_i32),
        ("n_code0", c_i32), ("n_codek", c_i32), ("use_tables", c_i32),
        ("rowptr_src", c_vp), ("col_src", c_vp), ("code_src", c_vp), ("dis", c_vp),
        ("g", c_vp), ("g_sn", c_i64), ("g_sk", c_i64),
        ("eps", c_vp),
        ("gx", c_vp), ("gx_sn", c_i64), ("gx_sk", c_i64),
        ("gtable0", c_vp), ("gtablek", c_vp),
        ("gx_slot", c_vp * 16), ("accumulate_mask", ctypes.c_uint32), ("storage", c_i32),
        ("n_dyn", c_vp),
    ]


class TableGradDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i32), ("K", c_i32), ("D", c_i32), ("nodes_per_tile", c_i32), ("n_code0", c_i32), ("n_codek", c_i32),
        ("n_dict", c_i32), ("dict_src", c_i32),
        ("tile_ptr", c_vp), ("tile_pack", c_vp),
        ("g", c_vp), ("g_sn", c_i64), ("g_sk", c_i64),
        ("uid", c_vp), ("uid_stride", c_i64), ("theta", c_vp), ("gh", c_vp),
        ("gtable0", c_vp), ("gtablek", c_vp), ("gdict", c_vp),
        ("workspace", c_vp), ("workspace_bytes", ctypes.c_size_t),
        ("kernel", c_i32),
        ("dict_pack", c_vp), ("dict_pack_K", c_i32),
        ("extra_slab", c_vp), ("extra_nslab", c_i32), ("extra_elems", c_i64), ("extra_out", c_vp), ("storage", c_i32),
        ("fuse_pre", c_vp), ("fuse_ptab", c_vp), ("fuse_uid", c_vp), ("fuse_uid_stride", c_i64), ("fuse_n_dict", c_i32),
        ("fuse_g", c_vp), ("fuse_gtheta", c_vp), ("fuse_alphas", c_vp), ("fuse_galphas", c_vp),
        ("fuse_workspace", c_vp), ("fuse_workspace_bytes", ctypes.c_size_t), ("accumulate_dict", c_i32),
        ("pending", c_vp),
        ("n_dyn", c_vp),
        ("max_multiplicity", c_i32),
    ]


class DictGradDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i32), ("K", c_i32), ("D", c_i32), ("n_dict", c_i32),
        ("uid", c_vp), ("uid_stride", c_i64), ("theta", c_vp), ("gh", c_vp), ("gdict", c_vp),
        ("workspace", c_vp), ("workspace_bytes", ctypes.c_size_t), ("defer_reduce", c_i32), ("dominant", c_vp),
        ("n_dyn", c_vp),
    ]


class DictGradMultiDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i32), ("D", c_i32), ("n_dict", c_i32), ("L", c_i32),
        ("uid", c_vp), ("uid_stride", c_i64),
        ("theta", c_vp * 16), ("gh", c_vp * 16), ("K", c_i32 * 16),
        ("gdict", c_vp), ("workspace", c_vp), ("workspace_bytes", ctypes.c_size_t), ("dominant", c_vp), ("n_dyn", c_vp),
    ]


class CombineBwdDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i32), ("K", c_i32), ("D", c_i32), ("mode", c_i32),
        ("pre", c_vp), ("gh", c_vp), ("theta", c_vp),
        ("gout", c_vp), ("go_sn", c_i64), ("go_sk", c_i64),
        ("periph", c_vp), ("p_sn", c_i64), ("p_sk", c_i64),
        ("ptab", c_vp), ("uid", c_vp), ("uid_stride", c_i64),
        ("g", c_vp), ("gv", c_vp), ("gtheta", c_vp),
        ("workspace", c_vp), ("workspace_bytes", ctypes.c_size_t), ("n_dict", c_i32),
        ("alphas", c_vp), ("galphas", c_vp), ("storage", c_i32),
    ]


class BnDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i64), ("C", c_i32), ("relu", c_i32), ("eps", ctypes.c_float), ("momentum", ctypes.c_float),
        ("x", c_vp), ("x_stride", c_i64), ("gamma", c_vp), ("beta", c_vp),
        ("running_mean", c_vp), ("running_var", c_vp), ("mean", c_vp), ("invstd", c_vp),
        ("z", c_vp), ("z_stride", c_i64), ("residual", c_vp), ("r_stride", c_i64),
        ("stat_slot", c_vp), ("stats_ready", c_i32), ("out_slot", c_vp), ("num_batches_tracked", c_vp),
        ("outer_gamma", c_vp), ("outer_beta", c_vp), ("outer_eps", ctypes.c_float), ("outer_momentum", ctypes.c_float),
        ("outer_running_mean", c_vp), ("outer_running_var", c_vp), ("outer_num_batches_tracked", c_vp),
        ("outer_mean", c_vp), ("outer_invstd", c_vp),
        ("n_dyn", c_vp),
    ]


class BnBwdDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i64), ("C", c_i32), ("relu", c_i32),
        ("x", c_vp), ("x_stride", c_i64), ("dz", c_vp), ("dz_stride", c_i64),
        ("gamma", c_vp), ("beta", c_vp), ("mean", c_vp), ("invstd", c_vp),
        ("dx", c_vp), ("dx_stride", c_i64), ("dgamma", c_vp), ("dbeta", c_vp),
        ("stat_slot", c_vp), ("reduce_only", c_i32), ("residual_grad", c_vp), ("rg_stride", c_i64),
        ("outer_mean", c_vp), ("outer_invstd", c_vp),
        ("n_dyn", c_vp),
    ]


class ReduceJob(ctypes.Structure):
    _fields_ = [("slab", c_vp), ("nslab", c_i32), ("elems", c_i64), ("out", c_vp * 4), ("n_out", c_i64 * 4)]


class WgradDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i64), ("O", c_i32), ("I", c_i32),
        ("dy", c_vp), ("dy_stride", c_i64), ("x", c_vp), ("x_stride", c_i64),
        ("dw", c_vp), ("db", c_vp), ("workspace", c_vp), ("workspace_bytes", ctypes.c_size_t),
        ("x_mean", c_vp), ("x_invstd", c_vp), ("x_gamma", c_vp), ("x_beta", c_vp), ("x_relu", c_i32),
        ("defer", c_vp), ("dy_mask", c_vp), ("n_dyn", c_vp), ("math", c_i32),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.math = DENSE_MATH


class LinearBnDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i64), ("O", c_i32), ("I", c_i32),
        ("x", c_vp), ("w", c_vp), ("bias", c_vp), ("y", c_vp),
        ("w_transposed", c_i32), ("pro", c_i32), ("epi", c_i32), ("pro_relu", c_i32),
        ("in_slot", c_vp), ("in_gamma", c_vp), ("in_beta", c_vp),
        ("in_eps", ctypes.c_float), ("momentum", ctypes.c_float),
        ("in_mean", c_vp), ("in_invstd", c_vp),
        ("running_mean", c_vp), ("running_var", c_vp), ("num_batches_tracked", c_vp),
        ("x2", c_vp), ("xt", c_vp), ("dgamma", c_vp), ("dbeta", c_vp),
        ("out_slot", c_vp),
        ("e_x", c_vp), ("e_mean", c_vp), ("e_invstd", c_vp), ("e_gamma", c_vp), ("e_beta", c_vp),
        ("o_mean", c_vp), ("o_invstd", c_vp), ("o_gamma", c_vp), ("o_dgamma", c_vp), ("o_dbeta", c_vp),
        ("n_dyn", c_vp), ("math", c_i32), ("workspace", c_vp), ("workspace_bytes", ctypes.c_size_t),
        ("w_split_ready", c_i32),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.math = DENSE_MATH


class SplitJob(ctypes.Structure):
    _fields_ = [("w", c_vp), ("wn", c_i64), ("wk", c_i64), ("O", c_i32), ("I", c_i32), ("frag", c_vp)]


class AttnDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i32), ("K", c_i32), ("D", c_i32),
        ("x", c_vp), ("x_sn", c_i64), ("x_sk", c_i64),
        ("gin", c_vp), ("whh", c_vp), ("acts", c_vp), ("hsum", c_vp), ("w", c_vp), ("out", c_vp),
        ("gout", c_vp), ("dx", c_vp), ("ds", c_vp), ("dgin", c_vp), ("hprev", c_vp),
    ]


class AttnScanDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i32), ("K", c_i32), ("D", c_i32),
        ("x", c_vp), ("x_sn", c_i64), ("x_sk", c_i64),
        ("w_ih", c_vp * 2), ("w_hh", c_vp * 2), ("b_ih", c_vp * 2), ("b_hh", c_vp * 2),
        ("acts", c_vp), ("hsum", c_vp), ("w", c_vp), ("out", c_vp), ("w_pad", c_vp),
        ("gout", c_vp), ("dx", c_vp), ("ds", c_vp), ("dgin", c_vp), ("whh_slab", c_vp), ("dwhh_pad", c_vp),
    ]


class LinearDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i64), ("O", c_i32), ("I", c_i32),
        ("x", c_vp), ("x_stride", c_i64), ("w", c_vp), ("bias", c_vp), ("y", c_vp), ("y_stride", c_i64),
        ("w_transposed", c_i32), ("y_block_cols", c_i32), ("y_block_stride", c_i64), ("x_mask", c_vp), ("n_dyn", c_vp),
        ("math", c_i32), ("workspace", c_vp), ("workspace_bytes", ctypes.c_size_t),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.math = DENSE_MATH


class LinearGroupDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i64), ("O", c_i32), ("I", c_i32), ("group", c_i32),
        ("x", c_vp * 16), ("x_stride", c_i64), ("w", c_vp), ("bias", c_vp), ("y", c_vp), ("relu", c_i32), ("n_dyn", c_vp),
        ("math", c_i32), ("workspace", c_vp), ("workspace_bytes", ctypes.c_size_t),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.math = DENSE_MATH


class HopMlpDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i64), ("K", c_i32), ("DI", c_i32), ("DO", c_i32), ("H", c_i32),
        ("s", c_vp), ("w1", c_vp), ("b1", c_vp), ("w2", c_vp), ("b2", c_vp), ("theta", c_vp),
        ("wc", c_vp), ("bc", c_vp), ("h1", c_vp), ("h2", c_vp), ("out", c_vp), ("gout", c_vp), ("gs", c_vp), ("gflat", c_vp),
        ("workspace", c_vp), ("workspace_bytes", ctypes.c_size_t),
    ]


class TgsDesc(ctypes.Structure):
    _fields_ = [
        ("M", c_i64), ("C", c_i32), ("D", c_i32), ("R", c_i32),
        ("idx", c_vp), ("col_offset", c_vp), ("table", c_vp), ("bias", c_vp),
        ("out", c_vp), ("out_stride", c_i64), ("gout", c_vp), ("gout_stride", c_i64), ("gtable", c_vp),
        ("workspace", c_vp), ("workspace_bytes", ctypes.c_size_t),
        ("n_dyn", c_vp),
    ]


class EncTablesDesc(ctypes.Structure):
    _fields_ = [
        ("H", c_i32), ("num_components", c_i32), ("num_encoders", c_i32),
        ("comp_emb", c_vp * 16), ("comp_rows", c_i32 * 16), ("comp_encoder", c_i32 * 16),
        ("enc_w", c_vp * 4), ("enc_b", c_vp * 4), ("enc_gate", c_vp * 4), ("enc_mult", ctypes.c_float * 4),
        ("enc_squash", c_i32 * 4),
        ("table", c_vp), ("pre", c_vp), ("bias", c_vp), ("gtable", c_vp), ("gbias", c_vp),
        ("comp_gemb", c_vp * 16), ("enc_gw", c_vp * 4), ("enc_gb", c_vp * 4), ("enc_ggate", c_vp * 4),
    ]


class PoolDesc(ctypes.Structure):
    _fields_ = [
        ("N", c_i64), ("G", c_i32), ("D", c_i32), ("mode", c_i32),
        ("graph_ptr", c_vp), ("batch", c_vp), ("x", c_vp), ("x_stride", c_i64), ("out", c_vp),
        ("gout", c_vp), ("gx", c_vp), ("gx_stride", c_i64),
        ("n_dyn", c_vp),
    ]


class DatasetView(ctypes.Structure):
    _fields_ = [
        ("K", c_i32), ("G", c_i32), ("node_ptr", c_vp), ("pair_ptr", c_vp), ("ent_ptr", c_vp),
        ("rowptr_dst", c_vp), ("rowptr_src", c_vp), ("col_dst", c_vp), ("col_src", c_vp),
        ("code_dst", c_vp), ("code_src", c_vp), ("ent_rel", c_vp), ("ent", c_vp),
    ]


class RowGather(ctypes.Structure):
    _fields_ = [("src", c_vp), ("dst", c_vp), ("row_bytes", c_i32)]


class CollateDesc(ctypes.Structure):
    _fields_ = [
        ("ds", DatasetView), ("B", c_i32), ("N", c_i32), ("A", c_i64), ("n_ent", c_i64), ("hdr", c_vp),
        ("rowptr_dst", c_vp), ("col_dst", c_vp), ("code_dst", c_vp),
        ("rowptr_src", c_vp), ("col_src", c_vp), ("code_src", c_vp),
        ("batch", c_vp), ("node_src", c_vp),
        ("nodes_per_tile", c_i32), ("tile_ptr", c_vp), ("tile_pack", c_vp), ("ent_node_ptr", c_vp),
        ("num_prefix", c_i32), ("prefix_ptr", c_vp), ("prefix_pack", c_vp), ("prefix_scratch", c_vp),
        ("n_node_rows", c_i32), ("node_rows", RowGather * 8),
        ("n_graph_rows", c_i32), ("graph_rows", RowGather * 8),
    ]


# name -> (restype, argtypes); every symbol include/kpgnn.h declares
SIGNATURES = {
    "kpgnn_abi_version": (ctypes.c_int, []),
    "kpgnn_last_error": (ctypes.c_char_p, []),
    "kpgnn_device_info": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)] * 3 + [ctypes.c_char_p, ctypes.c_int]),
    "kpgnn_csr_stats": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp]),
    "kpgnn_csr_workspace_bytes": (ctypes.c_size_t, [c_i64, c_i64, c_i64, c_i32]),
    "kpgnn_csr_build": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i64, c_i64,
                                       c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp,
                                       c_vp, ctypes.c_size_t, c_vp]),
    "kpgnn_aggregate_fwd": (ctypes.c_int, [ctypes.POINTER(AggFwdDesc), c_vp]),
    "kpgnn_aggregate_bwd": (ctypes.c_int, [ctypes.POINTER(AggBwdDesc), c_vp]),
    "kpgnn_table_grad_workspace_bytes": (ctypes.c_size_t, [c_i32] * 7),
    "kpgnn_table_grad": (ctypes.c_int, [ctypes.POINTER(TableGradDesc), c_vp]),
    "kpgnn_table_grad_fuse_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32]),
    "kpgnn_dict_grad_workspace_bytes": (ctypes.c_size_t, [c_i32] * 4),
    "kpgnn_dict_grad": (ctypes.c_int, [ctypes.POINTER(DictGradDesc), c_vp]),
    "kpgnn_dict_grad_multi": (ctypes.c_int, [ctypes.POINTER(DictGradMultiDesc), c_vp]),
    "kpgnn_dict_grad_slabs": (c_i32, [c_i32]),
    "kpgnn_tile_pack_filter": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "kpgnn_tile_pack_prefixes": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "kpgnn_collate": (ctypes.c_int, [ctypes.POINTER(CollateDesc), c_vp]),
    "kpgnn_regression_loss": (ctypes.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "kpgnn_score_head_fwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "kpgnn_score_head_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "kpgnn_adam_step": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_double, c_vp]),
    "kpgnn_reduce_jobs": (ctypes.c_int, [c_vp, c_i32, c_vp]),
    "kpgnn_adam_step_device": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, ctypes.c_double, ctypes.c_double,
                                              ctypes.c_double, ctypes.c_double, ctypes.c_double, c_vp]),
    "kpgnn_multi_copy": (ctypes.c_int, [c_i32, c_vp, c_vp, c_vp, c_vp]),
    "kpgnn_dict_tile_pack": (ctypes.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "kpgnn_combine_bwd_workspace_bytes": (ctypes.c_size_t, [c_i32] * 3),
    "kpgnn_combine_bwd": (ctypes.c_int, [ctypes.POINTER(CombineBwdDesc), c_vp]),
    "kpgnn_bn_fwd": (ctypes.c_int, [ctypes.POINTER(BnDesc), c_vp]),
    "kpgnn_bn_bwd": (ctypes.c_int, [ctypes.POINTER(BnBwdDesc), c_vp]),
    "kpgnn_wgrad_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32]),
    "kpgnn_linear_wgrad": (ctypes.c_int, [ctypes.POINTER(WgradDesc), c_vp]),
    "kpgnn_linear_wgrad_pair": (ctypes.c_int, [ctypes.POINTER(WgradDesc), ctypes.POINTER(WgradDesc), c_vp]),
    "kpgnn_wgrad_group_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32, c_i32]),
    "kpgnn_linear_wgrad_group": (ctypes.c_int, [ctypes.POINTER(WgradDesc), c_vp, c_i32, c_vp]),
    "kpgnn_linear_bn": (ctypes.c_int, [ctypes.POINTER(LinearBnDesc), c_vp]),
    "kpgnn_enc_tables_fwd": (ctypes.c_int, [ctypes.POINTER(EncTablesDesc), c_vp]),
    "kpgnn_enc_tables_bwd": (ctypes.c_int, [ctypes.POINTER(EncTablesDesc), c_vp]),
    "kpgnn_segment_pool_fwd": (ctypes.c_int, [ctypes.POINTER(PoolDesc), c_vp]),
    "kpgnn_segment_pool_bwd": (ctypes.c_int, [ctypes.POINTER(PoolDesc), c_vp]),
    "kpgnn_stat_slot_bytes": (ctypes.c_size_t, [c_i32]),
    "kpgnn_stream_capture_id": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint64)]),
    "kpgnn_attn_fwd": (ctypes.c_int, [ctypes.POINTER(AttnDesc), c_vp]),
    "kpgnn_attn_bwd": (ctypes.c_int, [ctypes.POINTER(AttnDesc), c_vp]),
    "kpgnn_attn_scan_fwd": (ctypes.c_int, [ctypes.POINTER(AttnScanDesc), c_vp]),
    "kpgnn_attn_scan_bwd": (ctypes.c_int, [ctypes.POINTER(AttnScanDesc), c_vp]),
    "kpgnn_attn_scan_unpad": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "kpgnn_linear_fwd": (ctypes.c_int, [ctypes.POINTER(LinearDesc), c_vp]),
    "kpgnn_linear_split_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32, c_i32]),
    "kpgnn_linear_split_many": (ctypes.c_int, [ctypes.POINTER(SplitJob), c_i32, c_vp]),
    "kpgnn_linear_group_fwd": (ctypes.c_int, [ctypes.POINTER(LinearGroupDesc), c_vp]),
    "kpgnn_geo_theta_fwd": (ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "kpgnn_geo_theta_bwd": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "kpgnn_hop_mlp_workspace_bytes": (ctypes.c_size_t, [c_i64, c_i32, c_i32, c_i32, c_i32]),
    "kpgnn_hop_mlp_fwd": (ctypes.c_int, [ctypes.POINTER(HopMlpDesc), c_vp]),
    "kpgnn_hop_mlp_bwd": (ctypes.c_int, [ctypes.POINTER(HopMlpDesc), c_vp]),
    "kpgnn_table_gather_sum_fwd": (ctypes.c_int, [ctypes.POINTER(TgsDesc), c_vp]),
    "kpgnn_table_gather_sum_bwd": (ctypes.c_int, [ctypes.POINTER(TgsDesc), c_vp]),
    "kpgnn_table_gather_sum_bwd_workspace_bytes": (ctypes.c_size_t, [c_i64, c_i32, c_i32]),
}

_lib = None


def load(path=None):
    """Load (once) and return the ctypes handle; raises KpgnnError when the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or HIP_LIB_PATH
    if not os.path.exists(path):
        raise KpgnnError(
            f"{path} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m kp_gnn_amd.build` (or __graft_entry__.build()).")
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise KpgnnError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.kpgnn_abi_version() != 1:
        raise KpgnnError("libkpgnn_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().kpgnn_last_error()
        raise KpgnnError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
