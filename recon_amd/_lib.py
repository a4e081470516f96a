"""ctypes binding of librecon_hip.so (the C ABI declared in include/recon_hip.h).

The library is built in-tree by `make -C recon_amd/csrc` (or `__graft_entry__.build()`).  There is
NO fallback: if the shared object is missing or an entry point returns an error, a RuntimeError is
raised.  The product path never computes on the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RECON_HIP_LIB: another build of the library (A/B of two builds in one gpurun call: tools/ab_builds.sh); default: the in-tree one
LIB_PATH = os.environ.get("RECON_HIP_LIB") or os.path.join(_HERE, "csrc", "librecon_hip.so")

ERRORS = {0: "ok", -1: "invalid argument", -2: "unsupported shape", -3: "kernel launch failed",
          -4: "workspace too small"}

c_f32p = C.c_void_p
c_i32p = C.c_void_p
c_i64p = C.c_void_p


class ReconGraph(C.Structure):
    _fields_ = [("N", C.c_int32), ("E", C.c_int32),
                ("rowptr_dst", c_i32p), ("eid", c_i32p), ("src", c_i32p), ("dst", c_i32p),
                ("rowptr_src", c_i32p), ("slot_by_src", c_i32p),
                ("hub_chunk", C.c_int32), ("n_hub", C.c_int32), ("n_piece", C.c_int32),
                ("hub_node", c_i32p), ("hub_ptr", c_i32p), ("piece", c_i32p),
                ("n_hub_src", C.c_int32), ("n_piece_src", C.c_int32), ("hub_node_src", c_i32p), ("hub_ptr_src", c_i32p), ("piece_src", c_i32p),
                ("hub_ws", c_f32p), ("hub_ws_floats", C.c_int64),
                ("n_rows", C.c_int32), ("row_node", c_i32p), ("rowptr_rows", c_i32p), ("node_row", c_i32p)]


class GatFwdArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("E", C.c_int32), ("F", C.c_int32), ("R", C.c_int32), ("D", C.c_int32),
                ("H", C.c_int32), ("concat", C.c_int32), ("alpha", C.c_float),
                ("x", c_f32p), ("edge_embed", c_f32p), ("a", c_f32p), ("a_2", c_f32p), ("keep", c_f32p),
                ("P", c_f32p), ("Q", c_f32p), ("sigma", c_f32p), ("Z", c_f32p), ("out", c_f32p),
                ("ld_out", C.c_int32)]


class GatBwdArgs(C.Structure):
    _fields_ = [("fwd", GatFwdArgs), ("grad_out", c_f32p), ("ld_gout", C.c_int32),
                ("Gm", c_f32p), ("gP", c_f32p), ("partial", c_f32p),
                ("g_x", c_f32p), ("g_edge_embed", c_f32p), ("g_a", c_f32p), ("g_a_2", c_f32p)]


class GatAtpArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("E", C.c_int32), ("F", C.c_int32), ("R", C.c_int32), ("D", C.c_int32),
                ("H", C.c_int32), ("concat", C.c_int32), ("alpha", C.c_float),
                ("x", c_f32p), ("edge_embed", c_f32p), ("a", c_f32p), ("a_2", c_f32p), ("keep", c_f32p),
                ("u", c_f32p), ("c_node", c_f32p), ("c_rel", c_f32p), ("V", c_f32p), ("sigma", c_f32p),
                ("Z", c_f32p), ("Zk", c_f32p), ("out", c_f32p), ("ld_out", C.c_int32), ("a_split", C.c_void_p),
                ("split_mode", C.c_int32), ("keep_max", C.c_float), ("aux", C.c_void_p), ("ee_index", c_i32p), ("ee_rows", C.c_int32),
                ("io_bf16", C.c_int32)]


class GatAtpBwdArgs(C.Structure):
    _fields_ = [("fwd", GatAtpArgs), ("grad_out", c_f32p), ("ld_gout", C.c_int32), ("g_h", c_f32p), ("g_V", c_f32p),
                ("g_sigma", c_f32p), ("Gxs", c_f32p), ("gxd", c_f32p), ("Gs", c_f32p), ("g_u", c_f32p), ("q", c_f32p),
                ("partial", c_f32p), ("partial2", c_f32p), ("g_x", c_f32p), ("g_edge_embed", c_f32p), ("g_a", c_f32p), ("g_a_2", c_f32p),
                ("gh_split", C.c_void_p), ("g_ee_bf16", C.c_int32)]


class PropArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32), ("S", C.c_int32), ("L", C.c_int32), ("dd", C.c_int32),
                ("act", C.c_int32), ("adj", C.POINTER(C.c_void_p)), ("h0", c_f32p), ("h0_batch_stride", C.c_int64),
                ("head_idx", c_i64p), ("tail_idx", c_i64p), ("idx_batch_stride", C.c_int64),
                ("out", c_f32p), ("h_saved", c_f32p), ("trans", C.POINTER(C.c_void_p)), ("identity", c_f32p), ("stats", c_f32p),
                ("split_ws", C.c_void_p), ("split_ws_bytes", C.c_int64)]


class PropBwdArgs(C.Structure):
    _fields_ = [("fwd", PropArgs), ("grad_out", c_f32p), ("g_adj", C.POINTER(C.c_void_p)), ("g_h", c_f32p),
                ("g_trans", C.POINTER(C.c_void_p)), ("g_identity", c_f32p), ("identity_ws", c_f32p), ("wide_ws", c_f32p),
                ("head_blk", C.c_void_p), ("tail_blk", C.c_void_p), ("chain_ws", c_f32p)]


class PropB16Args(C.Structure):
    _fields_ = [("B", C.c_int32), ("C", C.c_int32), ("S", C.c_int32), ("L", C.c_int32), ("dd", C.c_int32),
                ("act", C.c_int32), ("adj", C.POINTER(C.c_void_p)), ("h0", C.c_void_p), ("h0_batch_stride", C.c_int64),
                ("head_idx", c_i64p), ("tail_idx", c_i64p), ("idx_batch_stride", C.c_int64),
                ("out", C.c_void_p), ("h_saved", C.c_void_p), ("trans", C.POINTER(C.c_void_p)), ("identity", C.c_void_p),
                ("zeros", C.c_void_p)]


class PropB16BwdArgs(C.Structure):
    _fields_ = [("fwd", PropB16Args), ("grad_out", C.c_void_p), ("g_adj", C.POINTER(C.c_void_p)), ("g_h", C.c_void_p),
                ("ws", C.c_void_p), ("head_blk", C.c_void_p), ("tail_blk", C.c_void_p), ("g_trans", C.POINTER(C.c_void_p)),
                ("g_identity", C.c_void_p), ("diag_ws", C.c_void_p), ("ident_ws", c_f32p)]


class ReconKG(C.Structure):
    _fields_ = [("num_entities", C.c_int64), ("pair_ptr", C.c_void_p), ("pair_tgt", C.c_void_p), ("pair_first_rel", C.c_void_p),
                ("not_loop", C.c_void_p), ("rel_ptr", C.c_void_p), ("rel_sorted", C.c_void_p)]


class GcnArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("n", C.c_int32), ("in_features", C.c_int32), ("out_features", C.c_int32),
                ("x", c_f32p), ("adj", c_f32p), ("weight", c_f32p), ("bias", c_f32p), ("support", c_f32p),
                ("out", c_f32p), ("w_split", C.c_void_p)]


class GcnBwdArgs(C.Structure):
    _fields_ = [("fwd", GcnArgs), ("grad_out", c_f32p), ("g_support", c_f32p), ("partial", c_f32p),
                ("g_x", c_f32p), ("g_adj", c_f32p), ("g_weight", c_f32p), ("g_bias", c_f32p), ("gs_split", C.c_void_p)]


class GcnB16Args(C.Structure):
    _fields_ = [("B", C.c_int32), ("n", C.c_int32), ("in_features", C.c_int32), ("out_features", C.c_int32),
                ("x", C.c_void_p), ("ldx", C.c_int64), ("adj", C.c_void_p), ("weight", C.c_void_p), ("bias", C.c_void_p),
                ("support", C.c_void_p), ("lds", C.c_int64), ("out", C.c_void_p), ("ldo", C.c_int64), ("w_planes", C.c_void_p),
                ("w_planes_valid", C.c_int32), ("node_ptr", C.c_void_p), ("adj_ptr", C.c_void_p), ("total_rows", C.c_int64)]


class GcnB16BwdArgs(C.Structure):
    _fields_ = [("fwd", GcnB16Args), ("grad_out", C.c_void_p), ("ldg", C.c_int64), ("g_support", C.c_void_p), ("partial", c_f32p),
                ("g_x", C.c_void_p), ("ldgx", C.c_int64), ("g_adj", C.c_void_p), ("g_weight", C.c_void_p), ("g_bias", C.c_void_p),
                ("zeros", C.c_void_p)]


class GcnB16StackArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("n", C.c_int32), ("in_features", C.c_int32), ("hidden", C.c_int32), ("L", C.c_int32),
                ("x", C.c_void_p), ("ldx", C.c_int64), ("adj", C.c_void_p), ("w_planes", C.POINTER(C.c_void_p)), ("bias", C.POINTER(C.c_void_p)),
                ("out", C.c_void_p), ("ldo", C.c_int64)]


class GcnB16StackTrainArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("n", C.c_int32), ("in_features", C.c_int32), ("hidden", C.c_int32), ("L", C.c_int32),
                ("x", C.c_void_p), ("ldx", C.c_int64), ("x_rows", C.c_void_p), ("ldxr", C.c_int64), ("adj", C.c_void_p),
                ("weight", C.POINTER(C.c_void_p)), ("bias", C.POINTER(C.c_void_p)),
                ("planes", C.POINTER(C.c_void_p)), ("acts", C.POINTER(C.c_void_p)), ("ldo", C.c_int64),
                ("grad_out", C.c_void_p), ("ldg", C.c_int64), ("g_support", C.POINTER(C.c_void_p)), ("partial", C.c_void_p),
                ("g_x", C.c_void_p), ("ldgx", C.c_int64), ("g_weight", C.POINTER(C.c_void_p)), ("g_bias", C.POINTER(C.c_void_p)),
                ("zeros", C.c_void_p)]


ACT = {"linear": 0, "relu": 1, "tanh": 2}

# every symbol include/recon_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("recon_version", C.c_int, []),
    ("recon_error_string", C.c_char_p, [C.c_int]),
    ("recon_graph_workspace_bytes", C.c_size_t, [C.c_int32, C.c_int32]),
    ("recon_graph_build", C.c_int, [c_i64p, c_i64p, C.POINTER(ReconGraph), C.c_void_p, C.c_size_t, C.c_void_p]),
    ("recon_graph_hubs_count", C.c_int, [C.POINTER(ReconGraph), C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    ("recon_graph_build_checked", C.c_int, [c_i64p, c_i64p, C.POINTER(ReconGraph), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    ("recon_graph_hubs_count_checked", C.c_int, [C.POINTER(ReconGraph), C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    ("recon_graph_build_counted", C.c_int, [c_i64p, c_i64p, C.POINTER(ReconGraph), C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_void_p]),
    ("recon_graph_hubs_read", C.c_int, [C.POINTER(ReconGraph), C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    ("recon_graph_hubs_fill", C.c_int, [C.POINTER(ReconGraph), C.c_void_p]),
    ("recon_graph_counts_read", C.c_int, [C.POINTER(ReconGraph), C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    ("recon_graph_rows_compact", C.c_int, [C.POINTER(ReconGraph), C.c_void_p]),
    ("recon_edges_prune_batch", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("recon_gat_atp_bwd_gee_bf16_supported", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("recon_slot_index", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("recon_edges_prune", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("recon_rows_expand", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    ("recon_graph_hub_ws_floats", C.c_size_t, [C.POINTER(ReconGraph), C.c_int32, C.c_int32, C.c_int32]),
    ("recon_spmm_rowsum_workspace_floats", C.c_size_t, [C.c_int32, C.c_int32]),
    ("recon_spmm_rowsum_fwd", C.c_int, [C.POINTER(ReconGraph), c_f32p, C.c_int32, c_f32p, c_f32p, C.c_void_p]),
    ("recon_spmm_rowsum_mod_fwd", C.c_int, [C.POINTER(ReconGraph), c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, C.c_void_p]),
    ("recon_gather_rows_pair_fwd", C.c_int, [c_f32p, c_i64p, C.c_int64, C.c_int32, c_f32p, C.c_void_p]),
    ("recon_spmm_rowsum_bwd", C.c_int, [c_i64p, C.c_int64, c_f32p, C.c_int32, c_f32p, C.c_void_p]),
    ("recon_gat_fwd", C.c_int, [C.POINTER(ReconGraph), C.POINTER(GatFwdArgs), C.c_void_p]),
    ("recon_gat_project", C.c_int, [C.POINTER(ReconGraph), C.POINTER(GatFwdArgs), C.c_void_p]),
    ("recon_gat_edge_fwd", C.c_int, [C.POINTER(ReconGraph), C.POINTER(GatFwdArgs), C.c_void_p]),
    ("recon_gat_bwd_partial_floats", C.c_size_t, [C.c_int32] * 6),
    ("recon_gat_bwd", C.c_int, [C.POINTER(ReconGraph), C.POINTER(GatBwdArgs), C.c_void_p]),
    ("recon_gat_atp_supported", C.c_int, [C.c_int32] * 6),
    ("recon_gat_atp_f16x2_supported", C.c_int, [C.c_int32] * 4),
    ("recon_gat_atp_fwd", C.c_int, [C.POINTER(ReconGraph), C.POINTER(GatAtpArgs), C.c_void_p]),
    ("recon_gat_atp_scores", C.c_int, [C.POINTER(ReconGraph), C.POINTER(GatAtpArgs), C.c_void_p]),
    ("recon_gat_atp_aggregate", C.c_int, [C.POINTER(ReconGraph), C.POINTER(GatAtpArgs), C.c_void_p]),
    ("recon_gat_atp_project", C.c_int, [C.POINTER(ReconGraph), C.POINTER(GatAtpArgs), C.c_void_p]),
    ("recon_gat_atp_bwd_partial_floats", C.c_size_t, [C.c_int32] * 6),
    ("recon_gat_atp_bwd_partial2_floats", C.c_size_t, [C.c_int32] * 6),
    ("recon_gat_atp_bwd", C.c_int, [C.POINTER(ReconGraph), C.POINTER(GatAtpBwdArgs), C.c_void_p]),
    ("recon_gat_atp_bwd_phase", C.c_int, [C.POINTER(ReconGraph), C.POINTER(GatAtpBwdArgs), C.c_int32, C.c_void_p]),
    ("recon_block_adjacency_fwd", C.c_int, [c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_void_p]),
    ("recon_block_adjacency_bwd_workspace_floats", C.c_size_t, [C.c_int32] * 3),
    ("recon_block_adjacency_bwd", C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    ("recon_propagate_fwd", C.c_int, [C.POINTER(PropArgs), C.c_void_p]),
    ("recon_propagate_bwd", C.c_int, [C.POINTER(PropBwdArgs), C.c_void_p]),
    ("recon_propagate_form", C.c_int, [C.POINTER(PropArgs)]),
    ("recon_propagate_identity_ws_floats", C.c_size_t, [C.c_int32]),
    ("recon_propagate_ws_bytes", C.c_size_t, [C.POINTER(PropArgs)]),
    ("recon_propagate_bwd_ws_floats", C.c_size_t, [C.POINTER(PropArgs)]),
    ("recon_propagate_bwd_chain_ws_floats", C.c_size_t, [C.POINTER(PropArgs)]),
    ("recon_propagate_b16_form", C.c_int, [C.POINTER(PropB16Args)]),
    ("recon_propagate_b16_fwd", C.c_int, [C.POINTER(PropB16Args), C.c_void_p]),
    ("recon_propagate_b16_bwd", C.c_int, [C.POINTER(PropB16BwdArgs), C.c_void_p]),
    ("recon_propagate_b16_bwd_diag_elems", C.c_size_t, [C.POINTER(PropB16Args)]),
    ("recon_block_adjacency_b16_fwd", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    ("recon_block_adjacency_b16_bwd_workspace_floats", C.c_size_t, [C.c_int32]),
    ("recon_block_adjacency_b16_bwd", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, c_f32p, C.c_void_p]),
    ("recon_rel_rows_mm", C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p] * 2),
    ("recon_rel_rows_mm_wgrad", C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_void_p] * 2),
    ("recon_start_entity_embeddings", C.c_int, [c_f32p, c_i64p, c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p,
                                                C.c_void_p]),
    ("recon_start_entity_embeddings_bwd", C.c_int, [c_f32p, c_i64p, c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_void_p, C.c_void_p]),
    ("recon_gcn_split_bytes", C.c_size_t, [C.c_int32, C.c_int32]),
    ("recon_gcn_fwd", C.c_int, [C.POINTER(GcnArgs), C.c_void_p]),
    ("recon_gcn_bwd_partial_floats", C.c_size_t, [C.c_int32] * 4),
    ("recon_gcn_bwd_split_bytes", C.c_size_t, [C.c_int32] * 3),
    ("recon_gcn_bwd", C.c_int, [C.POINTER(GcnBwdArgs), C.c_void_p]),
    ("recon_gcn_b16_planes_bytes", C.c_size_t, [C.c_int32, C.c_int32]),
    ("recon_gcn_b16_fwd", C.c_int, [C.POINTER(GcnB16Args), C.c_void_p]),
    ("recon_gcn_b16_bwd_partial_floats", C.c_size_t, [C.c_int32] * 4),
    ("recon_gcn_b16_bwd", C.c_int, [C.POINTER(GcnB16BwdArgs), C.c_void_p]),
    ("recon_gcn_b16_stack_fwd", C.c_int, [C.POINTER(GcnB16StackArgs), C.c_void_p]),
    ("recon_gcn_b16_transposed_planes", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    ("recon_gcn_b16_stack_train_fwd", C.c_int, [C.POINTER(GcnB16StackTrainArgs), C.c_void_p]),
    ("recon_gcn_b16_stack_bwd_partial_floats", C.c_size_t, [C.c_int32] * 5),
    ("recon_gcn_b16_stack_train_bwd", C.c_int, [C.POINTER(GcnB16StackTrainArgs), C.c_void_p]),
    ("recon_sgemm_small_workspace_floats", C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    ("recon_sgemm_ex_workspace_floats", C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    ("recon_sgemm_ex", C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_int32, c_f32p, C.c_int32,
                                 c_f32p, C.c_void_p]),
    ("recon_sgemm_small", C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_int32, c_f32p, C.c_int32,
                                    c_f32p, C.c_void_p]),
    ("recon_sgemm", C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_int32,
                              c_f32p, C.c_int32, C.c_void_p]),
    ("recon_gat_atp_split_bytes", C.c_size_t, [C.c_int32] * 4),
    ("recon_gat_atp_bwd_split_bytes", C.c_size_t, [C.c_int32] * 3),
    ("recon_sgemm_bx3_workspace_bytes", C.c_size_t, [C.c_int32, C.c_int32]),
    ("recon_sgemm_bx3", C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32,
                                  C.c_void_p, C.c_void_p]),
    ("recon_sgemm_bx3_tn_workspace_bytes", C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    ("recon_sgemm_bx3_tn", C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32,
                                     C.c_void_p, C.c_void_p]),
    ("recon_transe_margin_fwd", C.c_int, [c_f32p, c_f32p, c_i64p, C.c_int64, C.c_int32, C.c_int32, C.c_float, c_f32p, c_f32p, c_i64p, c_i64p, C.c_void_p, C.c_void_p]),
    ("recon_transe_margin_fwd_keys", C.c_int, [c_f32p, c_f32p, c_i64p, C.c_int64, C.c_int32, C.c_int32, C.c_float, c_f32p, c_f32p, c_i64p, C.c_int64, C.c_void_p]),
    ("recon_transe_margin_bwd", C.c_int, [c_f32p, c_f32p, c_i64p, C.c_int64, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    ("recon_kg_nhop_lds_bytes", C.c_size_t, [C.c_int64]),
    ("recon_kg_adj_count", C.c_int, [C.POINTER(ReconKG), c_i64p, C.c_int32, c_i64p, C.c_void_p, C.c_void_p, c_i64p, c_i64p, c_i64p, c_i64p, C.c_void_p, C.c_void_p]),
    ("recon_kg_adj_fill", C.c_int, [C.POINTER(ReconKG), c_i64p, C.c_int32, c_i64p, C.c_int64, c_i64p, c_i64p, C.c_void_p]),
    ("recon_kg_nhop", C.c_int, [C.POINTER(ReconKG), c_i64p, C.c_int32, C.c_int32, C.c_int32, c_i64p, c_i64p, c_i64p, c_i64p, C.c_void_p, C.c_void_p]),
    ("recon_kg_nhop_count_early", C.c_int, [C.POINTER(ReconKG), c_i64p, C.c_int32, c_i64p, C.c_int32, c_i64p, c_i64p, c_i64p, C.c_void_p]),
    ("recon_rows_normalize_fwd", C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_float, c_f32p, c_f32p, C.c_void_p]),
    ("recon_rows_normalize_bwd", C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int32, C.c_float, c_f32p, c_f32p, C.c_void_p]),
    ("recon_gat_atp_bf16_io_supported", C.c_int, [C.c_int32] * 4),
    ("recon_set_nan_flag", C.c_int, [C.c_int32, C.c_void_p]),
    ("recon_config_set", C.c_int, [C.c_char_p, C.c_char_p]),
    ("recon_config_get", C.c_char_p, [C.c_char_p]),
    ("recon_hx2_aux_bytes", C.c_size_t, []),
    ("recon_sgemm_hx2_workspace_bytes", C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    ("recon_sgemm_hx2", C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32,
                                  C.c_void_p, C.c_void_p]),
    ("recon_sgemm_hx2_presplit", C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_void_p, C.c_void_p]),
    ("recon_sgemm_hx2_tn_workspace_bytes", C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    ("recon_sgemm_hx2_tn", C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32,
                                     C.c_void_p, C.c_void_p]),
    ("recon_sgemm_hx2_tn_presplit", C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_void_p, C.c_void_p]),
]

_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises RuntimeError if the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "librecon_hip.so not found at %s: build it with `make -C recon_amd/csrc` "
                "(recon_amd has no CPU fallback)" % LIB_PATH)
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so).  Device pointers and streams handed to the
        # C ABI come from torch, so the library must bind to THAT runtime: load torch first, otherwise the
        # dynamic loader resolves librecon_hip.so's libamdhip64 to /opt/rocm and the process ends up with two
        # HIP runtimes (every launch then fails with RECON_ERR_LAUNCH).
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.recon_version() != 2:
            raise RuntimeError("librecon_hip.so ABI version mismatch")
        _lib = h
    return _lib


def config_set(name, value):
    """Override one run-time switch of the library (csrc/config.hip; value None: unset).  Returns the previous value (or None)."""
    h = lib()
    prev = h.recon_config_get(name.encode())
    prev = prev.decode() if prev is not None else None
    check(h.recon_config_set(name.encode(), None if value is None else str(value).encode()), "recon_config_set(%s)" % name)
    return prev


class config:
    """`with _lib.config(RECON_PROP_FWD="w"): ...` — switches set for the block, restored afterwards (tests, A/B tools)."""

    def __init__(self, **kv):
        self.kv, self.prev = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.prev[k] = config_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            config_set(k, v)
        return False


def check(code, what):
    if code != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, ERRORS.get(code, "unknown"), code))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


try:                                    # the raw handle without building a torch.cuda.Stream object (3-5 us per call otherwise: a launch-bound
    from torch._C import _cuda_getCurrentRawStream as _raw_stream      # step makes a dozen of them)
except ImportError:                     # pragma: no cover
    _raw_stream = None


def current_stream():
    import torch
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


def on_device(dev):
    """`with on_device(t.device):` — torch.cuda.device(dev) only when it is not the current device already (the usual case: one process
    per GPU); entering and leaving the real context costs ~4 us per call."""
    import torch
    if dev.index is None or torch.cuda.current_device() == dev.index:
        return _NULL_CTX
    return torch.cuda.device(dev)
