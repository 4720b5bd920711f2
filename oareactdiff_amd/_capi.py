"""ctypes binding of liboard_hip.so (include/oard.h).  No CPU fallback: if the library is
missing or fails to load, importing the HIP backend raises."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB

OARD_MAX_OBJECTS = 8
ABI_VERSION = 2040            # OARD_VERSION of csrc/oard_hip.hip (checked by lib(): a stale library is refused, not misread)
OARD_OK, OARD_EINVAL, OARD_ENOTCOMPLETE, OARD_EHIP, OARD_ENOMEM = 0, -1, -2, -3, -4
ERRORS = {OARD_EINVAL: "invalid argument / unsupported configuration", OARD_ENOTCOMPLETE: "topology is not complete-per-sample",
          OARD_EHIP: "HIP runtime error", OARD_ENOMEM: "workspace too small"}

PREC_GCL_BF16X3, PREC_EQUI_BF16X3, PREC_TRAIN_BF16X3 = 1, 2, 4      # oard_config.precision bits
TAP_S, TAP_VEC, TAP_EDGE, TAP_POS_FRAME, TAP_DPOS, TAP_HOUT, TAP_LABELS, TAP_NE1 = 1, 2, 3, 4, 5, 6, 7, 8

EXPORTS = ["oard_version", "oard_supported", "oard_param_count", "oard_packed_bytes", "oard_pack_weights",
           "oard_topology_create", "oard_topology_destroy", "oard_topology_num_nodes", "oard_topology_num_edges",
           "oard_topology_num_inner_edges", "oard_topology_num_samples", "oard_topology_check_edge_index",
           "oard_workspace_bytes", "oard_forward", "oard_active_inner_edges", "oard_debug_lin3u_table", "oard_sampler_step", "oard_sampler_step_dev", "oard_tap", "oard_debug_stop_after", "oard_debug_option", "oard_timing_enable", "oard_timing_reset",
           "oard_timing_get",
           "oard_topology_create_parts", "oard_topology_export", "oard_tape_bytes", "oard_tape_entry", "oard_forward_train",
           "oard_packed_bwd_bytes", "oard_pack_weights_bwd", "oard_gcl_backward_dx", "oard_edge_node_sums",
           "oard_equi_backward_dx", "oard_scalarize_backward", "oard_equi_msg_backward", "oard_lin3u_forward", "oard_lin3u_backward", "oard_wgrad_scratch_bytes", "oard_wgrad",
           "oard_train_scratch_bytes", "oard_train_scratch_poison", "oard_train_scratch_entry", "oard_train_tail_backward",
           "oard_train_layer_backward", "oard_train_init_backward", "oard_train_stage_backward",
           "oard_loss_prepare", "oard_loss_terms", "oard_adamw_step", "oard_adamw_step_dev", "oard_nan_replace",
           "oard_graph_create", "oard_graph_destroy", "oard_graph_num_nodes", "oard_graph_num_edges", "oard_graph_object_rows",
           "oard_graph_is_complete", "oard_graph_workspace_bytes", "oard_graph_forward", "oard_library_stream"]
STAGE_RECOMPUTE, STAGE_UPDATE, STAGE_MESSAGE, STAGE_GCL_NODE, STAGE_NODE_PRE, STAGE_GCL_EDGE, STAGE_EQUI_EDGE = range(7)
SCRATCH_XH, SCRATCH_XQ, SCRATCH_CR, SCRATCH_DCD, SCRATCH_DCR = range(1, 6)

# oard_topology_export tables / oard_tape_entry tensors (include/oard.h)
TOPO_NODE_REF, TOPO_NODE_OBJ, TOPO_NODE_ROW, TOPO_NODE_SAMPLE, TOPO_NODE_TIDX, TOPO_SAMPLE_PTR, TOPO_GROUP_PTR, \
    TOPO_INNER_SRC, TOPO_INNER_TGT, TOPO_ROW_SRC, TOPO_ROW_TGT = range(1, 12)
TAPE_HIN, TAPE_GEO, TAPE_RBF, TAPE_PP0, TAPE_X1 = 1, 2, 3, 4, 5
TAPE_S_IN, TAPE_VEC_IN, TAPE_EW, TAPE_AGG, TAPE_S_MID, TAPE_Z1, TAPE_Z2, TAPE_ATT, TAPE_Z3, TAPE_ZD1, TAPE_CD, TAPE_S_A, TAPE_VEC_A = range(16, 29)


class OardConfig(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32), ("num_radial", C.c_int32), ("num_layers", C.c_int32), ("in_hidden", C.c_int32),
        ("n_obj", C.c_int32), ("node_nf", C.c_int32 * OARD_MAX_OBJECTS), ("enc_alias", C.c_int32 * OARD_MAX_OBJECTS),
        ("condition_nf", C.c_int32), ("condition_time", C.c_int32), ("pos_dim", C.c_int32), ("cutoff", C.c_float),
        ("reflect_equiv", C.c_int32), ("precision", C.c_int32),
    ]


class OardError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != OARD_OK:
        raise OardError(f"{what} failed: {ERRORS.get(rc, rc)}")


_lib = None


def lib() -> C.CDLL:
    """Loads the in-tree library (built by `python -m oareactdiff_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise OardError(f"{LIB} not found: build it with `python -m oareactdiff_amd.build` "
                        "(there is no CPU fallback for the HIP backend)")
    L = C.CDLL(LIB)
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    cfgp = C.POINTER(OardConfig)
    L.oard_version.restype = C.c_int
    got = L.oard_version()
    if got != ABI_VERSION:      # e.g. an OARD_LIB override / ablation build that predates a change of oard_config: its fields would be misread
        raise OardError(f"{LIB} has ABI version {got}, this package expects {ABI_VERSION}: rebuild it "
                        "(`python -m oareactdiff_amd.build --force`, or the script that made the OARD_LIB build)")
    L.oard_supported.argtypes = [cfgp]; L.oard_supported.restype = C.c_int
    L.oard_param_count.argtypes = [cfgp]; L.oard_param_count.restype = sz
    L.oard_packed_bytes.argtypes = [cfgp]; L.oard_packed_bytes.restype = sz
    L.oard_pack_weights.argtypes = [cfgp, C.POINTER(vp), sz, vp, sz, vp]; L.oard_pack_weights.restype = C.c_int
    L.oard_topology_create.argtypes = [cfgp, C.POINTER(i64), C.POINTER(i64), i64, C.POINTER(vp)]
    L.oard_topology_create.restype = C.c_int
    L.oard_topology_destroy.argtypes = [vp]; L.oard_topology_destroy.restype = None
    for f in ("oard_topology_num_nodes", "oard_topology_num_edges", "oard_topology_num_inner_edges",
              "oard_topology_num_samples"):
        getattr(L, f).argtypes = [vp]; getattr(L, f).restype = i64
    L.oard_topology_check_edge_index.argtypes = [vp, vp, i64, vp, vp]; L.oard_topology_check_edge_index.restype = C.c_int
    L.oard_workspace_bytes.argtypes = [cfgp, vp]; L.oard_workspace_bytes.restype = sz
    L.oard_forward.argtypes = [cfgp, vp, vp, C.POINTER(vp), vp, C.c_int, vp, C.POINTER(vp), vp, sz, vp, vp]
    L.oard_forward.restype = C.c_int
    L.oard_debug_lin3u_table.argtypes = [cfgp, vp, C.c_int, vp, vp]; L.oard_debug_lin3u_table.restype = C.c_int
    L.oard_active_inner_edges.argtypes = [cfgp, vp, vp, sz, C.POINTER(i64), vp]; L.oard_active_inner_edges.restype = C.c_int
    L.oard_sampler_step.argtypes = [cfgp, vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                    C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(vp), vp]
    L.oard_sampler_step.restype = C.c_int
    L.oard_sampler_step_dev.argtypes = [cfgp, vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, C.c_int,
                                        C.POINTER(vp), vp]
    L.oard_sampler_step_dev.restype = C.c_int
    L.oard_tap.argtypes = [cfgp, vp, vp, C.c_int, C.c_int, vp, vp]; L.oard_tap.restype = C.c_int
    ci = C.c_int
    L.oard_topology_create_parts.argtypes = [cfgp, C.POINTER(i64), C.POINTER(i64), i64, ci, C.POINTER(vp)]
    L.oard_topology_create_parts.restype = ci
    L.oard_topology_export.argtypes = [vp, ci, vp, i64, vp]; L.oard_topology_export.restype = ci
    L.oard_tape_bytes.argtypes = [cfgp, vp]; L.oard_tape_bytes.restype = sz
    L.oard_tape_entry.argtypes = [cfgp, vp, ci, ci, C.POINTER(sz), C.POINTER(i64), C.POINTER(i64)]; L.oard_tape_entry.restype = ci
    L.oard_forward_train.argtypes = [cfgp, vp, vp, C.POINTER(vp), vp, ci, vp, C.POINTER(vp), vp, sz, vp, sz, vp, vp]
    L.oard_forward_train.restype = ci
    L.oard_packed_bwd_bytes.argtypes = [cfgp]; L.oard_packed_bwd_bytes.restype = sz
    L.oard_pack_weights_bwd.argtypes = [cfgp, C.POINTER(vp), sz, vp, sz, vp]; L.oard_pack_weights_bwd.restype = ci
    L.oard_gcl_backward_dx.argtypes = [cfgp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]; L.oard_gcl_backward_dx.restype = ci
    L.oard_edge_node_sums.argtypes = [cfgp, vp, vp, vp, vp, vp]; L.oard_edge_node_sums.restype = ci
    L.oard_equi_backward_dx.argtypes = [cfgp, vp, vp, ci, vp, vp, vp, vp, vp]; L.oard_equi_backward_dx.restype = ci
    L.oard_scalarize_backward.argtypes = [cfgp, vp, vp, vp, vp, ci, vp, vp, vp, vp]; L.oard_scalarize_backward.restype = ci
    L.oard_equi_msg_backward.argtypes = [cfgp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]; L.oard_equi_msg_backward.restype = ci
    L.oard_lin3u_forward.argtypes = [cfgp, vp, ci, vp, i64, vp, vp]; L.oard_lin3u_forward.restype = ci
    L.oard_lin3u_backward.argtypes = [cfgp, vp, ci, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp]; L.oard_lin3u_backward.restype = ci
    L.oard_wgrad_scratch_bytes.argtypes = [ci, ci, i64]; L.oard_wgrad_scratch_bytes.restype = sz
    L.oard_wgrad.argtypes = [vp, ci, ci, ci, ci, ci, vp, ci, ci, ci, ci, ci, ci, i64, vp, vp, vp, sz, vp]; L.oard_wgrad.restype = ci
    pvp = C.POINTER(vp)
    L.oard_train_scratch_bytes.argtypes = [cfgp, vp]; L.oard_train_scratch_bytes.restype = sz
    L.oard_train_scratch_poison.argtypes = [cfgp, vp, vp, sz, vp]; L.oard_train_scratch_poison.restype = ci
    L.oard_train_scratch_entry.argtypes = [cfgp, vp, ci, C.POINTER(sz), C.POINTER(i64), C.POINTER(i64)]; L.oard_train_scratch_entry.restype = ci
    L.oard_train_tail_backward.argtypes = [cfgp, vp, vp, vp, vp, pvp, vp, vp, pvp, pvp, vp, sz, vp]; L.oard_train_tail_backward.restype = ci
    L.oard_train_layer_backward.argtypes = [cfgp, vp, vp, vp, vp, ci, vp, vp, vp, pvp, pvp, vp, sz, vp]; L.oard_train_layer_backward.restype = ci
    L.oard_train_init_backward.argtypes = [cfgp, vp, vp, vp, vp, pvp, vp, vp, pvp, pvp, vp, sz, vp]; L.oard_train_init_backward.restype = ci
    L.oard_train_stage_backward.argtypes = [cfgp, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, pvp, pvp, vp, sz, vp]
    L.oard_train_stage_backward.restype = ci
    L.oard_nan_replace.argtypes = [cfgp, vp, vp, pvp, pvp, vp]; L.oard_nan_replace.restype = ci
    pf, cf = C.POINTER(C.c_float), C.c_float
    L.oard_loss_prepare.argtypes = [cfgp, vp, pvp, pvp, pvp, pvp, vp, vp, ci, pf, pf, ci, ci, pvp, pvp, vp]; L.oard_loss_prepare.restype = ci
    L.oard_loss_terms.argtypes = [cfgp, vp, pvp, pvp, pvp, pvp, pvp, vp, vp, ci, pf, pf, pf, ci, ci, vp, vp, pvp, vp]; L.oard_loss_terms.restype = ci
    cd = C.c_double
    L.oard_adamw_step.argtypes = [vp, vp, vp, vp, vp, i64, cd, cd, cd, cd, cd, i64, ci, cd, vp]; L.oard_adamw_step.restype = ci
    L.oard_adamw_step_dev.argtypes = [vp, vp, vp, vp, vp, i64, cd, cd, cd, cd, cd, ci, ci, vp, ci, vp, vp, vp, vp]; L.oard_adamw_step_dev.restype = ci
    L.oard_library_stream.argtypes = [ci, pvp]; L.oard_library_stream.restype = ci
    # general edge lists (csrc/oard_general.hip)
    L.oard_graph_create.argtypes = [cfgp, vp, vp, i64, vp, i64, pvp]; L.oard_graph_create.restype = ci
    L.oard_graph_destroy.argtypes = [vp]; L.oard_graph_destroy.restype = None
    L.oard_graph_num_nodes.argtypes = [vp]; L.oard_graph_num_nodes.restype = i64
    L.oard_graph_num_edges.argtypes = [vp]; L.oard_graph_num_edges.restype = i64
    L.oard_graph_object_rows.argtypes = [vp, ci]; L.oard_graph_object_rows.restype = i64
    L.oard_graph_is_complete.argtypes = [vp]; L.oard_graph_is_complete.restype = ci
    L.oard_graph_workspace_bytes.argtypes = [cfgp, vp]; L.oard_graph_workspace_bytes.restype = sz
    L.oard_graph_forward.argtypes = [cfgp, vp, pvp, sz, pvp, vp, ci, vp, pvp, vp, sz, vp, vp]; L.oard_graph_forward.restype = ci
    L.oard_debug_stop_after.argtypes = [C.c_int]; L.oard_debug_stop_after.restype = C.c_int
    L.oard_debug_option.argtypes = [C.c_char_p, C.c_int]; L.oard_debug_option.restype = C.c_int
    L.oard_timing_enable.argtypes = [C.c_int]; L.oard_timing_enable.restype = C.c_int
    L.oard_timing_reset.argtypes = []; L.oard_timing_reset.restype = C.c_int
    L.oard_timing_get.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]; L.oard_timing_get.restype = C.c_int
    for env, opt in (("OARD_GCL_VARIANT", b"gcl_variant"), ("OARD_EQUI_VARIANT", b"equi_variant"),
                     ("OARD_NODE_VARIANT", b"node_variant"), ("OARD_GCL_SKIP", b"gcl_skip"), ("OARD_EQUI_SKIP", b"equi_skip"), ("OARD_GCL_PERSIST", b"gcl_persist"), ("OARD_GCL_GRID", b"gcl_grid"), ("OARD_PARTS", b"parts"), ("OARD_SEQUENTIAL", b"sequential"), ("OARD_POISON", b"poison"), ("OARD_AUTO_SMALL", b"auto_small"), ("OARD_AUTO_TINY", b"auto_tiny"), ("OARD_NPB", b"npb"),
                     ("OARD_WGRAD_WGS", b"wgrad_wgs"), ("OARD_WGRAD_LDS", b"wgrad_lds"), ("OARD_WGRAD_T16", b"wgrad_t16"), ("OARD_WGRAD_QUEUE", b"wgrad_queue"), ("OARD_GATE_FOLD", b"gate_fold"), ("OARD_WGRAD_SHAPES", b"wgrad_shapes"), ("OARD_TRAIN_DUAL", b"train_dual"), ("OARD_SMALL_SPLIT", b"small_split"),
                     ("OARD_SKIP_FAMILIES", b"skip_families")):
        if os.environ.get(env) not in (None, ""):
            check(L.oard_debug_option(opt, int(os.environ[env])), f"oard_debug_option({opt.decode()})")
    _lib = L
    return L


def timing_get(family: str):
    ms, n = C.c_double(0.0), C.c_int64(0)
    check(lib().oard_timing_get(family.encode(), C.byref(ms), C.byref(n)), "oard_timing_get")
    return ms.value, n.value
