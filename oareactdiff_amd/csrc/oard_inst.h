// Explicit instantiations of the heavy kernel families, one translation unit per family (oareactdiff_amd/build.py SOURCES).
// oard_hip.hip - the host code, which launches every kernel - includes this file with OARD_INST_DEFINE unset: every kernel listed here
// is declared `extern template`, i.e. NOT instantiated there; oard_inst_<family>.hip defines OARD_INST_DEFINE and one OARD_INST_UNIT_*
// and emits the family's kernels.  The lists are the launch shapes in use (oard_hip.hip: launch_gcl_v1s, launch_equi_v1,
// gcl_backward_impl, wgrad_impl, forward_impl); a shape that is launched but not listed here is simply instantiated in oard_hip.hip
// as before (correct, only slower to build); a shape listed here but never launched costs build time and nothing else.
// Experiment / probe builds (-DOARD_EXPERIMENTS, -DOARD_SINGLE_TU) keep everything in oard_hip.hip.
#pragma once
#if !defined(OARD_EXPERIMENTS) && !defined(OARD_SINGLE_TU)

#define OARD_UNPAREN(...) __VA_ARGS__
#ifdef OARD_INST_DEFINE
#define OARD_INST(K, ARGS) template __global__ void OARD_UNPAREN K ARGS;
#else
#define OARD_INST(K, ARGS) extern template __global__ void OARD_UNPAREN K ARGS;
#endif
#ifndef OARD_DIMS_LIST
#define OARD_DIMS_LIST X(196, 96) X(32, 8) X(32, 32)
#endif
#define OARD_BOOL4(M, D) M(D, true, true) M(D, false, true) M(D, true, false) M(D, false, false)      /* D = (Dims<h, r>), parenthesised */

#define OARD_NW(h) (((h) + 15) / 16 <= 16 ? ((h) + 15) / 16 : 8)       /* forward_impl: NW = D::HT <= 16 ? D::HT : 8 */
#define OARD_A_GCL (TopoDev, const float*, const float*, const float*, const float*, const float*, long long, long long, const float*, float*, float*, GclTape)

// ---- k_gcl_edge_p: persistent GCL throughput shape, inference and training-mode (oard_edge_p.h) --------------------------------------
#if !defined(OARD_INST_DEFINE) || defined(OARD_INST_UNIT_GCL_P)
#define OARD_I_GCL_P(D, S1, S3) OARD_INST((k_gcl_edge_p<OARD_UNPAREN D, S1, S3, false>), OARD_A_GCL) OARD_INST((k_gcl_edge_p<OARD_UNPAREN D, S1, S3, true>), OARD_A_GCL)
#define X(h, r) OARD_BOOL4(OARD_I_GCL_P, (Dims<h, r>))
OARD_DIMS_LIST
#undef X
#endif

// ---- k_gcl_edge_v1 (8-wave ring-3 shape: inference + training-mode; 4-wave shape), k_gcl_edge_small (oard_edge_v1.h, oard_edge_small.h) ---
#if !defined(OARD_INST_DEFINE) || defined(OARD_INST_UNIT_GCL_V1)
#define OARD_I_GCL_V1(D, S1, S3) \
    OARD_INST((k_gcl_edge_v1<OARD_UNPAREN D, 8, 2, S1, S3, false, 2, 3>), OARD_A_GCL) OARD_INST((k_gcl_edge_v1<OARD_UNPAREN D, 8, 2, S1, S3, true, 2, 3>), OARD_A_GCL) \
    OARD_INST((k_gcl_edge_v1<OARD_UNPAREN D, 4, 2, S1, S3, false, 2, 2>), OARD_A_GCL) \
    OARD_INST((k_gcl_edge_small<OARD_UNPAREN D, 8, S1, S3>), (TopoDev, const float*, LayerOff, const float*, const float*, const float*, const float*, long long, long long, float*, float*))
#define X(h, r) OARD_BOOL4(OARD_I_GCL_V1, (Dims<h, r>))
OARD_DIMS_LIST
#undef X
#endif

// ---- split-precision edge kernels (oard_edge_b3.h) -------------------------------------------------------------------------------------
#if !defined(OARD_INST_DEFINE) || defined(OARD_INST_UNIT_B3)
#define OARD_A_EQUI_B3 (TopoDev, const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*, ActList)
#define OARD_I_GCL_B3(D, S1, S3) OARD_INST((k_gcl_edge_b3<OARD_UNPAREN D, S1, S3, false>), OARD_A_GCL) OARD_INST((k_gcl_edge_b3<OARD_UNPAREN D, S1, S3, true>), OARD_A_GCL)
#define X(h, r) OARD_BOOL4(OARD_I_GCL_B3, (Dims<h, r>)) OARD_INST((k_equi_edge_b3<Dims<h, r>, false>), OARD_A_EQUI_B3) OARD_INST((k_equi_edge_b3<Dims<h, r>, true>), OARD_A_EQUI_B3)
OARD_DIMS_LIST
#undef X
#endif

// ---- EquiMessage edge kernel + fused node stage, the two backward edge kernels (oard_edge_v1.h, oard_node_v1.h, oard_edge_bwd.h) ----------
#if !defined(OARD_INST_DEFINE) || defined(OARD_INST_UNIT_EQUI)
#define OARD_A_EQUI (TopoDev, const float*, const float*, const float*, const float*, float*, float*, float*, ActList)
#define OARD_A_EQUI_NODE (TopoDev, const float*, LayerOff, const float*, const float*, const float*, const float*, const float*, float*, const float*, float*, float*, float*, ActList)
#define X(h, r) \
    OARD_INST((k_equi_edge_v1<Dims<h, r>, 8, false>), OARD_A_EQUI) OARD_INST((k_equi_edge_v1<Dims<h, r>, 8, true>), OARD_A_EQUI) \
    OARD_INST((k_equi_edge_v1<Dims<h, r>, 4, false>), OARD_A_EQUI) \
    OARD_INST((k_equi_node_v1<Dims<h, r>, OARD_NW(h), false, false>), OARD_A_EQUI_NODE) OARD_INST((k_equi_node_v1<Dims<h, r>, OARD_NW(h), false, true>), OARD_A_EQUI_NODE) \
    OARD_INST((k_equi_node_v1<Dims<h, r>, OARD_NW(h), true, false>), OARD_A_EQUI_NODE) OARD_INST((k_equi_node_v1<Dims<h, r>, OARD_NW(h), true, true>), OARD_A_EQUI_NODE) \
    OARD_INST((k_gcl_edge_bwd<Dims<h, r>, 8, 2, true>), (TopoDev, const float*, long long, long long, GclBwdArgs)) \
    OARD_INST((k_gcl_edge_bwd<Dims<h, r>, 8, 2, false>), (TopoDev, const float*, long long, long long, GclBwdArgs)) \
    OARD_INST((k_equi_edge_bwd<Dims<h, r>, 8>), (TopoDev, const float*, const float*, const float*, float*, float*))
OARD_DIMS_LIST
#undef X
#endif

// ---- node-side kernels of both passes (oard_node_v1.h, oard_node_bwd.h, oard_rows.h).  The row-generic dense kernels are templates of
// tile counts, not of the widths: the shapes of the production widths are listed (other widths instantiate theirs in oard_hip.hip). ---------
#if !defined(OARD_INST_DEFINE) || defined(OARD_INST_UNIT_NODE)
#define OARD_A_GCL_NODE (TopoDev, const float*, LayerOff, const float*, const float*, float*, float*, float*)
#define OARD_A_OUT (TopoDev, const float*, PackOff, const float*, const float*, float*, float*, int*)
#define OARD_A_SCAL_BWD (TopoDev, const float*, const float*, int, const float*, const float*, float*, float*)
#define OARD_A_MSG_BWD (TopoDev, const float*, Strided3, Strided3, const float*, Strided3, const float*, int, Strided3, float*, float*, float*, float*, int, int, int)
#define X(h, r) \
    OARD_INST((k_gcl_node_v1<Dims<h, r>, OARD_NW(h), false>), OARD_A_GCL_NODE) OARD_INST((k_gcl_node_v1<Dims<h, r>, OARD_NW(h), true>), OARD_A_GCL_NODE) \
    OARD_INST((k_out_v1<Dims<h, r>, OARD_NW(h)>), OARD_A_OUT) \
    OARD_INST((k_scalarize_bwd<Dims<h, r>, 2, false>), OARD_A_SCAL_BWD) OARD_INST((k_scalarize_bwd<Dims<h, r>, 2, true>), OARD_A_SCAL_BWD) \
    OARD_INST((k_equi_msg_bwd<Dims<h, r> >), OARD_A_MSG_BWD)
OARD_DIMS_LIST
#undef X
OARD_INST((k_lin3u_bwd_fused<4>), (const float*, const float*, const float*, long long, float*, float*))
#define OARD_I_ROWS(KB, EPI) OARD_INST((k_rows_dense<KB, EPI, 8>), (RowsDense)) OARD_INST((k_rows_dense_long<KB, EPI, 8, 4>), (RowsDense))
OARD_I_ROWS(13, 0) OARD_I_ROWS(13, 2) OARD_I_ROWS(13, 3) OARD_I_ROWS(7, 3) OARD_I_ROWS(6, 0) OARD_I_ROWS(6, 1) OARD_I_ROWS(4, 1) OARD_I_ROWS(4, 3)
OARD_INST((k_rows_dense<26, 1, 8>), (RowsDense)) OARD_INST((k_rows_dense<26, 3, 8>), (RowsDense))
OARD_INST((k_rows_dense2<13, 1, 8, 13>), (RowsDense2)) OARD_INST((k_rows_dense2<13, 2, 8, 13>), (RowsDense2))
OARD_INST((k_rows_dense2<26, 1, 8, 13>), (RowsDense2)) OARD_INST((k_rows_dense2<39, 2, 8, 13>), (RowsDense2))
#endif

// ---- weight-gradient GEMMs (oard_wgrad_t16.h, oard_edge_bwd.h): not templates of the widths ----------------------------------------------
#if !defined(OARD_INST_DEFINE) || defined(OARD_INST_UNIT_WGRAD)
#define OARD_A_WG (const float*, int, int, const float*, int, int, long long, long long, long long, int, int, float*, float*, float*)
OARD_INST((k_wgrad_t16<4, 7, false>), (WgtArgs)) OARD_INST((k_wgrad_t16<4, 7, true>), (WgtArgs)) OARD_INST((k_wgrad_t16<5, 3, false>), (WgtArgs))
OARD_INST((k_wgrad_t16<5, 5, true>), (WgtArgs)) OARD_INST((k_wgrad_t16<5, 7, true>), (WgtArgs)) OARD_INST((k_wgrad_t16<6, 5, false>), (WgtArgs))
OARD_INST((k_wgrad_t16<6, 7, false>), (WgtArgs))
OARD_INST((k_wgrad<false, 7>), OARD_A_WG) OARD_INST((k_wgrad<false, 8>), OARD_A_WG) OARD_INST((k_wgrad<true, 7>), OARD_A_WG) OARD_INST((k_wgrad<true, 8>), OARD_A_WG)
OARD_INST((k_wgrad_lds<false, 7, 4, 2>), OARD_A_WG) OARD_INST((k_wgrad_lds<false, 8, 4, 2>), OARD_A_WG)
OARD_INST((k_wgrad_lds<true, 7, 4, 2>), OARD_A_WG) OARD_INST((k_wgrad_lds<true, 8, 4, 2>), OARD_A_WG)
#endif

#endif  // !OARD_EXPERIMENTS && !OARD_SINGLE_TU
