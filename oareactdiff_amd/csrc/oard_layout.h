// oard_layout.h — host-side descriptions shared by the packer and the launcher:
// compile-time dimension helper, packed-weight offsets, canonical parameter order,
// topology tables and workspace carving.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/oard.h"

#define OARD_MAX_LAYERS 16
#define OARD_MAX_GROUP 1024   // atoms per (object, sample) group handled by k_geom

template <int H_, int R_>
struct Dims {
    static constexpr int H = H_, R = R_;
    static constexpr int HT = (H + 15) / 16, HP = HT * 16;          // hidden tiles / padded
    static constexpr int W = 3 * H + R, WB = (W + 15) / 16, WP = WB * 16;  // edge-state width
    static constexpr int RB = (R + 15) / 16, RP = RB * 16;
    static constexpr int H2 = H / 2, PB = (H2 + 15) / 16, PP = PB * 16;    // pos_expansion hidden
    static constexpr int D1T = (3 * H + 15) / 16, D1P = D1T * 16;          // dir_proj hidden (natural order)
    static constexpr int H4 = H / 4;                                        // lin3 hidden
    static_assert(H % 4 == 0 && R % 4 == 0, "feature widths must be multiples of 4");
};

struct RDims {  // the same numbers at run time
    int H, R, HT, HP, W, WB, WP, RB, RP, H2, PB, PP, D1T, D1P, H4;
    explicit RDims(int h, int r) {
        H = h; R = r; HT = (H + 15) / 16; HP = HT * 16; W = 3 * H + R; WB = (W + 15) / 16; WP = WB * 16;
        RB = (R + 15) / 16; RP = RB * 16; H2 = H / 2; PB = (H2 + 15) / 16; PP = PB * 16;
        D1T = (3 * H + 15) / 16; D1P = D1T * 16; H4 = H / 4;
    }
};

// EquiUpdate's frame-scalar MLP (lin3: 3 -> 48 -> 8 -> 1, leftnet.py:333) sees (x, 0, 0) under the exact node frame: a smooth function
// of ONE variable per layer, evaluated for every (node, channel).  Tabulated at pack time on [-L3T_X, L3T_X) in steps of L3T_H (a
// power of two: index and fraction are exact in float32), values and derivatives from a float64 evaluation; cubic Hermite in the
// kernel.  The table is CHECKED against the float64 function at three points per interval when it is built and carries a flag: the
// kernel uses it only if the worst deviation is below 2e-7 of the function's range (float32 resolution of the direct evaluation),
// and evaluates the MLP directly for arguments outside the table.
#define L3T_X 16.0f
#define L3T_H 0.03125f
#define L3T_N 1024              // 2 L3T_X / L3T_H intervals
#define L3T_FLOATS (2 * (L3T_N + 1) + 4)

// ---- packed weight blob: offsets in floats -----------------------------------------------------
struct LayerOff {
    // GCLMessage (leftnet.py:128-183)
    size_t ln_g_w, ln_g_b, W1a, b1, W1b, W1c, W2, b2, watt, batt, W3, b3, nm0, nm0b, nm1, nm1b;
    // EquiMessage (leftnet.py:186-289)
    size_t ln_q_w, ln_q_b, xp0, xp2, dp0, dp0b, dp2, dp2b, rbfp;
    // EquiUpdate (leftnet.py:292-346)
    size_t vp, xv0, xv2, l3u;   // l3u raw: w0[48*3] b0[48] w2[8*48] b2[8] w4[8] b4[1]
    size_t l3t;                 // the frame-scalar MLP as a table (k_lin3u_table): [L3T_N + 1] x (f, f' h), then the "table is good" flag
    // LDS weight streams of the two hot edge kernels, chunks in consumption order (oard_edge_v1.h)
    size_t gcl_stream, equi_stream;
    size_t gcl_b3, equi_b3;     // split-precision streams of the two edge kernels (oard_edge_b3.h); built only when those kernels are enabled
    int xcross;                 // reflect_equiv = False: the Equi message carries x (x) coord_cross as well (leftnet.py:268-272)
};
struct PackOff {
    size_t emb, emb_b, nbemb, nbemb_b, s2v, s2v_b, rl0, rl0_b, rl2, rl2_b;
    size_t lin3;      // raw: w0[H4*3] b0[H4] w2[H4] b2[1]
    size_t pe0;       // raw [H2*3]
    size_t pe1, embout, embout_b, v1p, v2p, un0, un0_b, un2, un2_b;
    size_t c0row;     // [WP] constant state of a masked edge
    size_t u0;        // [HP] W1c(layer 0) . c0row: stage S1 of layer 0 on inter-object edges
    size_t rbf_means, rbf_betas;  // [RP]
    size_t enc[OARD_MAX_OBJECTS], dec[OARD_MAX_OBJECTS];  // raw MLP blocks
    LayerOff layer[OARD_MAX_LAYERS];
    size_t total;
    int signed_scal;  // reflect_equiv = False: no |.| on the second frame component of the edge scalarisation (leftnet.py:794-796)
};

// ---- topology tables (device pointers) -----------------------------------------------------------
struct TopoDev {
    int N, B, n_obj, n_groups;
    int npb;       // real nodes per 16-column workgroup of the node stages (< 16 on small batches: more workgroups)
    long long E, A;
    const int *node_obj, *node_row, *node_ref, *node_tidx, *node_sample, *sample_ptr;
    const int *edge_ptr, *edge_src, *edge_tgt;   // implicit ("logical") edge ids: node n owns ids edge_ptr[n] + rank
    // physical rows of the edge-state / message buffers: inner (same-object) edges first, in the
    // target-sorted order of the act_* list (row a == inner entry a), then the inter-object edges in
    // logical order, then one spare row (index E) for padding columns
    const int *edge_row;                         // [E]   logical id -> physical row
    const int *row_src, *row_tgt;                // [E+1] physical row -> nodes
    const int *row_eid;                          // [E+1] physical row -> logical id (the message buffer m is kept in logical order)
    const int *grp_ptr;                       // [n_groups+1], group q = sample*n_obj + obj
    const int *act_ptr, *act_src, *act_tgt, *act_edge;
    const long long *ref_edge_ptr;            // [N] first reference-order edge of internal node n
};

// Inner (same-object) edges whose distance is inside the cutoff, compacted once per forward call (k_active_list): EquiMessage is
// exactly zero on the others (model/leftnet.py:748-753, 768-771: rbf * mask = 0 -> rbf_proj = 0), so k_equi_edge_v1 runs the
// listed rows only and the node stage walks the list.  Target-sorted like the rows themselves: every per-node sum keeps its order.
// All pointers NULL = every inner row is active (training-mode forward; first-generation kernels).
struct ActList {
    const int* rows;   // [n_act] physical row (== inner entry) of the k-th active edge, ascending
    const int* src;    // [n_act] its source node (act_src[rows[k]])
    const int* pre;    // [A + 1] number of active rows in front of row a  (node n owns list entries [pre[act_ptr[n]], pre[act_ptr[n + 1]]))
    const int* n;      // [1]     n_act
};

// A topology is split into up to OARD_MAX_PARTS independent sub-batches (contiguous ranges of samples):
// oard_forward runs them concurrently on internal streams so that the low-occupancy node stages and the
// tail of one part's edge kernels overlap with the other parts' edge kernels.  Reactions never interact,
// so the split changes nothing in the results.
#define OARD_MAX_PARTS 8
struct TopoPart {
    TopoDev d;
    void* dev_block = nullptr;     // one allocation holding every table of this part (from the table pool, oard_hip.hip)
    size_t dev_cap = 0;            // its capacity in the pool
    void* stage = nullptr;         // pinned block the tables were uploaded from (kept until the topology is destroyed)
    size_t stage_cap = 0;
    size_t ws_off = 0;             // byte offset of this part's slice of the workspace
    int conc = 1;                  // sub-batches of this topology that run concurrently (launch-shape heuristics)
    mutable size_t vec_final = 0;  // workspace offset (within the slice) of the vec buffer holding the final state
};
struct oard_topology {
    int device = 0;                // HIP device the tables live on (oard_topology_destroy returns them to THAT device's pool)
    int n_parts = 0;
    TopoPart parts[OARD_MAX_PARTS];
    int n_obj = 0, B = 0;
    long long N = 0, E = 0, A = 0;
    int max_group = 0, max_ns = 0;
    int obj_start[OARD_MAX_OBJECTS + 1] = {};   // reference (object-major) row ranges of the objects
    hipStream_t side[OARD_MAX_PARTS] = {};
    hipEvent_t ev_fork = nullptr, ev_join[OARD_MAX_PARTS] = {};
    // reference-order tables (whole batch) for oard_topology_check_edge_index: dense sample id, rank of the node inside its sample,
    // first reference-order edge id of the node
    void* ref_block = nullptr;
    size_t ref_cap = 0;
    void* ref_stage = nullptr;
    size_t ref_stage_cap = 0;
    hipEvent_t ready = nullptr;        // recorded on the null stream behind the table uploads
    // streams that have been given work reading this topology's tables (noted by the entry points): oard_topology_destroy records an
    // event on each and the table pool hands the blocks out again only after those events have fired - no hipFree (a device-wide sync)
    mutable hipStream_t used_on[4] = {};
    mutable int n_used = 0;
    mutable bool used_many = false;    // more than 4 distinct streams: destroy falls back to a device synchronisation
    const int *ref_sample = nullptr, *ref_rank = nullptr;
    const long long* ref_ptr = nullptr;
    int N_ref = 0;
};

// ---- workspace carving (byte offsets) --------------------------------------------------------------
struct WsOff {
    size_t pos, pf64, pf32, x1, pp0, labels, hin, zemb, nb, s, s1, ne1, xh, P, Q, xq, vec, vec2, v2buf, sc0, vdot,
        geo, d64, rbuf, ew, mbuf, xmsg, vmsg, dpos, hout, total;   // xmsg..vmsg double as qbuf [A][3][HP] (v1)
    size_t al_rows, al_src, al_pre, al_cnt, al_n;   // ActList of this call + the per-block counts it is built from
    size_t d1s;                    // split-precision EquiMessage kernel: SiLU(d1) in wave-tile order, [ceil(A / 128) * 8][D1T][256] floats
    size_t small_a, small_b;       // stage-split EquiMessage latency path: d1 [A+1][D1P]; 0 = not allocated (large topologies)
};
