/*
 * oard.h — C ABI of liboard_hip.so: the MI355X (gfx950) implementation of OA-ReactDiff's
 * per-timestep denoising call,
 *
 *     EGNNDynamics.forward   oa_reactdiff/dynamics/egnn_dynamics.py:63-168
 *       -> LEFTNet.forward   oa_reactdiff/model/leftnet.py:724-891
 *
 * The reference has no FFI of its own (it is pure Python on torch ops); the interface each
 * entry point replaces is therefore the Python call it stands in for, cited per function.
 * Conventions: every function returns 0 on success or a negative OARD_E* code; nothing here
 * synchronises a stream or the device (oard_topology_create does its work on the host and leaves
 * ONE asynchronous upload behind; the only waits are in oard_topology_check_edge_index, a verification call);
 * all `*_dev` pointers are device pointers on the current HIP device; float tensors are
 * contiguous row-major fp32; the library never touches torch.
 * Threading: calls on DISTINCT (topology, workspace, tape, scratch) objects may run concurrently from several host threads /
 * on several streams: everything a call needs travels in its arguments (oard_config incl. the precision bits, the packed
 * blob, the topology, caller-owned buffers).  Process-global and NOT thread-safe are only the debugging facilities:
 * oard_debug_option / oard_debug_stop_after (A/B switches, read at launch time) and the oard_timing_* event recorder
 * (keep it disabled when calling from several threads).  One process per GPU remains the deployment model
 * (torch.distributed / RCCL).
 */
#ifndef OARD_H
#define OARD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OARD_MAX_OBJECTS 8

#define OARD_OK 0
#define OARD_EINVAL (-1)      /* bad argument / unsupported configuration */
#define OARD_ENOTCOMPLETE (-2)/* topology is not "complete graph per sample" */
#define OARD_EHIP (-3)        /* a HIP runtime call failed */
#define OARD_ENOMEM (-4)      /* workspace too small */

typedef void* oard_stream_t;  /* hipStream_t */

/* Mirrors the constructor arguments of EGNNDynamics / BaseDynamics
 * (oa_reactdiff/dynamics/_base.py:10-80) and the LEFTNet model_config
 * (oa_reactdiff/trainer/train_ts1x.py:43-56) that shape the forward pass. */
typedef struct oard_config {
    int32_t hidden;          /* hidden_channels H                                   */
    int32_t num_radial;      /* num_radial R                                        */
    int32_t num_layers;      /* num_layers L                                        */
    int32_t in_hidden;       /* in_hidden_channels (= embed_dim + time + conditions) */
    int32_t n_obj;           /* len(fragment_names)                                 */
    int32_t node_nf[OARD_MAX_OBJECTS];   /* node_nfs[k] (pos_dim + feature width)   */
    int32_t enc_alias[OARD_MAX_OBJECTS]; /* encoder/decoder index object k uses
                                            (enforce_same_encoding, _base.py:110-113) */
    int32_t condition_nf;
    int32_t condition_time;  /* bool                                                */
    int32_t pos_dim;         /* must be 3                                           */
    float   cutoff;
    int32_t reflect_equiv;   /* LEFTNet reflect_equiv (leftnet.py:268-272, 331, 794-796): 1 = production setting; 0 = the Equi
                                message carries x (x) coord_cross and the edge scalarisation keeps its sign              */
    int32_t precision;       /* OARD_PREC_* bits: arithmetic of the two MFMA edge stages (0 = fp32 everywhere).  A property of
                                the call, not of the process: oard_pack_weights builds the bf16 streams for the bits set, and
                                oard_forward / oard_forward_train must be given the blob packed with the same bits. */
} oard_config;

/* Split precision (csrc/oard_edge_b3.h): every fp32 value as three bf16 terms, six bf16 MFMAs per K block, fp32 accumulation -
 * fp32-grade results (parity <= 1e-5 against the float64 reference like the fp32 kernels).  No counterpart in the reference (which
 * is fp32 torch); the fp32 kernels are the default. */
#define OARD_PREC_GCL_BF16X3 1    /* GCLMessage edge stage of inference calls (throughput launch shape)          */
#define OARD_PREC_EQUI_BF16X3 2   /* EquiMessage edge stage of inference calls                                   */
#define OARD_PREC_TRAIN_BF16X3 4  /* both edge stages of the training-mode forward (the backward stays fp32)      */

/* Library / ABI version (major*1000 + minor). */
int oard_version(void);

/* 0 if this build has kernels for (hidden, num_radial) and the switches in cfg, else OARD_EINVAL. */
int oard_supported(const oard_config* cfg);

/* ---- weights --------------------------------------------------------------------------------
 * Replaces: nn.Module parameter storage of EGNNDynamics (state-dict layout in
 * oareactdiff_amd/spec.py:state_spec == reference state_dict()).  `params_dev[i]` is the device
 * pointer of the i-th tensor in that canonical order (buffers included; unused tensors may be
 * NULL).  Packs every Linear into MFMA-fragment order (16x16 chunks) plus the
 * padded bias / LayerNorm vectors and the constant inter-object edge row.  Call again after
 * any weight update. */
size_t oard_param_count(const oard_config* cfg);
size_t oard_packed_bytes(const oard_config* cfg);
int oard_pack_weights(const oard_config* cfg, const float* const* params_dev, size_t n_params,
                      void* packed_dev, size_t packed_bytes, oard_stream_t stream);

/* ---- topology -------------------------------------------------------------------------------
 * Replaces: what get_edges_index / get_subgraph_mask / compute_frag_index derive from
 * (combined_mask, n_frag_switch) (oa_reactdiff/utils/_graph_tools.py:9-59,
 * egnn_dynamics.py:177-182).  Built once per sample() / batch; host arrays in, device index
 * tables out.  Nodes are given in the reference's order (object-major). */
typedef struct oard_topology oard_topology;

int oard_topology_create(const oard_config* cfg, const int64_t* combined_mask_host,
                         const int64_t* n_frag_switch_host, int64_t n_nodes, oard_topology** out);
/* As oard_topology_create with an explicit number of concurrent sub-batches (0 = the library's choice, see
 * "parts" below).  Training needs 1: the tape and the backward kernels work on one sub-batch. */
int oard_topology_create_parts(const oard_config* cfg, const int64_t* combined_mask_host,
                               const int64_t* n_frag_switch_host, int64_t n_nodes, int parts, oard_topology** out);
void oard_topology_destroy(oard_topology* topo);
int64_t oard_topology_num_nodes(const oard_topology* topo);
int64_t oard_topology_num_edges(const oard_topology* topo);        /* sum n_s (n_s - 1)         */
int64_t oard_topology_num_inner_edges(const oard_topology* topo);  /* same-object ordered pairs */
int64_t oard_topology_num_samples(const oard_topology* topo);
/* Writes 1 to *ok_dev iff edge_index_dev ([2,E] int64, row-major, reference node ids) is the edge SET
 * get_edges_index(combined_mask, remove_self_edge=True) would produce for this topology (utils/_graph_tools.py:30-36),
 * in any order: every ordered pair of distinct nodes of one sample exactly once.  EGNNDynamics.forward accepts any edge_index
 * (egnn_dynamics.py:63-72); its outputs are per node, so any ordering of the complete set is the same computation.
 * Synchronises the stream (once per topology). */
int oard_topology_check_edge_index(const oard_topology* topo, const int64_t* edge_index_dev,
                                   int64_t n_edges, int32_t* ok_dev, oard_stream_t stream);

/* ---- forward --------------------------------------------------------------------------------
 * Replaces: EGNNDynamics.forward(xh, edge_index, t, conditions, n_frag_switch, combined_mask)
 * (egnn_dynamics.py:63-168) including LEFTNet.forward, for update_pocket_coords=True,
 * edge_attr=None.  xh_dev[k] / out_dev[k]: [n_k, node_nf[k]].  t_dev: [B] (one per sample) or,
 * if t_is_scalar, [1].  conditions_dev: [B, condition_nf] (may be NULL when condition_nf==0).
 * status_dev[0] is set to 1 if the predicted displacement contains a NaN
 * (egnn_dynamics.py:138-143 — the caller applies the randn replacement). */
size_t oard_workspace_bytes(const oard_config* cfg, const oard_topology* topo);
int oard_forward(const oard_config* cfg, const oard_topology* topo, const void* packed_dev,
                 const float* const* xh_dev, const float* t_dev, int t_is_scalar,
                 const float* conditions_dev, float* const* out_dev,
                 void* workspace_dev, size_t workspace_bytes, int32_t* status_dev,
                 oard_stream_t stream);

/* Inner (same-object) edges that were inside the cutoff in the LAST inference-mode oard_forward on this workspace - the rows
 * EquiMessage actually ran (model/leftnet.py:748-753: the others carry an exactly-zero message and are skipped; equal to
 * oard_topology_num_inner_edges while nothing is masked).  A measurement aid (bench.py charges the roofline with it):
 * synchronises the stream and copies one int per sub-batch.  -1 if the forward did not build the list (training mode, debug
 * option equi_skip = 0). */
int oard_active_inner_edges(const oard_config* cfg, const oard_topology* topo, const void* workspace_dev, size_t workspace_bytes,
                            int64_t* n_active_host, oard_stream_t stream);

/* EquiUpdate's frame-scalar MLP (model/leftnet.py:333, lin3: 3 -> 48 -> 8 -> 1 on (x, 0, 0)) is evaluated from a table that
 * oard_pack_weights builds and CHECKS per layer (csrc/oard_layout.h: L3T_*): this copies layer `layer`'s table - OARD_L3T_FLOATS floats:
 * (f, f' h) at the 1025 grid points of [-16, 16], then [flag, worst midpoint deviation, range of f, -] - to out_dev (tests). */
#define OARD_L3T_FLOATS 2054
int oard_debug_lin3u_table(const oard_config* cfg, const void* packed_dev, int layer, float* out_dev, oard_stream_t stream);

/* ---- sampler step (next row N1) ----------------------------------------------------------------
 * Replaces the element-wise part of EnVariationalDiffusion.sample_p_zs_given_zt / sample_normal /
 * sample_p_xh_given_z0 (oa_reactdiff/diffusion/en_diffusion.py:562-702) and the CoM-free noise of
 * sample_combined_position_feature_noise (:278-305), per object k, rows in the reference's order:
 *   eps      = [ noise_pos - mean_group(noise_pos) | zero_feature_noise ? 0 : noise_feat ]
 *   mode 0:  out = z / a - eps_hat * b + c * eps ;  out_pos -= mean_group(out_pos)     (a = alpha_t|s, b, c = sigma)
 *   mode 1:  out = a * (z - b * eps_hat) + c * eps                                     (a = 1/alpha_0, b = sigma_0, c = sigma_x)
 *   mode 2:  out = eps                                                                 (initial z_T)
 *   mode 3:  out = a * z + c * eps                      (q(z_s | x): noised_representation, :269-287; a = alpha_s, c = sigma_s)
 *   mode 4:  out = a * z + c * eps ;  out_pos -= mean_group(out_pos)   (sample_p_zt_given_zs, :1050-1074; a = alpha_t|s, c = sigma_t|s)
 *   h0_dev[k] != NULL: the feature columns of out are overwritten by h0 (pos_only sampling, :526-530).
 * group = nodes of one object in one sample.  The scalars are host values (the schedule is a host table),
 * so a sampling loop built on this and oard_forward never synchronises. */
int oard_sampler_step(const oard_config* cfg, const oard_topology* topo, int mode,
                      const float* const* z_dev, const float* const* eps_hat_dev, const float* const* noise_dev,
                      const float* const* h0_dev, float a, float b, float c, int zero_feature_noise,
                      float* const* out_dev, oard_stream_t stream);

/* As oard_sampler_step with the three scalars (a, b, c) read from device memory (coef_dev[3]): the launch is then
 * identical for every step, so one captured hipGraph of [oard_forward, oard_sampler_step_dev] replays the whole loop
 * (the host only advances a device-side step index that selects the row of the schedule table). */
int oard_sampler_step_dev(const oard_config* cfg, const oard_topology* topo, int mode,
                          const float* const* z_dev, const float* const* eps_hat_dev, const float* const* noise_dev,
                          const float* const* h0_dev, const float* coef_dev, int zero_feature_noise,
                          float* const* out_dev, oard_stream_t stream);

/* ---- introspection (tests / profiling) --------------------------------------------------------
 * Copies an intermediate tensor of the LAST oard_forward on this workspace into dst_dev, in the
 * reference's node / edge order, dense [rows, cols] fp32.  `which`: see OARD_TAP_*.  `layer`
 * selects nothing yet (taps hold the state after the final layer) and must be 0. */
#define OARD_TAP_S 1          /* [N, H]      node scalars s after the last layer            */
#define OARD_TAP_VEC 2        /* [N, 3*H]    node vectors after the last layer              */
#define OARD_TAP_EDGE 3       /* [E, 3H+R]   edge state after the last layer                */
#define OARD_TAP_POS_FRAME 4  /* [N, 3]                                                     */
#define OARD_TAP_DPOS 5       /* [N, 3]      LEFTNet displacement before CoM removal        */
#define OARD_TAP_HOUT 6       /* [N, in_hidden]                                             */
#define OARD_TAP_LABELS 7     /* [N, 1]      frame-group label (as float, reference-order id of the group's first node) */
#define OARD_TAP_NE1 8         /* [N, 3*H]    CFConvS2V output (leftnet.py:791)              */
int oard_tap(const oard_config* cfg, const oard_topology* topo, const void* workspace_dev,
             int which, int layer, float* dst_dev, oard_stream_t stream);

/* Debug: make oard_forward return right after a stage so the taps expose intermediate state.
 * 0 = run everything (default); 1 = after the init stages (s = NeighborEmb output, edge state =
 * initial edgeweight); 100 + 10*l + 1 = after layer l's GCL node update; 100 + 10*l + 2 = after
 * layer l's EquiUpdate. */
int oard_debug_stop_after(int code);
/* Process-wide switches for tests and A/B measurements (defaults in brackets; all variants compute the same
 * function and have their own parity test):
 *   "gcl_variant" [2]   GCLMessage edge kernel: 0 = weights straight from L2 (v0, the simple cross-check kernel),
 *                       2 = LDS-streamed, 8 waves x 16 edges, 3 = the same with 4 waves (small launches)
 *                       (oard_edge_v1.h), 6 = latency kernel (oard_edge_small.h)
 *   "equi_variant" [2]  EquiMessage edge kernel: 0 = v0, 2 / 1 = LDS-streamed with 8 / 4 waves, 4 = latency kernel
 *   "node_variant" [1]  0 = one wave per 16 nodes (v0), 1 = one wave per hidden tile (oard_node_v1.h)
 *   "gcl_skip" [1]      skip S1 / S3 of the GCL chain on inter-object rows of the first / last layer
 *   "auto_small" [4], "auto_tiny" [8]  launch-shape heuristics: launches of <= 1024*auto_small (GCL) /
 *                       512*auto_small (Equi) 16-edge tiles use 4-wave workgroups, launches of <= 512*auto_tiny
 *                       tiles the latency kernels; 0 = always the throughput shape
 *   "npb" [0]           real nodes per workgroup of the node stages (0 = ceil(N/256) clamped to 1..16); read by
 *                       oard_topology_create
 *   "parts" [0]         sub-batches per topology (0 = 4 for B >= 32, 2 for B >= 16, else 1); read by
 *                       oard_topology_create;  "sequential" [0] = 1 runs them one after the other
 *   "poison" [0]        1 fills the workspace with NaN bit patterns before every forward (the tests use it to
 *                       prove that nothing depends on workspace contents)
 *   "wgrad_wgs" [512]   workgroups per weight-gradient GEMM (oard_wgrad): row chunks x task groups
 *   "small_split" [128] EquiMessage latency kernel: launches of <= this many 16-edge tiles run one launch per dense stage
 *                       (k_equi_small_s1 / _s2, a tile spread over several CUs); 0 = never */
int oard_debug_option(const char* name, int value);

/* Average duration (ms) and launch count per kernel family since the last reset, measured with
 * HIP events on the launch stream when timing is enabled (bench.py's roofline leg).
 * names: "gcl_edge", "equi_edge", "node", "init", "other" (forward), "gcl_edge_bwd", "equi_edge_bwd", "wgrad" (backward). */
int oard_timing_enable(int on);
int oard_timing_reset(void);
int oard_timing_get(const char* family, double* total_ms, int64_t* launches);

/* ---- training (next row N2): tape, backward of the edge stages ---------------------------------------
 * Replaces: torch autograd through GCLMessage / EquiMessage when DDPMModule.training_step
 * (oa_reactdiff/trainer/pl_trainer.py:327-347) back-propagates the loss of EnVariationalDiffusion.forward
 * (oa_reactdiff/diffusion/en_diffusion.py:56-248).  The reference has no hand-written backward.
 * All of this works on topologies created with parts == 1.
 *
 * oard_forward_train = oard_forward that additionally fills `tape_dev` (oard_tape_bytes) with what the
 * backward pass needs: the edge state entering every layer, the pre-activations of the edge MLPs, the node
 * state at the layer boundaries and the geometry constants.  oard_tape_entry locates one tensor in the tape
 * (byte offset, rows, floats per row); rows are in the library's internal order: nodes sample-major
 * (oard_topology_export NODE_REF maps them to the reference's rows), edges as physical rows (inner edges
 * first, target-sorted == INNER_SRC/INNER_TGT, then inter-object edges; ROW_SRC/ROW_TGT), feature widths
 * padded to multiples of 16 with zero pads (H -> HP, 3H+R -> WP, ...). */
#define OARD_TOPO_NODE_REF 1     /* [N] internal node -> row in the reference's (object-major) order            */
#define OARD_TOPO_NODE_OBJ 2     /* [N] object index                                                          */
#define OARD_TOPO_NODE_ROW 3     /* [N] row inside xh[object]                                                 */
#define OARD_TOPO_NODE_SAMPLE 4  /* [N] dense sample index                                                    */
#define OARD_TOPO_NODE_TIDX 5    /* [N] combined_mask value (row of t / conditions)                           */
#define OARD_TOPO_SAMPLE_PTR 6   /* [B+1] node range of every sample                                          */
#define OARD_TOPO_GROUP_PTR 7    /* [B*n_obj+1] node range of every (sample, object) group                    */
#define OARD_TOPO_INNER_SRC 8    /* [A] source node of inner edge a (rows sorted by target)                   */
#define OARD_TOPO_INNER_TGT 9    /* [A]                                                                       */
#define OARD_TOPO_ROW_SRC 10     /* [E] source node of physical edge row r                                    */
#define OARD_TOPO_ROW_TGT 11     /* [E]                                                                       */
int oard_topology_export(const oard_topology* topo, int which, int32_t* dst_dev, int64_t capacity,
                         oard_stream_t stream);

#define OARD_TAPE_HIN 1      /* [N][16]      encoder output | t | conditions  (LEFTNet input h)              */
#define OARD_TAPE_GEO 2      /* [A+1][12]    per inner edge: d, env, u[3], c[3], v[3], mask                   */
#define OARD_TAPE_RBF 3      /* [A+1][RP]    radial basis                                                     */
#define OARD_TAPE_PP0 4      /* [N][1]       pos_prjt[:, 0]                                                   */
#define OARD_TAPE_X1 5       /* [N][3]       node frame axis x1                                               */
#define OARD_TAPE_S_IN 16    /* [N][HP]      s entering layer l          (l = L: after the last layer)        */
#define OARD_TAPE_VEC_IN 17  /* [N][3][HP]   vec entering layer l        (l = L: after the last layer)        */
#define OARD_TAPE_EW 18      /* [E+1][WP]    edge state entering layer l (l = L: final; inter-object rows of l = 0
                                             are the constant row and are not materialised)                   */
#define OARD_TAPE_AGG 19     /* [N][HP]      mean gated message per source node                               */
#define OARD_TAPE_S_MID 20   /* [N][HP]      s after the GCL node update                                      */
#define OARD_TAPE_Z1 21      /* [E+1][HP]    edge_mlp.0 pre-activation                                        */
#define OARD_TAPE_Z2 22      /* [E+1][HP]    edge_mlp.1 pre-activation                                        */
#define OARD_TAPE_ATT 23     /* [E+1][1]     att_mlp pre-activation                                           */
#define OARD_TAPE_Z3 24      /* [E+1][WP]    edge_out_trans pre-activation                                    */
#define OARD_TAPE_ZD1 25     /* [A+1][D1P]   dir_proj.0 pre-activation                                        */
#define OARD_TAPE_CD 26      /* [A+1][3][HP] dir_proj output (before the product with rbf_proj)               */
#define OARD_TAPE_S_A 27     /* [N][HP]      s after the EquiMessage aggregation, (s + dx)/sqrt2 (EquiUpdate input) */
#define OARD_TAPE_VEC_A 28   /* [N][3][HP]   vec after the EquiMessage aggregation (EquiUpdate input)          */
size_t oard_tape_bytes(const oard_config* cfg, const oard_topology* topo);
int oard_tape_entry(const oard_config* cfg, const oard_topology* topo, int which, int layer,
                    size_t* offset_bytes, int64_t* rows, int64_t* row_floats);
int oard_forward_train(const oard_config* cfg, const oard_topology* topo, const void* packed_dev,
                       const float* const* xh_dev, const float* t_dev, int t_is_scalar,
                       const float* conditions_dev, float* const* out_dev,
                       void* workspace_dev, size_t workspace_bytes, void* tape_dev, size_t tape_bytes,
                       int32_t* status_dev, oard_stream_t stream);

/* Transposed weight streams of the two backward edge kernels (same params_dev convention as
 * oard_pack_weights); call again after any weight update. */
size_t oard_packed_bwd_bytes(const oard_config* cfg);
int oard_pack_weights_bwd(const oard_config* cfg, const float* const* params_dev, size_t n_params,
                          void* packed_bwd_dev, size_t packed_bytes, oard_stream_t stream);

/* Adjoint of layer `layer`'s GCLMessage edge part (leftnet.py:162-170 + the mean aggregation :172-180).
 *   in : dagg_dev [N][HP]    gradient w.r.t. the per-node mean message (pads zero)
 *        dew_dev  [E+1][WP]  gradient w.r.t. the edge state LEAVING the layer (pads and spare row finite)
 *   out: dew_dev             gradient w.r.t. the edge state ENTERING the layer (in place)
 *        dz3_dev [E+1][WP], dz2_dev / dz1_dev [E+1][HP], da_dev [E+1]: gradients w.r.t. the pre-activations
 *        of edge_out_trans, edge_mlp.1, edge_mlp.0, att_mlp;  mout_dev [E+1][HP]: the gated message m,
 *        all in physical row order - the operands of the weight-gradient GEMMs (oard_wgrad).
 *   Last layer: dz3 is produced for the inner rows [0, A) only (the forward skips edge_out_trans on
 *   inter-object rows there). */
int oard_gcl_backward_dx(const oard_config* cfg, const oard_topology* topo, const void* packed_bwd_dev, int layer,
                         const void* tape_dev, const float* dagg_dev, float* dew_dev, float* dz3_dev,
                         float* mout_dev, float* dz2_dev, float* da_dev, float* dz1_dev, oard_stream_t stream);
/* dP[n] / dQ[n] = sum of dz1 over the edges whose source / target is node n (the node halves of edge_mlp.0). */
int oard_edge_node_sums(const oard_config* cfg, const oard_topology* topo, const float* dz1_dev,
                        float* dP_dev, float* dQ_dev, oard_stream_t stream);
/* Adjoint of layer `layer`'s EquiMessage edge part (leftnet.py:247-249, dir_proj):
 *   in : dcd_dev [A+1][3][HP] gradient w.r.t. dir_proj's output;  out: dew_dev rows [0, A) += dir_proj.0^T ...,
 *        dzd1_dev [A+1][D1P] gradient w.r.t. dir_proj.0's pre-activation. */
int oard_equi_backward_dx(const oard_config* cfg, const oard_topology* topo, const void* packed_bwd_dev, int layer,
                          const void* tape_dev, const float* dcd_dev, float* dew_dev, float* dzd1_dev,
                          oard_stream_t stream);
/* Adjoint of EquiMessage's message formation + aggregation (leftnet.py:264-283, the gather half of k_equi_node_v1):
 *   in : xq_dev [N][3][H] x_proj output, cr_dev [A][3][H] rbf_proj(rbf), gx_dev [N][H] / gv_dev [N][3][H] gradients w.r.t. the
 *        aggregated scalar / vector messages (dense, unpadded);  vec_in, dir_proj's output and the edge frames come from the tape
 *   out: dcd_dev / dcr_dev [A+1][3][HP] gradients w.r.t. dir_proj's output / rbf_proj's output (pads untouched: pass zeroed
 *        buffers), dxq_dev [N][3][H], dvec_dev [N][3][H] (gradient w.r.t. vec entering the layer, identity path included). */
int oard_equi_msg_backward(const oard_config* cfg, const oard_topology* topo, const void* tape_dev, int layer,
                           const float* xq_dev, const float* cr_dev, const float* gx_dev, const float* gv_dev,
                           float* dcd_dev, float* dcr_dev, float* dxq_dev, float* dvec_dev, oard_stream_t stream);
/* EquiUpdate's frame-scalar MLP lin3 (leftnet.py:304-310, 333: Linear(3,48) SiLU Linear(48,8) SiLU Linear(8,1) on (x, 0, 0)) as a
 * differentiable op on n = N*H items (training; the inference forward fuses it into k_equi_node_v1).
 *   forward : out[i] = lin3(x[i], 0, 0)
 *   backward: dx[i], and per item the operands of the weight gradients for oard_wgrad:
 *             xa [n][4] = (x, 1, 0, 0), h1 [n][48], dz1 [n][48], dz2 [n][8], h2a [n][12] = dout * (h2[8], 1, 0, 0, 0)
 *             (lin3.0: dz1^T xa -> [48][4] = (d weight[:,0] | d bias);  lin3.2: dz2^T h1, bias = column sums of dz2;
 *              lin3.4: column sums of h2a = (d weight[8] | d bias)). */
int oard_lin3u_forward(const oard_config* cfg, const void* packed_dev, int layer, const float* x_dev, int64_t n,
                       float* out_dev, oard_stream_t stream);
int oard_lin3u_backward(const oard_config* cfg, const void* packed_dev, int layer, const float* x_dev, const float* dout_dev,
                        int64_t n, float* dx_dev, float* xa_dev, float* h1_dev, float* dz1_dev, float* h2a_dev,
                        float* dz2_dev, oard_stream_t stream);
/* Adjoint of the edge scalarisation + lin3 (leftnet.py:792-806, k_scalarize): from the gradient of the initial edge state
 * (dew_dev [E+1][WP], columns [0, 2H) of the inner rows) and NE1 (ne1_dev [N][3][ld], ld >= H, the CFConvS2V output) to
 *   dne1_dev [N][3][ld]     gradient w.r.t. NE1
 *   part_dev [N][5*(H/4)+1] per-node partial sums of the lin3 gradients: lin3.0.weight [H/4][3] | lin3.0.bias [H/4] |
 *                           lin3.2.weight [H/4] | lin3.2.bias [1]  (the caller adds the N rows up)
 * packed_dev: oard_pack_weights blob (lin3 weights), tape_dev: the forward's tape (edge frames). */
int oard_scalarize_backward(const oard_config* cfg, const oard_topology* topo, const void* packed_dev, const void* tape_dev,
                            const float* ne1_dev, int ld, const float* dew_dev, float* dne1_dev, float* part_dev,
                            oard_stream_t stream);
/* Weight gradient of a Linear layer from row-major operands (nn.Linear backward, dW = dY^T X, db = sum dY):
 *   dW[o][i] = sum_{r < rows} dY[r][op(o)] * act(X[r][ip(i)]),   db[o] = sum_r dY[r][op(o)]   (db may be NULL)
 * op(o) = (o / o_len) * o_pad + o % o_len undoes section padding (e.g. three 196-wide thirds stored 208 apart);
 * ncY / ncX = readable floats per row (multiples of 4, <= ld);  x_silu != 0 applies SiLU to X on load.
 * Deterministic: fixed chunking and summation order for a given shape. */
size_t oard_wgrad_scratch_bytes(int ncY, int ncX, int64_t rows);
int oard_wgrad(const float* dY_dev, int ldY, int ncY, int o_len, int o_pad, int MO,
               const float* X_dev, int ldX, int ncX, int x_silu, int i_len, int i_pad, int MI, int64_t rows,
               float* dW_dev, float* db_dev, void* scratch_dev, size_t scratch_bytes, oard_stream_t stream);

/* ---- The whole reverse sweep on the device (round 3) -------------------------------------------------------------------------
 * Replaces torch autograd through the node-side / init / output stages of LEFTNet.forward (oa_reactdiff/model/leftnet.py:
 * EquiUpdate :325-346, GCL node update :172-183, x_proj :245, pos_expansion + LayerNorm + edge_mlp.0 node halves :840-841,158,168,
 * output block :566-576,878-891, init head :744,781-809) and of the wrapper (egnn_dynamics.py:91-119 encoders, :137-160 velocity /
 * CoM / decoders).  Every entry reads the tape of oard_forward_train and ACCUMULATES parameter gradients into `grads_dev`:
 * a table of device pointers in the canonical parameter order of oard_pack_weights (nn.Linear shapes; NULL entries are skipped;
 * encoders / decoders shared by several objects simply appear several times).  `params_dev` is the same table for the weights
 * themselves.  Node cotangents are [N][HP] / [3N][HP] float32 with zero pads, the edge-state cotangent is [E+1][WP].  All scratch
 * is the caller's (`oard_train_scratch_bytes`); nothing synchronises, and the library's own job tables (the grouped weight-gradient
 * launches) travel through a ring of pinned / device slots that is allocated once per device, at its first use.  `packed_bwd_dev` = oard_pack_weights_bwd.
 *   oard_train_tail_backward   grad_out[k] ([n_k][node_nf_k], reference rows; NULL = zero) -> ds, dvec of the final node state
 *   oard_train_layer_backward  one layer: ds / dvec / dew in place (cotangents of the layer's outputs -> of its inputs)
 *   oard_train_init_backward   ds0, dew (cotangents of the state entering layer 0) -> init-head and encoder gradients
 *   oard_train_stage_backward  one stage of a layer in isolation (teacher-forced tests); RECOMPUTE first */
#define OARD_STAGE_RECOMPUTE 0
#define OARD_STAGE_UPDATE 1      /* in: ds, dvec          out: gs_a, gvec_a       */
#define OARD_STAGE_MESSAGE 2     /* in: gs_a, gvec_a      out: gx, dxq, dvec_in   */
#define OARD_STAGE_GCL_NODE 3    /* in: gx, dxq           out: dxh, dagg          */
#define OARD_STAGE_NODE_PRE 4    /* in: dxh, dP, dQ       out: ds_in              */
#define OARD_STAGE_GCL_EDGE 5    /* in: dagg              in/out: dew (out0)      out: dP, dQ   (+ every GCLMessage edge-MLP gradient) */
#define OARD_STAGE_EQUI_EDGE 6   /* in: dcd [A+1][3HP]    in/out: dew (out0)      (+ dir_proj gradients)  */
#define OARD_SCRATCH_XH 1
#define OARD_SCRATCH_XQ 2
#define OARD_SCRATCH_CR 3
#define OARD_SCRATCH_DCD 4
#define OARD_SCRATCH_DCR 5
size_t oard_train_scratch_bytes(const oard_config* cfg, const oard_topology* topo);
int oard_train_scratch_poison(const oard_config* cfg, const oard_topology* topo, void* scratch_dev, size_t scratch_bytes,
                              oard_stream_t stream);
int oard_train_scratch_entry(const oard_config* cfg, const oard_topology* topo, int which, size_t* offset_bytes, int64_t* rows,
                             int64_t* row_floats);
int oard_train_tail_backward(const oard_config* cfg, const oard_topology* topo, const void* packed_dev, const void* packed_bwd_dev,
                             const void* tape_dev, const float* const* grad_out_dev, float* ds_dev, float* dvec_dev,
                             const float* const* params_dev, float* const* grads_dev, void* scratch_dev, size_t scratch_bytes,
                             oard_stream_t stream);
int oard_train_layer_backward(const oard_config* cfg, const oard_topology* topo, const void* packed_dev, const void* packed_bwd_dev,
                              const void* tape_dev, int layer, float* ds_dev, float* dvec_dev, float* dew_dev,
                              const float* const* params_dev, float* const* grads_dev, void* scratch_dev, size_t scratch_bytes,
                              oard_stream_t stream);
int oard_train_init_backward(const oard_config* cfg, const oard_topology* topo, const void* packed_dev, const void* packed_bwd_dev,
                             const void* tape_dev, const float* const* xh_dev, const float* ds0_dev, const float* dew_dev,
                             const float* const* params_dev, float* const* grads_dev, void* scratch_dev, size_t scratch_bytes,
                             oard_stream_t stream);
int oard_train_stage_backward(const oard_config* cfg, const oard_topology* topo, const void* packed_dev, const void* packed_bwd_dev,
                              const void* tape_dev, int layer, int stage, const float* in0, const float* in1, const float* in2,
                              float* out0, float* out1, float* out2, const float* const* params_dev, float* const* grads_dev,
                              void* scratch_dev, size_t scratch_bytes, oard_stream_t stream);

/* The reference's NaN guard (egnn_dynamics.py:138-143) without its host sync: if `status_dev[0]` (written by oard_forward) is set,
 * the velocity columns of every out[k] are replaced by noise[k] ([n_k][3] N(0,1) draws of the caller) minus its per-(sample, object)
 * mean; otherwise nothing is written.  Asynchronous on `stream`. */
int oard_nan_replace(const oard_config* cfg, const oard_topology* topo, const int32_t* status_dev, const float* const* noise_dev,
                     float* const* out_dev, oard_stream_t stream);

/* ---- The training caller around the network call, fused (round 3) --------------------------------------------------------------
 * oard_loss_prepare  EnVariationalDiffusion.forward up to the network call (en_diffusion.py:56-123, noised_representation :250-281,
 *                    sample_combined_position_feature_noise :283-306): normalises the dataset-layout batch (pos float32 [n_k][3],
 *                    one_hot int64 [n_k][nf_k-4], charge int64 [n_k][1]; reference rows), removes the per-(sample, object) CoM of the
 *                    raw N(0,1) position noise, z_t = alpha_t x + sigma_t eps with gamma = gamma_table[t_int[b]].
 * oard_loss_terms    the rest of it + DDPMModule.compute_loss (pl_trainer.py:208-282, l2 training form): per-sample nll [B], logged
 *                    terms [2 n_obj][B] (normalised and un-normalised error per object) and d(mean_b nll)/d(network output).
 * oard_adamw_step    torch.optim.AdamW(amsgrad) (pl_trainer.py:150) over one flat bucket, gradient-clipping factor folded in.
 * norm_values / norm_biases [3], scales [n_obj]: host arrays.  t_int [B] float32 (device), gamma table [T+1] (device). */
int oard_loss_prepare(const oard_config* cfg, const oard_topology* topo, const float* const* pos_dev, const int64_t* const* one_hot_dev,
                      const int64_t* const* charge_dev, const float* const* noise_dev, const float* t_int_dev, const float* gamma_dev,
                      int T, const float* norm_values, const float* norm_biases, int pos_only, int fixed_mask, float* const* z_dev,
                      float* const* eps_dev, oard_stream_t stream);
int oard_loss_terms(const oard_config* cfg, const oard_topology* topo, const float* const* eps_dev, const float* const* net_dev,
                    const float* const* z_dev, const int64_t* const* one_hot_dev, const int64_t* const* charge_dev,
                    const float* t_int_dev, const float* gamma_dev, int T, const float* norm_values, const float* norm_biases,
                    const float* scales, int pos_only, int B, float* nll_dev, float* terms_dev, float* const* dnet_dev,
                    oard_stream_t stream);
int oard_adamw_step(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* max_exp_avg_sq_dev,
                    int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, int amsgrad,
                    double grad_scale, oard_stream_t stream);

/* The optimiser step WITHOUT a host round trip (round 4): the adaptive clipping decision of pl_trainer.py:391-418 (max_norm = 1.5 mean +
 * 3 std of the last <= `capacity` gradient norms, utils/training_tools.py:6-23) taken by one device thread, then AdamW with the scalars it
 * derived.  clip_state (device, doubles, 4 + capacity + 8 entries): [0] entries in the history, [1] optimiser steps taken so far, [2] steps
 * skipped so far, [3] reserved, [4 ..) the history, newest first; the caller initialises [0 .. 4 + capacity) and may read it back at any time.
 * grad_norm / flag (device, float): the gradient's 2-norm and the non-finite flag of the step (flag != 0 or a non-finite norm: the step is
 * skipped - parameters, moments, history and step count untouched, [2] counts it).  clip = 0: no clipping (history untouched).
 * out4 (device, float[4]): grad_norm, max_norm (NaN when not clipping / skipped), the clipping factor applied, 1 if skipped. */
int oard_adamw_step_dev(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, float* max_exp_avg_sq_dev,
                        int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay, int amsgrad, int clip,
                        double* clip_state_dev, int capacity, const float* grad_norm_dev, const float* flag_dev, float* out4_dev,
                        oard_stream_t stream);

/* The library's own streams of the current device (three non-blocking streams, chosen once per device so that each runs on a hardware
 * queue of its own beside the caller's stream - the runtime serves a process's streams from 4 queues): 0 = the training sweep's
 * gradient stream (and the second sub-batch of a multi-part forward), 1 = the third sub-batch, 2 = topology uploads (and the fourth).
 * For callers that want a second in-order queue WITHOUT adding a fifth stream to the process: DDPMTrainer's two-micro-batch step runs
 * its second micro-batch on stream 1, which is idle in training.  No counterpart in the reference. */
int oard_library_stream(int which, oard_stream_t* out);

/* ---- general edge lists (round 6) --------------------------------------------------------------------------------------------
 * Replaces: EGNNDynamics.forward on an edge_index that is NOT the complete graph per sample.  The reference accepts any edge list
 * (dynamics/egnn_dynamics.py:63-72), builds incomplete ones with get_edges_index(..., edge_cutoff=) (utils/_graph_tools.py:31-33, plumbed
 * through trainer/pl_trainer.py:68,94 and dynamics/_base.py:59), and its model tests run disconnected and cut graphs
 * (tests/model/test_equiv.py:177-230, tests/model/test_subgraphs.py:285-339).  Production never leaves the complete graph
 * (trainer/train_ts1x.py:106), so this path is built for parity first (csrc/oard_general.h): explicit edge list, CSR gathers in
 * edge order, the reference's literal node frame (leftnet.py:812-834) on float64 geometry, float64 accumulation, raw (unpacked) parameters.
 * Its dense layers run on the float64 matrix pipe (k_general_gemm_f64, csrc/oard_general.hip); the environment variable
 * OARD_GENERAL_GEMM=threads, read on every call, runs them on plain threads instead (the cross-check form; same results, ~18 x the time).
 * Inference only.
 *
 * oard_graph_create: host arrays in (combined_mask / n_frag_switch as for oard_topology_create; edge_index [2, E] row-major, reference
 * node ids; self loops, duplicates and edges across samples are taken as given, as the reference does), device tables out.
 * oard_graph_is_complete: 1 iff the edge SET is get_edges_index(combined_mask, remove_self_edge=True) in any order - the caller may then
 * use oard_forward instead.  oard_graph_forward: as oard_forward, with the canonical parameter pointers (oard_pack_weights' order) in
 * place of the packed blob; asynchronous on `stream`; the workspace is the caller's and carries everything the call needs. */
typedef struct oard_graph oard_graph;
int oard_graph_create(const oard_config* cfg, const int64_t* combined_mask_host, const int64_t* n_frag_switch_host, int64_t n_nodes,
                      const int64_t* edge_index_host, int64_t n_edges, oard_graph** out);
void oard_graph_destroy(oard_graph* graph);
int64_t oard_graph_num_nodes(const oard_graph* graph);
int64_t oard_graph_num_edges(const oard_graph* graph);
int64_t oard_graph_object_rows(const oard_graph* graph, int object);   /* rows of xh[object] */
int oard_graph_is_complete(const oard_graph* graph);
size_t oard_graph_workspace_bytes(const oard_config* cfg, const oard_graph* graph);
int oard_graph_forward(const oard_config* cfg, const oard_graph* graph, const float* const* params_dev, size_t n_params,
                       const float* const* xh_dev, const float* t_dev, int t_is_scalar, const float* conditions_dev,
                       float* const* out_dev, void* workspace_dev, size_t workspace_bytes, int32_t* status_dev, oard_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OARD_H */
