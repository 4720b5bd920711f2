// oard_train_stages.h — the backward sweep of one denoising call, stage by stage, on the device (training path, row N2).
//
// Included by oard_hip.hip (uses its layout structs, Packer, wgrad_impl and the edge-stage backward launchers).
// What the reference does instead: torch autograd through LEFTNet.forward (oa_reactdiff/model/leftnet.py:724-891).  Here every
// stage of the forward has a hand-written adjoint that reads the training tape (TapeOff), re-evaluates the stage's cheap hidden
// activations from the taped stage inputs and writes (a) the cotangent of the stage inputs and (b) the parameter gradients,
// ACCUMULATED into a caller-owned table of nn.Linear-shaped buffers in the canonical parameter order (ParamIdx):
//
//   tr_recompute     k_node_pre_v1 + k_gcl_node_v1 re-evaluated: xh, node-MLP hidden, xq; rbf_proj(rbf)     leftnet.py:840-841,158,172-183,245
//   tr_update_bwd    adjoint of EquiUpdate (second half of k_equi_node_v1)                                    leftnet.py:325-346, 861-864
//   tr_msg_bwd       adjoint of the EquiMessage gather half (k_equi_msg_bwd) + rbf_proj gradient             leftnet.py:264-283, 857-859
//   tr_gcl_node_bwd  adjoint of the GCL node update + x_proj (k_gcl_node_v1)                                  leftnet.py:172-183, 245
//   tr_equi_edge_bwd / tr_gcl_edge_bwd   the MFMA edge kernels of oard_edge_bwd.h + their weight-gradient GEMMs leftnet.py:162-170, 247-249
//   tr_pre_bwd       adjoint of pos_expansion + LayerNorm + the node halves of edge_mlp.0 (k_node_pre_v1)     leftnet.py:840-841, 158, 168
//   tr_tail_bwd      adjoint of the output block, the velocity / CoM epilogue and the decoders               leftnet.py:566-576, 878-891; egnn_dynamics.py:137-160
//   tr_init_bwd      adjoint of the init head (embedding, NeighborEmb, radial_lin, S2V, scalarisation + lin3) and the encoders
//                                                                                                             leftnet.py:744, 781-809; egnn_dynamics.py:91-119
// Node buffers are [N][HP] (vec: [3 N][HP], row 3 n + x), pads exactly 0 (oard_rows.h).
#pragma once
#include "oard_rows.h"

// ---- scratch of the sweep (caller-owned, oard_train_scratch_bytes): byte offsets ---------------------------------------------------
struct TrainWs {
    // recompute of the current layer (tr_recompute), read by the later stages of the same layer
    size_t hid, s1, xh, zm, hm, xln, zq, hq, xq, cr;
    // EquiUpdate adjoint
    size_t v12, sc, vdot, scal, zx, hx, cvec, dabc, dzx, dscal, dsc, dv12, l3_h1;      // l3_h1: per-wave partial blocks of k_lin3u_bwd_fused
    // message adjoint / edge stages
    size_t gx, gs_a, gvec_a, dxq, dvec_in, dcd, dcr, dzd1, dz3, mout, dz2, dz1, da, dPQ, dagg;
    // GCL node / pre adjoint
    size_t dzq, dxln, lng, dsm, dzm, dxh, t1, t2, dhid, dzh;
    size_t gatep;              // per-wave partial sums of the att_mlp gradients, written by k_gcl_edge_bwd ([E / 16 + 32][HP])
    size_t lng2, dsn;          // node_pre adjoint: its own LayerNorm product buffer, cotangent of s entering the layer (copy read by the gradient stream)
    // column sums, weight-gradient partials (used by the gradient stream only)
    size_t csum, cpart, wg, wgq;
    size_t wg_bytes, wgq_bytes, total;
    size_t layer_bytes;        // everything above lives twice: buffer set l & 1 for layer l (the gradient stream of layer l runs beside the cotangent chain of layer l - 1)
};
static TrainWs make_train_ws(const oard_config* c, const TopoDev& td) {
    const RDims d(c->hidden, c->num_radial);
    const size_t N = td.N, E = td.E + 1, A = td.A + 1, HP = d.HP, F = sizeof(float);
    TrainWs w;
    memset(&w, 0, sizeof(w));
    size_t cur = 0;
    auto take = [&](size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; };
    w.hid = take(N * d.PP * F); w.s1 = take(N * HP * F); w.xh = take(N * HP * F); w.zm = take(N * HP * F); w.hm = take(N * HP * F);
    w.xln = take(N * HP * F); w.zq = take(N * HP * F); w.hq = take(N * HP * F); w.xq = take(N * 3 * HP * F); w.cr = take(A * 3 * HP * F);
    w.v12 = take(3 * N * 2 * HP * F); w.sc = take(N * HP * F); w.vdot = take(N * HP * F); w.scal = take(N * HP * F);
    w.zx = take(N * HP * F); w.hx = take(N * HP * F); w.cvec = take(N * HP * F); w.dabc = take(N * 3 * HP * F); w.dzx = take(N * HP * F);
    w.dscal = take(N * HP * F); w.dsc = take(N * HP * F); w.dv12 = take(3 * N * 2 * HP * F);
    const size_t items = N * HP;
    w.l3_h1 = take((size_t)512 * 4 * L3F_PART * F);
    w.gx = take(N * HP * F); w.gs_a = take(N * HP * F); w.gvec_a = take(3 * N * HP * F); w.dxq = take(N * 3 * HP * F);
    w.dvec_in = take(3 * N * HP * F); w.dcd = take(A * 3 * HP * F); w.dcr = take(A * 3 * HP * F); w.dzd1 = take(A * d.D1P * F);
    w.dz3 = take(E * d.WP * F); w.mout = take(E * HP * F); w.dz2 = take(E * HP * F); w.dz1 = take(E * HP * F); w.da = take(E * F);
    w.dPQ = take(2 * N * HP * F); w.dagg = take(N * HP * F);
    w.dzq = take(N * HP * F); w.dxln = take(N * HP * F); w.lng = take(N * HP * F); w.dsm = take(N * HP * F); w.dzm = take(N * HP * F);
    w.dxh = take(N * HP * F); w.t1 = take(N * HP * F); w.t2 = take(N * HP * F); w.dhid = take(N * d.PP * F); w.dzh = take(N * d.PP * F);
    w.lng2 = take(N * HP * F); w.dsn = take(N * HP * F);
    w.gatep = take((E / 16 + 32) * HP * F);
    w.layer_bytes = cur;
    cur = 2 * w.layer_bytes;
    w.csum = take(4096 * F);
    w.cpart = take(std::max((size_t)cdiv(std::max<size_t>(E, 3 * N), CS_ROWS) * std::max<size_t>(d.WP, 1024),
                            (size_t)cdiv(N * HP, CS_ROWS) * 16) * F);
    // partials of the weight-gradient GEMMs: the largest requirement over every (dY width, X width, rows) shape of the sweep (the
    // chunk count of oard_wgrad's plan is not monotone in the widths: narrow operands are cut into more row chunks)
    {
        const int HPi = d.HP, WPi = d.WP, D1 = d.D1P, RPi = d.RP, PPi = d.PP;
        const int64_t Er = (int64_t)E, Ar = (int64_t)A, Nr = (int64_t)N, Ir = (int64_t)items;
        const struct { int y, x; int64_t r; } shapes[] = {
            {WPi, HPi, Er}, {HPi, HPi, Er}, {HPi, WPi, Er},                                             // GCLMessage edge MLP
            {3 * HPi, D1, Ar}, {D1, WPi, Ar}, {3 * HPi, RPi, Ar}, {HPi, HPi, Ar}, {HPi, RPi, Ar},       // dir_proj, rbf_proj, radial_lin
            {3 * HPi, HPi, Nr}, {HPi, HPi, Nr}, {2 * HPi, HPi, 3 * Nr}, {HPi, HPi, 3 * Nr}, {HPi, 16, Nr}, {16, HPi, Nr}, {HPi, PPi, Nr},
            {32, 16, Nr}, {16, 32, Nr}};
        (void)Ir;
        w.wg_bytes = 0;
        for (const auto& sh : shapes) w.wg_bytes = std::max(w.wg_bytes, oard_wgrad_scratch_bytes(sh.y, sh.x, sh.r));
    }
    w.wg = take(w.wg_bytes);
    {   // partials of the QUEUED (node-level, grouped-launch) products of one layer, each with its own slice; a product that does not
        // fit is launched on its own instead (wgrad_impl), so this is a performance knob, not a correctness bound
        const int HPi = d.HP, RPi = d.RP, PPi = d.PP;
        const int64_t Ar = (int64_t)A, Nr = (int64_t)N;
        const struct { int y, x; int64_t r; int n; } shapes[] = {{3 * HPi, HPi, Nr, 2}, {HPi, HPi, Nr, 9}, {2 * HPi, HPi, 3 * Nr, 1},
                                                                 {3 * HPi, RPi, Ar, 1}, {HPi, PPi, Nr, 1}};
        size_t sum = 0;
        for (const auto& sh : shapes) sum += sh.n * align_up(oard_wgrad_scratch_bytes(sh.y, sh.x, sh.r), 256);
        w.wgq_bytes = sum + sum / 4;
    }
    w.wgq = take(w.wgq_bytes);
    w.total = cur;
    return w;
}

// ---- the two streams of the sweep -------------------------------------------------------------------------------------------------------
// The cotangent chain (dX kernels) is a sequence of dependent launches, many of them small (276 workgroups of 16 nodes); the weight
// gradients (GEMMs over the rows, reduces, column sums) only CONSUME what the chain leaves behind.  They run on a second stream, forked
// by an event after each stage: a layer's weight-gradient GEMMs fill the chip while the next layer's node-side adjoints trickle through.
// The per-layer scratch exists twice (buffer set = layer & 1) and the chain waits for layer l + 2's gradients before it starts layer l.
struct TrainStreams {
    hipStream_t w = nullptr;
    hipEvent_t pool[32];
    hipEvent_t layer_done[OARD_MAX_LAYERS + 1];
    unsigned next = 0;
    bool ok = false, tried = false;
};
int g_train_dual = 1;           // oard_debug_option("train_dual", 0): everything on the caller's stream
static TrainStreams& train_streams() {
    static TrainStreams all[64];
    int d = 0;
    (void)hipGetDevice(&d);
    TrainStreams& t = all[d & 63];
    if (!t.tried) {
        t.tried = true;
        t.w = device_streams()[0];                     // the library's first stream (oard_hip.hip: device_streams)
        bool ok = t.w != nullptr;
        for (int i = 0; ok && i < 32; ++i) ok = hipEventCreateWithFlags(&t.pool[i], hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i <= OARD_MAX_LAYERS; ++i) ok = hipEventCreateWithFlags(&t.layer_done[i], hipEventDisableTiming) == hipSuccess;
        t.ok = ok;
    }
    return t;
}

// ---- context of one sweep ------------------------------------------------------------------------------------------------------------
struct TrainCtx {
    const oard_config* c;
    const TopoDev* tp;
    const float* wb;            // packed weights (oard_pack_weights)
    PackOff po;
    const float* pb;            // packed transposed weights (oard_pack_weights_bwd)
    BwdOff bo;
    NodeBwdOff nb;
    const char* tape;
    TapeOff to;
    char* ws;                   // scratch
    TrainWs w;
    const float* const* params; // the module's parameters, canonical order (raw nn.Linear layouts), or nullptr
    float* const* grads;        // canonical parameter order; nullptr entries are skipped
    ParamIdx pi;
    hipStream_t st;             // the cotangent chain (the caller's stream)
    hipStream_t stw;            // weight gradients: reduces, column sums, weight-gradient GEMMs (== st when the split is off)
    int par = 0;                // buffer set of the per-layer scratch (layer & 1)
    TrainCtx(const oard_config* c_, const TopoDev* tp_, const void* packed, const void* packed_bwd, const void* tape_, void* ws_,
             const float* const* params_, float* const* grads_, hipStream_t st_)
        : c(c_), tp(tp_), wb((const float*)packed), po(make_layout(c_)), pb((const float*)packed_bwd), bo(make_bwd_layout(c_)),
          nb(make_node_bwd_layout(c_, bo.total)), tape((const char*)tape_), to(make_tape(c_, *tp_)), ws((char*)ws_),
          w(make_train_ws(c_, *tp_)), params(params_), grads(grads_), pi(c_), st(st_), stw(st_) {
        TrainStreams& ts = train_streams();
        if (g_train_dual && ts.ok) stw = ts.w;
    }
    bool dual() const { return stw != st; }
    // the gradient stream may start on what the chain has enqueued so far
    void fork() const {
        if (!dual()) return;
        TrainStreams& ts = train_streams();
        hipEvent_t e = ts.pool[ts.next++ & 31];
        (void)hipEventRecord(e, st);
        (void)hipStreamWaitEvent(stw, e, 0);
    }
    // the chain waits for everything the gradient stream has been given
    void join() const {
        if (!dual()) return;
        TrainStreams& ts = train_streams();
        hipEvent_t e = ts.pool[ts.next++ & 31];
        (void)hipEventRecord(e, stw);
        (void)hipStreamWaitEvent(st, e, 0);
    }
    // layer l's gradient work is complete on the device when ts.layer_done[l] has fired; the chain may not start writing buffer
    // set l & 1 again (layer l - 2) before that
    void layer_done(int l) const { if (dual()) (void)hipEventRecord(train_streams().layer_done[l], stw); }
    void wait_layer(int l) const { if (dual() && l < c->num_layers) (void)hipStreamWaitEvent(st, train_streams().layer_done[l], 0); }
    TrainCtx on_grad_stream() const { TrainCtx y = *this; y.st = stw; return y; }        // element-wise / dense kernels that only serve a gradient
    float* f(size_t off) const { return (float*)(ws + off + (off < w.layer_bytes ? (size_t)par * w.layer_bytes : 0)); }
    const float* t(size_t off) const { return (const float*)(tape + off); }
    float* g(int idx) const { return grads ? grads[idx] : nullptr; }
};

#ifndef OARD_ROWS_LONG
#define OARD_ROWS_LONG 1
#endif
// Y = epi(W X + b) on rows; KB is a template parameter of the kernel, dispatched over the block counts the model has
template <int KB, int EPI>
static void rows_dense_launch(hipStream_t st, const RowsDense& a) {
    if (a.rows <= 0 || a.MT <= 0) return;
    ScopedLaunch sl_(F_NODE, st);
    if constexpr (KB <= 13) {
        if (OARD_ROWS_LONG && a.rows >= 16384) {                  // long inputs (inner-edge rows): 4 row groups per workgroup share the weight loads
            hipLaunchKernelGGL((k_rows_dense_long<KB, EPI, 8, 4>), dim3((unsigned)cdiv(a.rows, 64)), dim3(512), 0, st, a);
            return;
        }
    }
    hipLaunchKernelGGL((k_rows_dense<KB, EPI, 8>), dim3((unsigned)cdiv(a.rows, 16)), dim3(512), 0, st, a);
}
template <int KB, int EPI = EPI_NONE>
static void rows_dense(const TrainCtx& x, long long rows, const float* X, int ldx, const float* W, int MT, float* Y, int ldy,
                       const float* bias = nullptr, const float* X2 = nullptr, int ldx2 = 0, int KB1 = KB, const float* Z = nullptr,
                       int ldz = 0, float* Zo = nullptr, int ldzo = 0, float scale = 1.0f) {
    RowsDense a;
    a.X = X; a.ldx = ldx; a.X2 = X2 ? X2 : X; a.ldx2 = X2 ? ldx2 : ldx; a.KB1 = KB1; a.W = W; a.bias = bias; a.Y = Y; a.ldy = ldy;
    a.Z = Z; a.ldz = ldz; a.Zo = Zo; a.ldzo = ldzo; a.rows = rows; a.MT = MT; a.scale = scale;
    rows_dense_launch<KB, EPI>(x.st, a);
}
// first layer as rows_dense (its arguments up to `scale`), then up to two second layers on its output (k_rows_dense2); the first layer must
// have D::HT-many output tiles (MT == KB2).  OARD_ROWS_FUSE2 = 0: the same as separate launches (A/B, bit-identical)
#ifndef OARD_ROWS_FUSE2
#define OARD_ROWS_FUSE2 1
#endif
template <int KB, int EPI, int KB2>
static void rows_dense2(const TrainCtx& x, long long rows, const float* X, int ldx, const float* W, float* Y, int ldy, const float* X2, int ldx2,
                        int KB1, const float* Z, int ldz, float* Zo, int ldzo, const RowsOut2& o0, const RowsOut2* o1 = nullptr) {
    if (!OARD_ROWS_FUSE2) {
        rows_dense<KB, EPI>(x, rows, X, ldx, W, KB2, Y, ldy, nullptr, X2, ldx2, KB1, Z, ldz, Zo, ldzo);
        for (const RowsOut2* o : {&o0, o1}) {
            if (!o) continue;
            if (o->Zadd) rows_dense<KB2, EPI_ADD>(x, rows, Y, ldy, o->W, o->MT, o->Y, o->ldy, nullptr, nullptr, 0, KB2, o->Zadd, o->ldz);
            else rows_dense<KB2>(x, rows, Y, ldy, o->W, o->MT, o->Y, o->ldy);
        }
        return;
    }
    RowsDense2 q;
    RowsDense& a = q.a;
    a.X = X; a.ldx = ldx; a.X2 = X2 ? X2 : X; a.ldx2 = X2 ? ldx2 : ldx; a.KB1 = KB1; a.W = W; a.bias = nullptr; a.Y = Y; a.ldy = ldy;
    a.Z = Z; a.ldz = ldz; a.Zo = Zo; a.ldzo = ldzo; a.rows = rows; a.MT = KB2; a.scale = 1.0f;
    q.o[0] = o0; q.o[1] = o1 ? *o1 : o0; q.n2 = o1 ? 2 : 1;
    if (rows <= 0) return;
    ScopedLaunch sl_(F_NODE, x.st);
    hipLaunchKernelGGL((k_rows_dense2<KB, EPI, 8, KB2>), dim3((unsigned)cdiv(rows, 16)), dim3(512), 0, x.st, q);
}
// out[c * ostride] (+)= scale * sum_{r in [r0, r1)} w(r) act(X[r][c])
static void colsum(const TrainCtx& x, const float* X, int ld, long long r0, long long r1, int ncols, float* out, int accumulate = 1,
                   const float* wrow = nullptr, int x_silu = 0, float scale = 1.0f) {
    if (out == nullptr || ncols <= 0) return;
    ScopedLaunch sl_(F_WGRAD, x.stw);
    float* part = x.f(x.w.cpart);
    if (ncols <= 16 && !x_silu && r1 - r0 > 4 * CS_ROWS) {            // narrow and long: one thread per row slice
        const int nchn = (int)cdiv(r1 - r0, CSN_ROWS);
        hipLaunchKernelGGL(k_colsum_narrow, dim3((unsigned)nchn), dim3(256), 0, x.stw, X, ld, r0, r1, ncols, wrow, part);
        hipLaunchKernelGGL(k_colsum_fin, dim3((unsigned)cdiv(ncols, 4)), dim3(256), 0, x.stw, (const float*)part, nchn, ncols, out, accumulate, scale);
        return;
    }
    if (r1 - r0 >= 8 * CSL_ROWS && (ncols & 3) == 0 && ncols >= 8 && ncols <= 1024 && (ld & 3) == 0 && (((uintptr_t)X) & 15) == 0) {
        // (the partial buffer holds rows / 64 x 1024 floats: 32-row chunks only for ncols <= 512)
        const int rpb = r1 - r0 >= 64 * CSL_ROWS ? CSL_ROWS : (ncols <= 512 ? CSL_ROWS_SHORT : 2 * CSL_ROWS_SHORT);
        const int nchl = (int)cdiv(r1 - r0, rpb);                     // long and wide enough: float4 loads, several row lanes per block
        hipLaunchKernelGGL(k_colsum_long, dim3((unsigned)nchl), dim3(256), 0, x.stw, X, ld, r0, r1, ncols, wrow, x_silu, part, rpb);
        hipLaunchKernelGGL(k_colsum_fin, dim3((unsigned)cdiv(ncols, 4)), dim3(256), 0, x.stw, (const float*)part, nchl, ncols, out, accumulate, scale);
        return;
    }
    const int nch = (int)std::max<long long>(1, cdiv(std::max<long long>(r1 - r0, 0), CS_ROWS));
    if (r1 > r0)
        hipLaunchKernelGGL(k_colsum_part, dim3((unsigned)nch, (unsigned)cdiv(ncols, 256)), dim3(256), 0, x.stw, X, ld, r0, r1, ncols, wrow,
                           x_silu, part);
    else
        (void)hipMemsetAsync(part, 0, (size_t)ncols * sizeof(float), x.stw);
    hipLaunchKernelGGL(k_colsum_fin, dim3((unsigned)cdiv(ncols, 4)), dim3(256), 0, x.stw, (const float*)part, nch, ncols, out, accumulate, scale);
}
// weight / bias gradient of one nn.Linear, accumulated into the table entries (skipped when the table has no entry)
static int wg(const TrainCtx& x, const float* dY, int ldY, int ncY, int o_len, int o_pad, int MO, const float* X, int ldX, int ncX,
              int x_silu, int i_len, int i_pad, int MI, long long rows, float* dW, int ldW, float* db) {
    if ((dW == nullptr && db == nullptr) || rows <= 0) return OARD_OK;
    return wgrad_impl(dY, ldY, ncY, o_len, o_pad, MO, X, ldX, ncX, x_silu, i_len, i_pad, MI, rows, dW, ldW, db, 1, x.f(x.w.wg), x.w.wg_bytes,
                      x.stw);
}
#define TR_TRY(expr) do { int rc__ = (expr); if (rc__ != OARD_OK) return rc__; } while (0)

// ---- element-wise pieces ---------------------------------------------------------------------------------------------------------------
// pos_expansion hidden layer (leftnet.py:642-648 on pos_prjt = (pp0, 0, 0): only column 0 of mlp.0 takes part)
OARD_KERNEL __global__ void k_pe_hidden(const float* __restrict__ pe0, const float* __restrict__ pp0, int N, int H2, int PP, float* __restrict__ hid) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * PP) return;
    const int n = (int)(i / PP), k = (int)(i % PP);
    hid[i] = k < H2 ? silu1(pe0[k * 3] * pp0[n]) : 0.f;
}
// dzh = dhid * SiLU'(pe0[k] pp0[n])
OARD_KERNEL __global__ void k_pe_dz(const float* __restrict__ pe0, const float* __restrict__ pp0, int N, int H2, int PP, const float* __restrict__ dhid,
                        float* __restrict__ dzh) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * PP) return;
    const int n = (int)(i / PP), k = (int)(i % PP);
    float v = 0.f;
    if (k < H2) { const float z = pe0[k * 3] * pp0[n]; v = dhid[i] * dsilu1(z); }
    dzh[i] = v;
}
// EquiUpdate: sc = <vec1, x1> (the frame scalar before lin3), vdot = <vec1, vec2> / sqrt(H)       leftnet.py:329-335
OARD_KERNEL __global__ void k_upd_sc(const float* __restrict__ v12, const float* __restrict__ x1, int N, int HP, float inv_sqrt_h,
                         float* __restrict__ sc, float* __restrict__ vdot) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * HP) return;
    const int n = (int)(i / HP), ch = (int)(i % HP);
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* r = v12 + ((size_t)n * 3 + k) * 2 * HP;
        a += r[ch] * x1[n * 3 + k];
        b += r[ch] * r[HP + ch];
    }
    sc[i] = a; vdot[i] = b * inv_sqrt_h;
}
// adjoint seeds of EquiUpdate's outputs:  s_out = s_a + (a + b + vdot)/sqrt2,  vec_out[x] = vec_a[x] + c vec2[x]
//   dabc = [ds/sqrt2 | ds/sqrt2 | sum_x dvec[x] vec2[x]],  d vec2[x] = dvec[x] c + vec1[x] ds/sqrt2/sqrtH,  d vec1[x] = vec2[x] ds/sqrt2/sqrtH (+ dsc x1[x] later)
OARD_KERNEL __global__ void k_upd_seed(const float* __restrict__ ds, const float* __restrict__ dvec, const float* __restrict__ v12,
                           const float* __restrict__ cvec, int N, int HP, float inv_sqrt_h, float* __restrict__ dabc,
                           float* __restrict__ dv12) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * HP) return;
    const int n = (int)(i / HP), ch = (int)(i % HP);
    const float dsq = ds[i] * 0.70710678118654752f, c = cvec[i];
    float dc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const size_t row = (size_t)n * 3 + k;
        const float v1 = v12[row * 2 * HP + ch], v2 = v12[row * 2 * HP + HP + ch], dv = dvec[row * HP + ch];
        dc += dv * v2;
        dv12[row * 2 * HP + ch] = v2 * dsq * inv_sqrt_h;
        dv12[row * 2 * HP + HP + ch] = dv * c + v1 * dsq * inv_sqrt_h;
    }
    float* o = dabc + (size_t)n * 3 * HP;
    o[ch] = dsq; o[HP + ch] = dsq; o[2 * HP + ch] = dc;
}
OARD_KERNEL __global__ void k_upd_dv1(const float* __restrict__ dsc, const float* __restrict__ x1, int N, int HP, float* __restrict__ dv12) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * HP) return;
    const int n = (int)(i / HP), ch = (int)(i % HP);
    const float d = dsc[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) dv12[((size_t)n * 3 + k) * 2 * HP + ch] += d * x1[n * 3 + k];
}
OARD_KERNEL __global__ void k_scale_rows(const float* __restrict__ a, float scale, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * scale;
}
#define EW_GRID(n) dim3((unsigned)cdiv((long long)(n), 256)), dim3(256)

// =====================================================================================================================================
// tr_recompute: the node-side forward of layer l from the taped stage inputs (s_in, agg, s_mid): xh, the hidden activations of the
// node MLP / x_proj, xq, and cr = rbf_proj(rbf).  Restates k_node_pre_v1 / k_gcl_node_v1.
// =====================================================================================================================================
template <class D>
static int tr_recompute(const TrainCtx& x, int l) {
    const TopoDev& tp = *x.tp;
    const LayerOff& lo = x.po.layer[l];
    const int N = tp.N, HP = D::HP;
    const float* s_in = x.t(x.to.s_in[l]);
    const float* agg = x.t(x.to.agg[l]);
    const float* s_mid = x.t(x.to.s_mid[l]);
    const float* pp0 = x.t(x.to.pp0);
    { ScopedLaunch sl_(F_NODE, x.st);
      hipLaunchKernelGGL(k_pe_hidden, EW_GRID((long long)N * D::PP), 0, x.st, x.wb + x.po.pe0, pp0, N, D::H2, D::PP, x.f(x.w.hid)); }
    // s1 = s_in + pos_expansion.mlp.1 (hid)                                           leftnet.py:840-841
    rows_dense<D::PB, EPI_ADD>(x, N, x.f(x.w.hid), D::PP, x.wb + x.po.pe1, D::HT, x.f(x.w.s1), HP, nullptr, nullptr, 0, D::PB, s_in, HP);
    { ScopedLaunch sl_(F_NODE, x.st);                                                   // xh = LN_gcl(s1)   :158
      hipLaunchKernelGGL(k_rows_ln_fwd, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, x.st, (const float*)x.f(x.w.s1), (const float*)nullptr, HP,
                         D::H, HP, x.wb + lo.ln_g_w, x.wb + lo.ln_g_b, x.f(x.w.xh), (float*)nullptr, (long long)N); }
    // hm = SiLU(node_mlp.0 [xh | agg] + b)                                              :174-176
    rows_dense<2 * D::HT, EPI_SILU>(x, N, x.f(x.w.xh), HP, x.wb + lo.nm0, D::HT, x.f(x.w.hm), HP, x.wb + lo.nm0b, agg, HP, D::HT, nullptr, 0,
                                    x.f(x.w.zm), HP);
    { ScopedLaunch sl_(F_NODE, x.st);                                                   // xln = LN_msg(s_mid)   :245
      hipLaunchKernelGGL(k_rows_ln_fwd, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, x.st, s_mid, (const float*)nullptr, HP, D::H, HP,
                         x.wb + lo.ln_q_w, x.wb + lo.ln_q_b, x.f(x.w.xln), (float*)nullptr, (long long)N); }
    // hq = SiLU(x_proj.0 xln), xq = x_proj.2 hq - one launch
    rows_dense2<D::HT, EPI_SILU, D::HT>(x, N, x.f(x.w.xln), HP, x.wb + lo.xp0, x.f(x.w.hq), HP, nullptr, 0, D::HT, nullptr, 0, x.f(x.w.zq), HP,
                                        RowsOut2{x.wb + lo.xp2, x.f(x.w.xq), 3 * HP, 3 * D::HT, nullptr, 0});
    // cr = rbf_proj(rbf) on the inner edges                                             :247
    rows_dense<D::RB>(x, tp.A, x.t(x.to.rbuf), D::RP, x.wb + lo.rbfp, 3 * D::HT, x.f(x.w.cr), 3 * HP);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// =====================================================================================================================================
// tr_update_bwd: adjoint of EquiUpdate (leftnet.py:325-346 with the exact frame [x1, 0, 0]; forward: second half of k_equi_node_v1)
//   in: ds, dvec = cotangents of the layer's outputs (s_out [N][HP], vec_out [3N][HP]);  out: gs_a, gvec_a = cotangents of (s_a, vec_a)
// =====================================================================================================================================
template <class D>
static int tr_update_bwd(const TrainCtx& x, int l, const float* ds, const float* dvec, float* gs_a, float* gvec_a) {
    const TopoDev& tp = *x.tp;
    const LayerOff& lo = x.po.layer[l];
    const NodeBwdLayerOff& nl = x.nb.layer[l];
    const int N = tp.N, HP = D::HP, H = D::H, HT = D::HT;
    const long long NH = (long long)N * HP;
    const float inv_sqrt_h = 1.0f / sqrtf((float)H);
    const float* s_a = x.t(x.to.s_a[l]);
    const float* vec_a = x.t(x.to.vec_a[l]);
    const float* x1 = x.t(x.to.x1);
    const int u = x.pi.upd0 + 9 * l;
    float *v12 = x.f(x.w.v12), *sc = x.f(x.w.sc), *scal = x.f(x.w.scal), *zx = x.f(x.w.zx), *hx = x.f(x.w.hx), *cvec = x.f(x.w.cvec),
          *dabc = x.f(x.w.dabc), *dzx = x.f(x.w.dzx), *dscal = x.f(x.w.dscal), *dsc = x.f(x.w.dsc), *dv12 = x.f(x.w.dv12);
    // ---- forward pieces ----
    rows_dense<HT>(x, 3LL * N, vec_a, HP, x.wb + lo.vp, 2 * HT, v12, 2 * HP);                                    // (vec1 | vec2)   :326
    { ScopedLaunch sl_(F_NODE, x.st);
      hipLaunchKernelGGL(k_upd_sc, EW_GRID(NH), 0, x.st, (const float*)v12, x1, N, HP, inv_sqrt_h, sc, x.f(x.w.vdot));
      hipLaunchKernelGGL(k_lin3u_fwd, EW_GRID(NH), 0, x.st, x.wb + lo.l3u, (const float*)sc, NH, scal); }        // frame-scalar MLP :333
    // hx = SiLU(xvec_proj.0 [s_a | scalar]) (:337-339), c = third part of xvec_proj.2 hx - one launch
    rows_dense2<2 * HT, EPI_SILU, HT>(x, N, s_a, HP, x.wb + lo.xv0, hx, HP, scal, HP, HT, nullptr, 0, zx, HP,
                                      RowsOut2{x.wb + lo.xv2 + (size_t)2 * HT * HT * 256, cvec, HP, HT, nullptr, 0});
    // ---- adjoint ----
    { ScopedLaunch sl_(F_NODE, x.st);
      hipLaunchKernelGGL(k_upd_seed, EW_GRID(NH), 0, x.st, ds, dvec, (const float*)v12, (const float*)cvec, N, HP, inv_sqrt_h, dabc, dv12); }
    // dzx = (xvec_proj.2^T dabc) SiLU'(zx);  d [s_a | scalar] = xvec_proj.0^T dzx:  the s half lands on ds (identity path of s_out = s_a + ...),
    // the scalar half goes to lin3 - one launch
    { const RowsOut2 o_s{x.pb + nl.xv0T, gs_a, HP, HT, ds, HP}, o_c{x.pb + nl.xv0T + (size_t)HT * HT * 256, dscal, HP, HT, nullptr, 0};
      rows_dense2<3 * HT, EPI_MUL_DSILU, HT>(x, N, dabc, 3 * HP, x.pb + nl.xv2T, dzx, HP, nullptr, 0, 3 * HT, zx, HP, nullptr, 0, o_s, &o_c); }
    // lin3 adjoint + its parameter gradients in one pass (k_lin3u_bwd_fused): per-wave partial blocks, then a fixed-order reduce
    constexpr int L3_WAVES = 4, L3_BLOCKS = 512;
    { ScopedLaunch sl_(F_NODE, x.st);
      static bool l3_attr[64] = {};
      int dv_ = 0; (void)hipGetDevice(&dv_); dv_ &= 63;
      const size_t l3_lds = (size_t)L3_WAVES * L3F_WAVE_FLOATS * sizeof(float);
      if (!l3_attr[dv_]) { TR_TRY(set_lds(k_lin3u_bwd_fused<L3_WAVES>, l3_lds)); l3_attr[dv_] = true; }
      hipLaunchKernelGGL((k_lin3u_bwd_fused<L3_WAVES>), dim3(L3_BLOCKS), dim3(L3_WAVES * 64), l3_lds, x.st, x.wb + lo.l3u, (const float*)sc,
                         (const float*)dscal, NH, dsc, x.f(x.w.l3_h1));
      hipLaunchKernelGGL(k_upd_dv1, EW_GRID(NH), 0, x.st, (const float*)dsc, x1, N, HP, dv12); }
    rows_dense<2 * HT, EPI_ADD>(x, 3LL * N, dv12, 2 * HP, x.pb + nl.vpT, HT, gvec_a, HP, nullptr, nullptr, 0, 2 * HT, dvec, HP);
    // ---- parameter gradients (gradient stream) ----
    x.fork();
    { ScopedLaunch sl_(F_WGRAD, x.stw);
      hipLaunchKernelGGL(k_lin3u_reduce, dim3((unsigned)cdiv(497, 4)), dim3(256), 0, x.stw, (const float*)x.f(x.w.l3_h1), L3_BLOCKS * L3_WAVES,
                         x.g(u + 3), x.g(u + 4), x.g(u + 5), x.g(u + 6), x.g(u + 7), x.g(u + 8)); }
    TR_TRY(wg(x, dabc, 3 * HP, 3 * HP, H, HP, 3 * H, hx, HP, HP, 0, H, HP, H, N, x.g(u + 2), H, nullptr));               // xvec_proj.2 [3H][H]
    TR_TRY(wg(x, dzx, HP, HP, H, HP, H, s_a, HP, HP, 0, H, HP, H, N, x.g(u + 1), 2 * H, nullptr));                        // xvec_proj.0[:, 0:H]
    TR_TRY(wg(x, dzx, HP, HP, H, HP, H, scal, HP, HP, 0, H, HP, H, N, x.g(u + 1) ? x.g(u + 1) + H : nullptr, 2 * H, nullptr));   // [:, H:2H]
    TR_TRY(wg(x, dv12, 2 * HP, 2 * HP, H, HP, 2 * H, vec_a, HP, HP, 0, H, HP, H, 3LL * N, x.g(u + 0), H, nullptr));       // vec_proj [2H][H]
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// =====================================================================================================================================
// tr_msg_bwd: adjoint of the message formation + aggregation of EquiMessage (first half of k_equi_node_v1), + rbf_proj's gradient.
//   s_a = (s_mid + dx)/sqrt2, vec_a = vec_in + dvec.   in: gs_a, gvec_a;  out: gx = gs_a/sqrt2 (= d s_mid through this path), dxq,
//   dvec_in (cotangent of the vec entering the layer), dcd / dcr per inner edge (scratch)
// =====================================================================================================================================
template <class D>
static int tr_msg_bwd(const TrainCtx& x, int l, const float* gs_a, const float* gvec_a, float* gx, float* dxq, float* dvec_in) {
    const TopoDev& tp = *x.tp;
    const int N = tp.N, HP = D::HP, H = D::H;
    { ScopedLaunch sl_(F_NODE, x.st);
      hipLaunchKernelGGL(k_scale_rows, EW_GRID((long long)N * HP), 0, x.st, gs_a, 0.70710678118654752f, (long long)N * HP, gx); }
    // (spare row A of the per-edge gradients - padding columns of the consumer's last MFMA tile read it - is zeroed by the kernel)
    const Strided3 xq3{x.f(x.w.xq), 3 * HP, HP}, vec3{x.t(x.to.vec_in[l]), 3 * HP, HP}, cr3{x.f(x.w.cr), 3 * HP, HP}, gv3{gvec_a, 3 * HP, HP};
    LAUNCH(F_NODE, (k_equi_msg_bwd<D>), N, 256, x.st, tp, x.t(x.to.geo), xq3, vec3, x.t(x.to.cd[l]), cr3, (const float*)gx, HP, gv3,
           x.f(x.w.dcd), x.f(x.w.dcr), dxq, dvec_in, HP, x.c->reflect_equiv ? 0 : 1, 1);
    const int m = x.pi.msg0 + 9 * l;
    x.fork();
    TR_TRY(wg(x, x.f(x.w.dcr), 3 * HP, 3 * HP, H, HP, 3 * H, x.t(x.to.rbuf), D::RP, D::RP, 0, D::R, D::R, D::R, tp.A, x.g(m + 6), D::R, nullptr));
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// =====================================================================================================================================
// tr_gcl_node_bwd: adjoint of the GCL node update and x_proj (k_gcl_node_v1; leftnet.py:172-183, 245)
//   s_mid = xh + node_mlp([xh | agg]);  xq = x_proj(LN_msg(s_mid)).   in: gx (direct cotangent of s_mid), dxq;  out: dxh, dagg
// =====================================================================================================================================
template <class D>
static int tr_gcl_node_bwd(const TrainCtx& x, int l, const float* gx, const float* dxq, float* dxh, float* dagg) {
    const TopoDev& tp = *x.tp;
    const LayerOff& lo = x.po.layer[l];
    const NodeBwdLayerOff& nl = x.nb.layer[l];
    const int N = tp.N, HP = D::HP, H = D::H, HT = D::HT;
    const int g = x.pi.gcl0 + 14 * l, m = x.pi.msg0 + 9 * l;
    const float* agg = x.t(x.to.agg[l]);
    const float* s_mid = x.t(x.to.s_mid[l]);
    float *dzq = x.f(x.w.dzq), *dxln = x.f(x.w.dxln), *lng = x.f(x.w.lng), *dsm = x.f(x.w.dsm), *dzm = x.f(x.w.dzm);
    rows_dense2<3 * HT, EPI_MUL_DSILU, HT>(x, N, dxq, 3 * HP, x.pb + nl.xp2T, dzq, HP, nullptr, 0, 3 * HT, x.f(x.w.zq), HP, nullptr, 0,
                                           RowsOut2{x.pb + nl.xp0T, dxln, HP, HT, nullptr, 0});      // dzq, then dxln = x_proj.0^T dzq
    { ScopedLaunch sl_(F_NODE, x.st);                         // d s_mid = gx + LN_msg^T dxln
      hipLaunchKernelGGL(k_rows_ln_bwd, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, x.st, s_mid, HP, H, HP, x.wb + lo.ln_q_w, (const float*)dxln,
                         gx, dsm, lng, (long long)N); }
    { const RowsOut2 o_x{x.pb + nl.nm0T, dxh, HP, HT, dsm, HP},                                                 // residual path + xh half
                     o_a{x.pb + nl.nm0T + (size_t)HT * HT * 256, dagg, HP, HT, nullptr, 0};                      // agg half
      rows_dense2<HT, EPI_MUL_DSILU, HT>(x, N, dsm, HP, x.pb + nl.nm1T, dzm, HP, nullptr, 0, HT, x.f(x.w.zm), HP, nullptr, 0, o_x, &o_a); }
    x.fork();
    colsum(x, lng, HP, 0, N, H, x.g(m + 7));                  // message_layers.l.x_layernorm.weight
    colsum(x, dxln, HP, 0, N, H, x.g(m + 8));                 // .bias
    TR_TRY(wg(x, dxq, 3 * HP, 3 * HP, H, HP, 3 * H, x.f(x.w.hq), HP, HP, 0, H, HP, H, N, x.g(m + 5), H, nullptr));             // x_proj.2 [3H][H]
    TR_TRY(wg(x, dzq, HP, HP, H, HP, H, x.f(x.w.xln), HP, HP, 0, H, HP, H, N, x.g(m + 4), H, nullptr));                        // x_proj.0
    TR_TRY(wg(x, dsm, HP, HP, H, HP, H, x.f(x.w.hm), HP, HP, 0, H, HP, H, N, x.g(g + 6), H, x.g(g + 7)));                      // node_mlp.1
    TR_TRY(wg(x, dzm, HP, HP, H, HP, H, x.f(x.w.xh), HP, HP, 0, H, HP, H, N, x.g(g + 4), 2 * H, x.g(g + 5)));                  // node_mlp.0[:, 0:H], bias
    TR_TRY(wg(x, dzm, HP, HP, H, HP, H, agg, HP, HP, 0, H, HP, H, N, x.g(g + 4) ? x.g(g + 4) + H : nullptr, 2 * H, nullptr));  // [:, H:2H]
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// =====================================================================================================================================
// tr_equi_edge_bwd: EquiMessage edge part (k_equi_edge_bwd): dcd -> dew[0:A) += dir_proj^T, dir_proj gradients
// tr_gcl_edge_bwd: GCLMessage edge part (k_gcl_edge_bwd): dew (new state) + dagg -> dew (old state), dP / dQ, edge-MLP gradients
// =====================================================================================================================================
template <class D>
static int tr_equi_edge_bwd(const TrainCtx& x, int l, float* dew) {
    const TopoDev& tp = *x.tp;
    if (tp.A <= 0) return OARD_OK;
    const int H = D::H, HP = D::HP, W = D::W;
    const int m = x.pi.msg0 + 9 * l;
    TR_TRY(equi_backward_impl<D>(tp, x.pb + x.bo.layer[l].equi, x.f(x.w.dcd), x.t(x.to.zd1[l]), dew, x.f(x.w.dzd1), x.st));
    x.fork();
    TR_TRY(wg(x, x.f(x.w.dcd), 3 * HP, 3 * HP, H, HP, 3 * H, x.t(x.to.zd1[l]), D::D1P, D::D1P, 1, 3 * H, 3 * H, 3 * H, tp.A, x.g(m + 2), 3 * H,
              x.g(m + 3)));                                                                                       // dir_proj.2
    TR_TRY(wg(x, x.f(x.w.dzd1), D::D1P, D::D1P, 3 * H, 3 * H, 3 * H, x.t(x.to.ew[l + 1]), D::WP, D::WP, 0, W, W, W, tp.A, x.g(m + 0), W,
              x.g(m + 1)));                                                                                       // dir_proj.0
    return OARD_OK;
}
template <class D>
static int tr_gcl_edge_bwd(const TrainCtx& x, int l, const float* dagg, float* dew, float* dP, float* dQ) {
    const TopoDev& tp = *x.tp;
    const int N = tp.N, H = D::H, HP = D::HP, W = D::W, WP = D::WP, NL = x.c->num_layers;
    const long long E = tp.E, A = tp.A;
    if (E <= 0) {
        HIP_TRY(hipMemsetAsync(dP, 0, (size_t)N * HP * sizeof(float), x.st));
        HIP_TRY(hipMemsetAsync(dQ, 0, (size_t)N * HP * sizeof(float), x.st));
        return OARD_OK;
    }
    const int g = x.pi.gcl0 + 14 * l;
    float *dz3 = x.f(x.w.dz3), *mout = x.f(x.w.mout), *dz2 = x.f(x.w.dz2), *dz1 = x.f(x.w.dz1), *da = x.f(x.w.da);
    long long gate_rows = 0;
    float* gatep = (g_gate_fold && HP > H) ? x.f(x.w.gatep) : nullptr;
    TR_TRY(gcl_backward_impl<D>(x.c, tp, x.pb, x.bo.layer[l], l, x.tape, x.to, dagg, dew, dz3, mout, dz2, da, dz1, x.st, gatep, &gate_rows));
    LAUNCH(F_GCL_BWD, k_edge_node_sums, N, 256, x.st, tp, (const float*)dz1, HP, dP, dQ);
    x.fork();
    const long long rows3 = l == NL - 1 ? A : E;          // rows whose forward evaluated edge_out_trans
    TR_TRY(wg(x, dz3, WP, WP, W, W, W, mout, HP, HP, 0, H, H, H, rows3, x.g(g + 8), H, x.g(g + 9)));                               // edge_out_trans
    TR_TRY(wg(x, dz2, HP, HP, H, H, H, x.t(x.to.z1[l]), HP, HP, 1, H, H, H, E, x.g(g + 2), H, x.g(g + 3)));                         // edge_mlp.1
    const long long rows1 = l == 0 ? A : E;               // layer 0 sees the never-materialised constant row on inter-object edges
    float* w1c = x.g(g + 0) ? x.g(g + 0) + 2 * H : nullptr;
    TR_TRY(wg(x, dz1, HP, HP, H, H, H, x.t(x.to.ew[l]), WP, WP, 0, W, W, W, rows1, w1c, 2 * H + W, nullptr));                       // edge_mlp.0[:, 2H:]
    if (l == 0 && E > A && w1c != nullptr) {              // ... whose contribution is outer(sum_e dz1_e, c0row)
        colsum(x, dz1, HP, A, E, H, x.f(x.w.csum), 0);
        ScopedLaunch sl_(F_WGRAD, x.stw);
        hipLaunchKernelGGL(k_outer_acc, EW_GRID((long long)H * W), 0, x.stw, w1c, 2 * H + W, (const float*)x.f(x.w.csum), H, x.wb + x.po.c0row, W, 1);
    }
    // att_mlp weight: sum_e da_e SiLU(z2_e), bias: sum_e da_e - k_gcl_edge_bwd left per-wave sums over its 16 edges (feature H = the bias)
    if (gatep) {
        colsum(x, gatep, HP, 0, gate_rows, H, x.g(g + 10));
        colsum(x, gatep + H, HP, 0, gate_rows, 1, x.g(g + 11));
    } else {
        colsum(x, x.t(x.to.z2[l]), HP, 0, E, H, x.g(g + 10), 1, da, 1);
        colsum(x, da, 1, 0, E, 1, x.g(g + 11));
    }
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// =====================================================================================================================================
// tr_pre_bwd: adjoint of k_node_pre_v1: s1 = s_in + pos_expansion(pos_prjt); xh = LN_gcl(s1); P = W1a xh + b1; Q = W1b xh
//   in: dxh (from the node update), dP, dQ (node sums of the edge kernel's dz1);  out: ds_in
// =====================================================================================================================================
template <class D>
static int tr_pre_bwd(const TrainCtx& x, int l, const float* dxh, const float* dP, const float* dQ, float* ds_in) {
    const TopoDev& tp = *x.tp;
    const LayerOff& lo = x.po.layer[l];
    const NodeBwdLayerOff& nl = x.nb.layer[l];
    const int N = tp.N, HP = D::HP, H = D::H, HT = D::HT, W = D::W;
    const int g = x.pi.gcl0 + 14 * l;
    const float* xh = x.f(x.w.xh);
    float *t1 = x.f(x.w.t1), *t2 = x.f(x.w.t2), *lng = x.f(x.w.lng2), *dsn = x.f(x.w.dsn), *dhid = x.f(x.w.dhid), *dzh = x.f(x.w.dzh);
    rows_dense<HT, EPI_ADD>(x, N, dP, HP, x.pb + nl.W1aT, HT, t1, HP, nullptr, nullptr, 0, HT, dxh, HP);
    rows_dense<HT, EPI_ADD>(x, N, dQ, HP, x.pb + nl.W1bT, HT, t2, HP, nullptr, nullptr, 0, HT, t1, HP);
    { ScopedLaunch sl_(F_NODE, x.st);                         // the layer's buffer set keeps ds (the gradient stream reads it while the chain moves on)
      hipLaunchKernelGGL(k_rows_ln_bwd, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, x.st, (const float*)x.f(x.w.s1), HP, H, HP, x.wb + lo.ln_g_w,
                         (const float*)t2, (const float*)nullptr, dsn, lng, (long long)N); }
    HIP_TRY(hipMemcpyAsync(ds_in, dsn, (size_t)N * HP * sizeof(float), hipMemcpyDeviceToDevice, x.st));
    // ---- parameter gradients (gradient stream) ----
    x.fork();
    colsum(x, lng, HP, 0, N, H, x.g(g + 12));                 // gcl_layers.l.x_layernorm.weight
    colsum(x, t2, HP, 0, N, H, x.g(g + 13));                  // .bias
    TR_TRY(wg(x, dP, HP, HP, H, HP, H, xh, HP, HP, 0, H, HP, H, N, x.g(g + 0), 2 * H + W, x.g(g + 1)));                      // edge_mlp.0[:, 0:H], bias
    TR_TRY(wg(x, dQ, HP, HP, H, HP, H, xh, HP, HP, 0, H, HP, H, N, x.g(g + 0) ? x.g(g + 0) + H : nullptr, 2 * H + W, nullptr));   // [:, H:2H]
    // pos_expansion (shared by all layers: its gradient accumulates over l); nothing downstream needs its input cotangent
    if (x.g(x.pi.pe0_w) != nullptr || x.g(x.pi.pe1_w) != nullptr) {
        const TrainCtx y = x.on_grad_stream();
        rows_dense<HT>(y, N, dsn, HP, x.pb + x.nb.pe1T, D::PB, dhid, D::PP);
        { ScopedLaunch sl_(F_NODE, x.stw);
          hipLaunchKernelGGL(k_pe_dz, EW_GRID((long long)N * D::PP), 0, x.stw, x.wb + x.po.pe0, x.t(x.to.pp0), N, D::H2, D::PP, (const float*)dhid, dzh); }
        TR_TRY(wg(x, dsn, HP, HP, H, HP, H, x.f(x.w.hid), D::PP, D::PP, 0, D::H2, D::PP, D::H2, N, x.g(x.pi.pe1_w), D::H2, nullptr));
        if (x.g(x.pi.pe0_w) != nullptr) {                     // mlp.0.weight [H/2][3]: only column 0 sees a non-zero input
            colsum(x, dzh, D::PP, 0, N, D::H2, x.f(x.w.csum), 0, x.t(x.to.pp0));
            ScopedLaunch sl_(F_WGRAD, x.stw);
            hipLaunchKernelGGL(k_strided_acc, EW_GRID(D::H2), 0, x.stw, (const float*)x.f(x.w.csum), D::H2, x.g(x.pi.pe0_w), 3);
        }
    }
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// =====================================================================================================================================
// one layer of the reverse sweep.  ds [N][HP], dvec [3N][HP]: in = cotangents of the layer's outputs, out = of its inputs; dew [E+1][WP]:
// in = cotangent of the edge state leaving the layer, out = entering it.
// =====================================================================================================================================
template <class D>
static int tr_layer_bwd(const TrainCtx& x0, int l, float* ds, float* dvec, float* dew) {
    TrainCtx x = x0;
    x.par = l & 1;                                            // buffer set of this layer
    x.wait_layer(l + 2);                                      // ... which layer l + 2's gradients may still be reading
    float *gs_a = x.f(x.w.gs_a), *gvec_a = x.f(x.w.gvec_a), *gx = x.f(x.w.gx), *dxq = x.f(x.w.dxq), *dxh = x.f(x.w.dxh), *dagg = x.f(x.w.dagg);
    float* dP = x.f(x.w.dPQ);
    float* dQ = dP + (size_t)x.tp->N * D::HP;
    TR_TRY(tr_recompute<D>(x, l));
    TR_TRY(tr_update_bwd<D>(x, l, ds, dvec, gs_a, gvec_a));
    TR_TRY(tr_msg_bwd<D>(x, l, gs_a, gvec_a, gx, dxq, dvec));            // dvec <- cotangent of the vec entering the layer
    TR_TRY(tr_gcl_node_bwd<D>(x, l, gx, dxq, dxh, dagg));
    TR_TRY(tr_equi_edge_bwd<D>(x, l, dew));
    TR_TRY(tr_gcl_edge_bwd<D>(x, l, dagg, dew, dP, dQ));
    TR_TRY(tr_pre_bwd<D>(x, l, dxh, dP, dQ, ds));
    TR_TRY(wgq_flush());                                      // the layer's queued node-level weight gradients: one grouped launch
    x.layer_done(l);
    return OARD_OK;
}

// =====================================================================================================================================
// tr_tail_bwd: adjoint of the output block (k_out_v1; leftnet.py:566-576, 878-891), of the velocity / per-object CoM removal and
// of the decoders (k_post; egnn_dynamics.py:137-160).
//   in: go[k] = cotangent of out[k] ([n_k][nf_k], reference row order), nullptr = zero;  out: ds [N][HP], dvec [3N][HP] (cotangents
//   of the final node state)
// =====================================================================================================================================
struct TailPtrs {
    const float* go[OARD_MAX_OBJECTS];
    int node_nf[OARD_MAX_OBJECTS];
    size_t dec[OARD_MAX_OBJECTS], enc[OARD_MAX_OBJECTS];      // offsets of the raw MLP blocks in the packed blob
};
// v1 = |vec1_proj(vec)|_xyz per channel, v2s[n][x] = <vec[n][x], vec2_proj.weight>          one 64-thread block per node
OARD_KERNEL __global__ __launch_bounds__(64) void k_out_norms(const float* __restrict__ u, const float* __restrict__ vec, const float* __restrict__ v2p,
                                                  int HP, float* __restrict__ v1, float* __restrict__ v2s) {
    const int n = blockIdx.x;
    float p[3] = {0.f, 0.f, 0.f};
    for (int ch = threadIdx.x; ch < HP; ch += 64) {
        float q = 0.f;
        const float w = v2p[ch];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t i = ((size_t)n * 3 + k) * HP + ch;
            q += u[i] * u[i];
            p[k] += vec[i] * w;
        }
        v1[(size_t)n * HP + ch] = sqrtf(q);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) p[k] += __shfl_xor(p[k], d, 64);
        if (threadIdx.x == 0) v2s[n * 3 + k] = p[k];
    }
}
// per node: cotangent of dpos through the CoM removal, decoder adjoint (operands of its weight gradients in REFERENCE row order:
// objects are contiguous there), gate / vec2 adjoints.   dxg [N][16] = (0, dgate, 0 ...), dhout [N][16], dv2s [N][3]
OARD_KERNEL __global__ void k_post_bwd(TopoDev tp, TailPtrs tl, const float* __restrict__ wb, const float* __restrict__ hout,
                           const float* __restrict__ xg, const float* __restrict__ v2s, int emb, float* __restrict__ dxg,
                           float* __restrict__ dhout, float* __restrict__ dv2s, float* __restrict__ dgate, float* __restrict__ dec_dy,
                           float* __restrict__ dec_hid, float* __restrict__ dec_dz, float* __restrict__ dec_x) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= tp.N) return;
    const int obj = tp.node_obj[n], row = tp.node_row[n], nf = tl.node_nf[obj], d = nf - 3;
    const int q = tp.node_sample[n] * tp.n_obj + obj;
    const int g0 = tp.grp_ptr[q], g1 = tp.grp_ptr[q + 1];
    const float* G = tl.go[obj];
    float dd[3] = {0.f, 0.f, 0.f};
    if (G != nullptr) {                  // vel = dpos - mean_group(dpos): the projection is symmetric
        float m[3] = {0.f, 0.f, 0.f};
        for (int k = g0; k < g1; ++k) { const float* r = G + (size_t)tp.node_row[k] * nf; m[0] += r[0]; m[1] += r[1]; m[2] += r[2]; }
        const float inv = 1.0f / (float)(g1 - g0);
        const float* r = G + (size_t)row * nf;
#pragma unroll
        for (int k = 0; k < 3; ++k) dd[k] = r[k] - m[k] * inv;
    }
    const float gate = xg[(size_t)n * 16 + 1];
    float dg = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { dg += dd[k] * v2s[n * 3 + k]; dv2s[n * 3 + k] = gate * dd[k]; }
    dgate[n] = dg;
    for (int c = 0; c < 16; ++c) dxg[(size_t)n * 16 + c] = c == 1 ? dg : 0.f;
    // decoder: h[:emb] -> Linear(emb, 2d) SiLU -> Linear(2d, d)
    const float* W0 = wb + tl.dec[obj];
    const float* b0 = W0 + 2 * d * emb;
    const float* W1 = b0 + 2 * d;
    const float* h = hout + (size_t)n * 16;
    const size_t r = (size_t)tp.node_ref[n];
    float z[32], dh[16];
    for (int k = 0; k < 2 * d; ++k) {
        float a = 0.f;
        for (int i = 0; i < emb; ++i) a += W0[k * emb + i] * h[i];
        z[k] = a + b0[k];
    }
    for (int i = 0; i < 16; ++i) dh[i] = 0.f;
    for (int k = 0; k < 32; ++k) {
        float dhid = 0.f;
        if (k < 2 * d && G != nullptr)
            for (int o = 0; o < d; ++o) dhid += W1[o * 2 * d + k] * G[(size_t)row * nf + 3 + o];
        const float dz = k < 2 * d ? dhid * dsilu1(z[k]) : 0.f;
        dec_hid[r * 32 + k] = k < 2 * d ? silu_acc(z[k]) : 0.f;
        dec_dz[r * 32 + k] = dz;
        if (k < 2 * d)
            for (int i = 0; i < emb; ++i) dh[i] += W0[k * emb + i] * dz;
    }
    for (int i = 0; i < 16; ++i) {
        dec_dy[r * 16 + i] = (i < d && G != nullptr) ? G[(size_t)row * nf + 3 + i] : 0.f;
        dec_x[r * 16 + i] = i < emb ? h[i] : 0.f;
        dhout[(size_t)n * 16 + i] = dh[i];
    }
}
// du[x] = dv1 u[x] / v1 (zero subgradient at v1 = 0, as torch.norm);  zadd[x] = dv2s[x] vec2_proj.weight
OARD_KERNEL __global__ void k_out_du(const float* __restrict__ dv1, const float* __restrict__ v1, const float* __restrict__ u,
                         const float* __restrict__ dv2s, const float* __restrict__ v2p, int N, int HP, float* __restrict__ du,
                         float* __restrict__ zadd) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * HP) return;
    const int n = (int)(i / HP), ch = (int)(i % HP);
    const float nv = v1[i], f = nv > 0.f ? dv1[i] / nv : 0.f, w = v2p[ch];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const size_t j = ((size_t)n * 3 + k) * HP + ch;
        du[j] = f * u[j];
        zadd[j] = dv2s[n * 3 + k] * w;
    }
}

struct TrainTail {          // extra scratch of the tail / init stages (after TrainWs.total)
    size_t u, v1, v2s, zu, hu, xg, hout, dxg, dhout, dv2s, dgate, dzu, dv1, du, zadd, op_dy, op_hid, op_dz, op_x, eo_dy, eo_hid, eo_dz, eo_x;
    size_t ynb, nbe, ys, lns, s1v, ne1, dne1, part, zrl, hrl, df, ds1v, dlns, dys, ds0, segS, segG, segSn, segGn, dnbe, ipro, dynb, dhin, dfr,
        dzrl, dc0;
    size_t total;
};
static TrainTail make_train_tail(const oard_config* c, const TopoDev& td, size_t base) {
    const RDims d(c->hidden, c->num_radial);
    const size_t N = td.N, A = td.A + 1, HP = d.HP, F = sizeof(float), B = td.B, G = td.n_groups;
    TrainTail w;
    memset(&w, 0, sizeof(w));
    size_t cur = base;
    auto take = [&](size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; };
    w.u = take(3 * N * HP * F); w.v1 = take(N * HP * F); w.v2s = take(N * 4 * F); w.zu = take(N * HP * F); w.hu = take(N * HP * F);
    w.xg = take(N * 16 * F); w.hout = take(N * 16 * F); w.dxg = take(N * 16 * F); w.dhout = take(N * 16 * F); w.dv2s = take(N * 4 * F);
    w.dgate = take(N * F); w.dzu = take(N * HP * F); w.dv1 = take(N * HP * F); w.du = take(3 * N * HP * F); w.zadd = take(3 * N * HP * F);
    w.op_dy = take(N * 16 * F); w.op_hid = take(N * 32 * F); w.op_dz = take(N * 32 * F); w.op_x = take(N * 16 * F);
    w.eo_dy = take(N * 16 * F); w.eo_hid = take(N * 32 * F); w.eo_dz = take(N * 32 * F); w.eo_x = take(N * 16 * F);     // encoders (the decoders' gradients may still be read)
    w.ynb = take(N * HP * F); w.nbe = take(N * HP * F); w.ys = take(N * HP * F); w.lns = take(N * HP * F); w.s1v = take(N * HP * F);
    w.ne1 = take(3 * N * HP * F); w.dne1 = take(3 * N * HP * F); w.part = take(N * (5 * d.H4 + 1) * F);
    w.zrl = take(A * HP * F); w.hrl = take(A * HP * F); w.df = take(A * HP * F); w.ds1v = take(N * HP * F); w.dlns = take(N * HP * F);
    w.dys = take(N * HP * F); w.ds0 = take(N * HP * F); w.segS = take((B + 1) * HP * F); w.segG = take((G + 1) * HP * F);
    w.segSn = take((B + 1) * HP * F); w.segGn = take((G + 1) * HP * F); w.dnbe = take(N * HP * F); w.ipro = take(N * HP * F);
    w.dynb = take(N * HP * F); w.dhin = take(N * 16 * F); w.dfr = take(A * HP * F); w.dzrl = take(A * HP * F); w.dc0 = take(d.WP * F);
    w.total = cur;
    return w;
}

template <class D>
static int tr_tail_bwd(const TrainCtx& x, const TrainTail& tw, const oard_topology* topo, const float* const* go, float* ds, float* dvec) {
    const TopoDev& tp = *x.tp;
    const int N = tp.N, HP = D::HP, H = D::H, HT = D::HT, NL = x.c->num_layers, C = x.c->in_hidden, emb = embed_dim(x.c);
    const float* s_L = x.t(x.to.s_in[NL]);
    const float* vec_L = x.t(x.to.vec_in[NL]);
    const int o = x.pi.out0;
    float *u = x.f(tw.u), *v1 = x.f(tw.v1), *v2s = x.f(tw.v2s), *zu = x.f(tw.zu), *hu = x.f(tw.hu), *xg = x.f(tw.xg), *hout = x.f(tw.hout),
          *dxg = x.f(tw.dxg), *dhout = x.f(tw.dhout), *dv2s = x.f(tw.dv2s), *dgate = x.f(tw.dgate), *dzu = x.f(tw.dzu), *dv1 = x.f(tw.dv1),
          *du = x.f(tw.du), *zadd = x.f(tw.zadd);
    // ---- forward pieces ----
    rows_dense<HT>(x, 3LL * N, vec_L, HP, x.wb + x.po.v1p, HT, u, HP);
    LAUNCH(F_NODE, k_out_norms, N, 64, x.st, (const float*)u, vec_L, x.wb + x.po.v2p, HP, v1, v2s);
    rows_dense<2 * HT, EPI_SILU>(x, N, s_L, HP, x.wb + x.po.un0, HT, hu, HP, x.wb + x.po.un0_b, v1, HP, HT, nullptr, 0, zu, HP);
    rows_dense<HT>(x, N, hu, HP, x.wb + x.po.un2, 1, xg, 16, x.wb + x.po.un2_b);
    rows_dense<HT>(x, N, s_L, HP, x.wb + x.po.embout, 1, hout, 16, x.wb + x.po.embout_b);
    // ---- adjoint ----
    TailPtrs tl;
    memset(&tl, 0, sizeof(tl));
    for (int k = 0; k < x.c->n_obj; ++k) { tl.go[k] = go ? go[k] : nullptr; tl.node_nf[k] = x.c->node_nf[k]; tl.dec[k] = x.po.dec[k]; tl.enc[k] = x.po.enc[k]; }
    LAUNCH(F_NODE, k_post_bwd, cdiv(N, 128), 128, x.st, tp, tl, x.wb, (const float*)hout, (const float*)xg, (const float*)v2s, emb, dxg, dhout,
           dv2s, dgate, x.f(tw.op_dy), x.f(tw.op_hid), x.f(tw.op_dz), x.f(tw.op_x));
    rows_dense<1, EPI_MUL_DSILU>(x, N, dxg, 16, x.pb + x.nb.un2T, HT, dzu, HP, nullptr, nullptr, 0, 1, zu, HP);
    rows_dense<HT>(x, N, dzu, HP, x.pb + x.nb.un0T, HT, dv1 /* tmp: s half */, HP);
    rows_dense<1, EPI_ADD>(x, N, dhout, 16, x.pb + x.nb.emboutT, HT, ds, HP, nullptr, nullptr, 0, 1, dv1, HP);     // ds = un0^T(s half) + embedding_out^T dhout
    rows_dense<HT>(x, N, dzu, HP, x.pb + x.nb.un0T + (size_t)HT * HT * 256, HT, dv1, HP);                           // |vec1| half
    { ScopedLaunch sl_(F_NODE, x.st);
      hipLaunchKernelGGL(k_out_du, EW_GRID((long long)N * HP), 0, x.st, (const float*)dv1, (const float*)v1, (const float*)u, (const float*)dv2s,
                         x.wb + x.po.v2p, N, HP, du, zadd); }
    rows_dense<HT, EPI_ADD>(x, 3LL * N, du, HP, x.pb + x.nb.v1pT, HT, dvec, HP, nullptr, nullptr, 0, HT, zadd, HP);
    // ---- parameter gradients (gradient stream) ----
    x.fork();
    TR_TRY(wg(x, du, HP, HP, H, HP, H, vec_L, HP, HP, 0, H, HP, H, 3LL * N, x.g(o + 0), H, nullptr));                      // vec1_proj
    colsum(x, vec_L, HP, 0, 3LL * N, H, x.g(o + 1), 1, dv2s);                                                               // vec2_proj [1][H]
    TR_TRY(wg(x, dzu, HP, HP, H, HP, H, s_L, HP, HP, 0, H, HP, H, N, x.g(o + 2), 2 * H, x.g(o + 3)));                       // update_net.0[:, 0:H], bias
    TR_TRY(wg(x, dzu, HP, HP, H, HP, H, v1, HP, HP, 0, H, HP, H, N, x.g(o + 2) ? x.g(o + 2) + H : nullptr, 2 * H, nullptr));
    colsum(x, hu, HP, 0, N, H, x.g(o + 4) ? x.g(o + 4) + H : nullptr, 1, dgate);                                            // update_net.2 row 1 (the gate)
    colsum(x, dgate, 1, 0, N, 1, x.g(o + 5) ? x.g(o + 5) + 1 : nullptr);
    TR_TRY(wg(x, dhout, 16, 16, C, 16, C, s_L, HP, HP, 0, H, HP, H, N, x.g(x.pi.embout_w), H, x.g(x.pi.embout_b)));        // embedding_out [C][H]
    for (int k = 0; k < x.c->n_obj; ++k) {                 // decoders: Linear(emb, 2d) SiLU Linear(2d, d), rows of object k are contiguous in reference order
        const int a = x.c->enc_alias[k], d = x.c->node_nf[k] - 3, q = x.pi.dec0 + 4 * a;
        const long long r0 = topo->obj_start[k], nk = topo->obj_start[k + 1] - r0;
        if (nk <= 0 || go == nullptr || go[k] == nullptr) continue;
        TR_TRY(wg(x, x.f(tw.op_dz) + r0 * 32, 32, 32, 2 * d, 2 * d, 2 * d, x.f(tw.op_x) + r0 * 16, 16, 16, 0, emb, emb, emb, nk, x.g(q + 0), emb, x.g(q + 1)));
        TR_TRY(wg(x, x.f(tw.op_dy) + r0 * 16, 16, 16, d, d, d, x.f(tw.op_hid) + r0 * 32, 32, 32, 0, 2 * d, 2 * d, 2 * d, nk, x.g(q + 2), 2 * d, x.g(q + 3)));
    }
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// =====================================================================================================================================
// tr_init_bwd: adjoint of the init head and the encoders.
//   s0 = z_emb + sum_{a: m->n} f_a nbe[m] (+ the constant f of inter-object edges);  NE1 = S2V;  ew0 = [scalarise(NE1) | f | rbf]
//   in: ds0 [N][HP] (cotangent of the node state entering layer 0), dew [E+1][WP] (cotangent of the initial edge state);  xh: the
//   call's inputs (encoder gradients).
// =====================================================================================================================================
// sums of X over the nodes of every sample / (sample, object) group, fixed order            one block per sample
OARD_KERNEL __global__ __launch_bounds__(256) void k_seg_sums(TopoDev tp, const float* __restrict__ X, int HP, float* __restrict__ outS,
                                                  float* __restrict__ outG) {
    const int b = blockIdx.x, ch = threadIdx.x;
    if (ch >= HP) return;
    float s = 0.f;
    for (int o = 0; o < tp.n_obj; ++o) {
        const int q = b * tp.n_obj + o;
        float g = 0.f;
        for (int n = tp.grp_ptr[q]; n < tp.grp_ptr[q + 1]; ++n) g += X[(size_t)n * HP + ch];
        outG[(size_t)q * HP + ch] = g;
        s += g;
    }
    outS[(size_t)b * HP + ch] = s;
}
// S2V aggregation adjoint (k_s2v_agg_v1; leftnet.py:117-125): NE1[n][x] = sum_{a: m->n} f_a s1v[m] u_a[x].  One block per node, one
// thread per channel, both roles of the node (k_equi_msg_bwd's scheme): df[a] = dew[a][2H + ch] + <dNE1[n], u_a> s1v[m] for the
// incoming edges, ds1v[n] = sum over the outgoing edges (n -> m) of <dNE1[m], u> f.
template <class D>
__global__ __launch_bounds__(256) void k_s2v_bwd(TopoDev tp, const float* __restrict__ geo, const float* __restrict__ ew0,
                                                 const float* __restrict__ s1v, const float* __restrict__ dne1, const float* __restrict__ dew,
                                                 float* __restrict__ df, float* __restrict__ ds1v) {
    const int n = blockIdx.x, ch = threadIdx.x;
    if (ch >= D::HP) return;
    const bool real = ch < D::H;
    const int q_grp = tp.node_sample[n] * tp.n_obj + tp.node_obj[n];
    const int g0 = tp.grp_ptr[q_grp], ng = tp.grp_ptr[q_grp + 1] - g0, self = n - g0;
    const int a_n = tp.act_ptr[n];
    float dn[3] = {0.f, 0.f, 0.f};
    if (real)
#pragma unroll
        for (int k = 0; k < 3; ++k) dn[k] = dne1[((size_t)n * 3 + k) * D::HP + ch];
    float acc = 0.f;
    for (int kk = 0; kk < ng; ++kk) {
        if (kk == self) continue;
        const int m = g0 + kk;
        const size_t a = (size_t)a_n + kk - (kk > self ? 1 : 0);                    // (m -> n)
        if (!real) { df[a * D::HP + ch] = 0.f; continue; }
        const float* ge = geo + a * GEO_STRIDE;
        df[a * D::HP + ch] = dew[a * (size_t)D::WP + 2 * D::H + ch] + (dn[0] * ge[2] + dn[1] * ge[3] + dn[2] * ge[4]) * s1v[(size_t)m * D::HP + ch];
        const size_t b = (size_t)tp.act_ptr[m] + self - (self > kk ? 1 : 0);        // (n -> m)
        const float* gb = geo + b * GEO_STRIDE;
        const float t = dne1[((size_t)m * 3 + 0) * D::HP + ch] * gb[2] + dne1[((size_t)m * 3 + 1) * D::HP + ch] * gb[3] +
                        dne1[((size_t)m * 3 + 2) * D::HP + ch] * gb[4];
        acc += t * ew0[b * (size_t)D::WP + 2 * D::H + ch];
    }
    ds1v[(size_t)n * D::HP + ch] = real ? acc : 0.f;
}
// NeighborEmb adjoint (k_neighbor_v1; leftnet.py:81-89):  s0[n] = z_emb[n] + sum_{a: m->n} f_a nbe[m] + c0f (sum_sample nbe - sum_group nbe)
//   df[a] += ds0[n] nbe[m];   dnbe[n] = sum_{(n -> m)} f ds0[m] + c0f (sum_sample ds0 - sum_group ds0);   ipro[n] = ds0[n] inter[n]
template <class D>
__global__ __launch_bounds__(256) void k_nbr_bwd(TopoDev tp, const float* __restrict__ ew0, const float* __restrict__ c0f,
                                                 const float* __restrict__ nbe, const float* __restrict__ ds0, const float* __restrict__ segS,
                                                 const float* __restrict__ segG, const float* __restrict__ segSn, const float* __restrict__ segGn,
                                                 float* __restrict__ df, float* __restrict__ dnbe, float* __restrict__ ipro) {
    const int n = blockIdx.x, ch = threadIdx.x;
    if (ch >= D::HP) return;
    const size_t i = (size_t)n * D::HP + ch;
    if (ch >= D::H) { dnbe[i] = 0.f; ipro[i] = 0.f; return; }
    const int smp = tp.node_sample[n], q_grp = smp * tp.n_obj + tp.node_obj[n];
    const int g0 = tp.grp_ptr[q_grp], ng = tp.grp_ptr[q_grp + 1] - g0, self = n - g0;
    const int a_n = tp.act_ptr[n];
    const float dsn = ds0[i];
    float acc = 0.f;
    for (int kk = 0; kk < ng; ++kk) {
        if (kk == self) continue;
        const int m = g0 + kk;
        const size_t a = (size_t)a_n + kk - (kk > self ? 1 : 0);                    // (m -> n)
        df[a * D::HP + ch] += dsn * nbe[(size_t)m * D::HP + ch];
        const size_t b = (size_t)tp.act_ptr[m] + self - (self > kk ? 1 : 0);        // (n -> m)
        acc += ew0[b * (size_t)D::WP + 2 * D::H + ch] * ds0[(size_t)m * D::HP + ch];
    }
    const float c = c0f[ch];
    dnbe[i] = acc + c * (segS[(size_t)smp * D::HP + ch] - segG[(size_t)q_grp * D::HP + ch]);
    ipro[i] = dsn * (segSn[(size_t)smp * D::HP + ch] - segGn[(size_t)q_grp * D::HP + ch]);
}
// dfr[a] = df[a] env[a]
OARD_KERNEL __global__ void k_scale_by_env(const float* __restrict__ df, const float* __restrict__ geo, long long A, int HP, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A * HP) return;
    out[i] = df[i] * geo[(i / HP) * GEO_STRIDE + 1];
}
// the constant row of the inter-object edges, c0 = [lin3(0) x 2H | radial_lin(0) | 0 x R]  (k_c0row):  adjoint into the biases /
// last layers it is made of.  dc0 [W] = sum of dew over the inter-object rows (+ the NeighborEmb share of the f section in dc0f_extra)
OARD_KERNEL __global__ __launch_bounds__(256) void k_c0_bwd(const float* __restrict__ dc0, const float* __restrict__ dc0f_extra, int H, int H4,
                                                const float* __restrict__ l3b0, const float* __restrict__ l3w2, const float* __restrict__ rl0b,
                                                const float* __restrict__ rl2w, float* __restrict__ g_l3b0, float* __restrict__ g_l3w2,
                                                float* __restrict__ g_l3b2, float* __restrict__ g_rl0b, float* __restrict__ g_rl2w,
                                                float* __restrict__ g_rl2b) {
    __shared__ float red[256];
    __shared__ float dcf[256];
    const int t = threadIdx.x;
    // c0s = lin3.2(SiLU(lin3.0.bias)) + lin3.2.bias, replicated over the first 2H columns
    float s = 0.f;
    for (int c = t; c < 2 * H; c += 256) s += dc0[c];
    red[t] = s;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) { if (t < d) red[t] += red[t + d]; __syncthreads(); }
    const float dc0s = red[0];
    if (t < H4) {
        const float b = l3b0[t];
        if (g_l3w2) g_l3w2[t] += dc0s * silu_acc(b);
        if (g_l3b0) g_l3b0[t] += dc0s * l3w2[t] * dsilu1(b);
    }
    if (t == 0 && g_l3b2) g_l3b2[0] += dc0s;
    // c0f = radial_lin.2(SiLU(radial_lin.0.bias)) + radial_lin.2.bias
    for (int c = t; c < 256; c += 256) dcf[c] = c < H ? dc0[2 * H + c] + (dc0f_extra ? dc0f_extra[c] : 0.f) : 0.f;
    __syncthreads();
    if (t < H) {
        if (g_rl2b) g_rl2b[t] += dcf[t];
        const float hb = silu_acc(rl0b[t]);                  // this thread: hidden unit t
        float dh = 0.f;
        for (int o = 0; o < H; ++o) {
            dh += rl2w[o * H + t] * dcf[o];
            if (g_rl2w) g_rl2w[o * H + t] += dcf[o] * hb;
        }
        if (g_rl0b) g_rl0b[t] += dh * dsilu1(rl0b[t]);
    }
}
// encoders (k_prep; egnn_dynamics.py:95-104): Linear(d, 2d) SiLU Linear(2d, emb) per node; operands of the weight gradients in reference order
struct XhPtrs { const float* p[OARD_MAX_OBJECTS]; };
OARD_KERNEL __global__ void k_prep_bwd(TopoDev tp, TailPtrs tl, XhPtrs xh, const float* __restrict__ wb,
                           const float* __restrict__ dhin, int emb, float* __restrict__ op_dy, float* __restrict__ op_hid,
                           float* __restrict__ op_dz, float* __restrict__ op_x) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= tp.N) return;
    const int obj = tp.node_obj[n], row = tp.node_row[n], nf = tl.node_nf[obj], d = nf - 3;
    const float* xin = xh.p[obj] + (size_t)row * nf + 3;
    const float* W0 = wb + tl.enc[obj];
    const float* b0 = W0 + 2 * d * d;
    const float* W1 = b0 + 2 * d;
    const size_t r = (size_t)tp.node_ref[n];
    const float* dy = dhin + (size_t)n * 16;
    for (int k = 0; k < 32; ++k) {
        float z = 0.f, dhid = 0.f;
        if (k < 2 * d) {
            for (int i = 0; i < d; ++i) z += W0[k * d + i] * xin[i];
            z += b0[k];
            for (int o = 0; o < emb; ++o) dhid += W1[o * 2 * d + k] * dy[o];
        }
        op_hid[r * 32 + k] = k < 2 * d ? silu_acc(z) : 0.f;
        op_dz[r * 32 + k] = k < 2 * d ? dhid * dsilu1(z) : 0.f;
    }
    for (int i = 0; i < 16; ++i) {
        op_dy[r * 16 + i] = i < emb ? dy[i] : 0.f;
        op_x[r * 16 + i] = i < d ? xin[i] : 0.f;
    }
}

template <class D>
static int tr_init_bwd(const TrainCtx& x, const TrainTail& tw, const oard_topology* topo, const float* const* xh, const float* ds0_in,
                       const float* dew) {
    const TopoDev& tp = *x.tp;
    const int N = tp.N, HP = D::HP, H = D::H, HT = D::HT, W = D::W, WP = D::WP, C = x.c->in_hidden, emb = embed_dim(x.c);
    const long long A = tp.A, E = tp.E;
    const float* hin = x.t(x.to.hin);
    const float* s0 = x.t(x.to.s_in[0]);
    const float* ew0 = x.t(x.to.ew[0]);
    const float* geo = x.t(x.to.geo);
    const float* rbf = x.t(x.to.rbuf);
    const float* c0f = x.wb + x.po.c0row + 2 * H;
    float *ynb = x.f(tw.ynb), *nbe = x.f(tw.nbe), *ys = x.f(tw.ys), *lns = x.f(tw.lns), *s1v = x.f(tw.s1v), *ne1 = x.f(tw.ne1),
          *dne1 = x.f(tw.dne1), *part = x.f(tw.part), *zrl = x.f(tw.zrl), *hrl = x.f(tw.hrl), *df = x.f(tw.df), *ds1v = x.f(tw.ds1v),
          *dlns = x.f(tw.dlns), *dys = x.f(tw.dys), *ds0 = x.f(tw.ds0), *dnbe = x.f(tw.dnbe), *ipro = x.f(tw.ipro), *dynb = x.f(tw.dynb),
          *dhin = x.f(tw.dhin), *dfr = x.f(tw.dfr), *dzrl = x.f(tw.dzrl), *dc0 = x.f(tw.dc0);
    const long long NH = (long long)N * HP;
    // ---- forward pieces ----
    rows_dense<1>(x, N, hin, 16, x.wb + x.po.nbemb, HT, ynb, HP, x.wb + x.po.nbemb_b);                          // NeighborEmb.embedding   :82
    rows_dense<HT>(x, N, s0, HP, x.wb + x.po.s2v, HT, ys, HP, x.wb + x.po.s2v_b);                                // s2v.lin1                :116
    { ScopedLaunch sl_(F_INIT, x.st);
      hipLaunchKernelGGL(k_rows_ln_fwd, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, x.st, (const float*)ynb, (const float*)nullptr, HP, H, HP,
                         (const float*)nullptr, (const float*)nullptr, nbe, (float*)nullptr, (long long)N);
      hipLaunchKernelGGL(k_rows_ln_fwd, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, x.st, (const float*)ys, (const float*)nullptr, HP, H, HP,
                         (const float*)nullptr, (const float*)nullptr, lns, (float*)nullptr, (long long)N);
      hipLaunchKernelGGL(k_silu_rows, EW_GRID(NH), 0, x.st, (const float*)lns, NH, s1v); }
    if (A > 0) {
        TopoDev tp16 = tp;
        tp16.npb = 16;
        constexpr int NW = D::HT < 13 ? D::HT : 13;
        LAUNCH(F_INIT, (k_s2v_agg_v1<D, NW>), cdiv(N, 16), NW * 64, x.st, tp16, (const float*)s1v, ew0, geo, ne1);
        rows_dense<D::RB, EPI_SILU>(x, A, rbf, D::RP, x.wb + x.po.rl0, HT, hrl, HP, x.wb + x.po.rl0_b, nullptr, 0, D::RB, nullptr, 0, zrl, HP);
        // ---- edge scalarisation + lin3 (k_scalarize_bwd): d NE1 and the lin3 gradients (per-node partials) ----
        TR_TRY(scalarize_backward_impl<D>(x.c, tp, x.wb, x.tape, x.to, ne1, HP, dew, dne1, part, x.st));
        LAUNCH(F_INIT, (k_s2v_bwd<D>), N, 256, x.st, tp, geo, ew0, (const float*)s1v, (const float*)dne1, dew, df, ds1v);
    } else {
        HIP_TRY(hipMemsetAsync(ds1v, 0, (size_t)NH * sizeof(float), x.st));
    }
    // s1v = SiLU(LN0(ys)):  dys = LN0^T (ds1v SiLU'(lns));  ds0 = ds0_in + s2v^T dys
    { ScopedLaunch sl_(F_INIT, x.st);
      hipLaunchKernelGGL(k_mul_dsilu_rows, EW_GRID(NH), 0, x.st, (const float*)ds1v, (const float*)lns, NH, dlns);
      hipLaunchKernelGGL(k_rows_ln_bwd, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, x.st, (const float*)ys, HP, H, HP, (const float*)nullptr,
                         (const float*)dlns, (const float*)nullptr, dys, (float*)nullptr, (long long)N); }
    rows_dense<HT, EPI_ADD>(x, N, dys, HP, x.pb + x.nb.s2vT, HT, ds0, HP, nullptr, nullptr, 0, HT, ds0_in, HP);
    // NeighborEmb adjoint
    LAUNCH(F_INIT, k_seg_sums, tp.B, 256, x.st, tp, (const float*)ds0, HP, x.f(tw.segS), x.f(tw.segG));
    LAUNCH(F_INIT, k_seg_sums, tp.B, 256, x.st, tp, (const float*)nbe, HP, x.f(tw.segSn), x.f(tw.segGn));
    LAUNCH(F_INIT, (k_nbr_bwd<D>), N, 256, x.st, tp, ew0, c0f, (const float*)nbe, (const float*)ds0, (const float*)x.f(tw.segS),
           (const float*)x.f(tw.segG), (const float*)x.f(tw.segSn), (const float*)x.f(tw.segGn), df, dnbe, ipro);
    { ScopedLaunch sl_(F_INIT, x.st);
      hipLaunchKernelGGL(k_rows_ln_bwd, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, x.st, (const float*)ynb, HP, H, HP, (const float*)nullptr,
                         (const float*)dnbe, (const float*)nullptr, dynb, (float*)nullptr, (long long)N); }
    // d hin = embedding^T ds0 + neighbor_emb.embedding^T dynb  (only the encoder columns [0, emb) are used)
    rows_dense<HT>(x, N, ds0, HP, x.pb + x.nb.embT, 1, dhin, 16);
    rows_dense<HT, EPI_ADD>(x, N, dynb, HP, x.pb + x.nb.nbembT, 1, dhin, 16, nullptr, nullptr, 0, HT, dhin, 16);
    // radial_lin on the inner edges: f = (rl2 SiLU(rl0 rbf + b0) + b2) env
    if (A > 0) {
        { ScopedLaunch sl_(F_INIT, x.st);
          hipLaunchKernelGGL(k_scale_by_env, EW_GRID(A * HP), 0, x.st, (const float*)df, geo, A, HP, dfr); }
        rows_dense<HT, EPI_MUL_DSILU>(x, A, dfr, HP, x.pb + x.nb.rl2T, HT, dzrl, HP, nullptr, nullptr, 0, HT, zrl, HP);
    }
    // encoders: per-node adjoint, operands of the weight gradients in reference row order
    TailPtrs tl;
    memset(&tl, 0, sizeof(tl));
    XhPtrs xp;
    memset(&xp, 0, sizeof(xp));
    for (int k = 0; k < x.c->n_obj; ++k) { tl.node_nf[k] = x.c->node_nf[k]; tl.dec[k] = x.po.dec[k]; tl.enc[k] = x.po.enc[k]; xp.p[k] = xh[k]; }
    LAUNCH(F_INIT, k_prep_bwd, cdiv(N, 128), 128, x.st, tp, tl, xp, x.wb, (const float*)dhin, emb, x.f(tw.eo_dy), x.f(tw.eo_hid), x.f(tw.eo_dz),
           x.f(tw.eo_x));
    // ---- parameter gradients (gradient stream) ----
    x.fork();
    if (A > 0) {
        const int H4 = D::H4;
        colsum(x, part, 5 * H4 + 1, 0, N, 3 * H4, x.g(x.pi.lin30_w));
        colsum(x, part + 3 * H4, 5 * H4 + 1, 0, N, H4, x.g(x.pi.lin30_b));
        colsum(x, part + 4 * H4, 5 * H4 + 1, 0, N, H4, x.g(x.pi.lin32_w));
        colsum(x, part + 5 * H4, 5 * H4 + 1, 0, N, 1, x.g(x.pi.lin32_b));
        TR_TRY(wg(x, dfr, HP, HP, H, HP, H, hrl, HP, HP, 0, H, HP, H, A, x.g(x.pi.rl2_w), H, x.g(x.pi.rl2_b)));
        TR_TRY(wg(x, dzrl, HP, HP, H, HP, H, rbf, D::RP, D::RP, 0, D::R, D::RP, D::R, A, x.g(x.pi.rl0_w), D::R, x.g(x.pi.rl0_b)));
    }
    TR_TRY(wg(x, dys, HP, HP, H, HP, H, s0, HP, HP, 0, H, HP, H, N, x.g(x.pi.s2v_w), H, x.g(x.pi.s2v_b)));
    TR_TRY(wg(x, dynb, HP, HP, H, HP, H, hin, 16, 16, 0, C, 16, C, N, x.g(x.pi.nbemb_w), C, x.g(x.pi.nbemb_b)));
    TR_TRY(wg(x, ds0, HP, HP, H, HP, H, hin, 16, 16, 0, C, 16, C, N, x.g(x.pi.emb_w), C, x.g(x.pi.emb_b)));            // z_emb = embedding(hin)  :744
    // the constant row of the inter-object edges
    colsum(x, dew, WP, A, E, W, dc0, 0);
    colsum(x, ipro, HP, 0, N, H, x.f(x.w.csum), 0);
    if (H > 256 || x.params == nullptr) return OARD_EINVAL;
    LAUNCH(F_INIT, k_c0_bwd, 1, 256, x.stw, (const float*)dc0, (const float*)x.f(x.w.csum), H, D::H4, x.params[x.pi.lin30_b],
           x.params[x.pi.lin32_w], x.params[x.pi.rl0_b], x.params[x.pi.rl2_w], x.g(x.pi.lin30_b), x.g(x.pi.lin32_w), x.g(x.pi.lin32_b),
           x.g(x.pi.rl0_b), x.g(x.pi.rl2_w), x.g(x.pi.rl2_b));
    for (int k = 0; k < x.c->n_obj; ++k) {                 // Linear(d, 2d) SiLU Linear(2d, emb)
        const int a = x.c->enc_alias[k], d = x.c->node_nf[k] - 3, q = x.pi.enc0 + 4 * a;
        const long long r0 = topo->obj_start[k], nk = topo->obj_start[k + 1] - r0;
        if (nk <= 0) continue;
        TR_TRY(wg(x, x.f(tw.eo_dz) + r0 * 32, 32, 32, 2 * d, 2 * d, 2 * d, x.f(tw.eo_x) + r0 * 16, 16, 16, 0, d, d, d, nk, x.g(q + 0), d, x.g(q + 1)));
        TR_TRY(wg(x, x.f(tw.eo_dy) + r0 * 16, 16, 16, emb, emb, emb, x.f(tw.eo_hid) + r0 * 32, 32, 32, 0, 2 * d, 2 * d, 2 * d, nk, x.g(q + 2), 2 * d, x.g(q + 3)));
    }
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}
