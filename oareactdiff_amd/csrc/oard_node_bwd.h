// oard_node_bwd.h — hand-written backward of init / node stages whose torch formulation is memory-bound.
//
// k_scalarize_bwd: adjoint of k_scalarize (edge scalarisation + lin3, leftnet.py:792-806):
//     S_c = <NE1[node], frame_c>,  S_1 = |S_1|,   out = (lin3(S) + S_0) * env,   lin3 = Linear(3, H/4) SiLU Linear(H/4, 1)
// evaluated for every (inner edge, side, channel).  The per-(edge, channel) MLP is tiny but there are 2 A H ~ 4e7 items
// at B = 64: an eager formulation materialises [items, H/4] tensors several times (hundreds of ms); here the hidden
// layer is ONE MFMA per 16 hidden units: A = [W0 | b0] (16 hidden x 4), B = (S_0, S_1, S_2, 1) x 16 channels, so a
// lane (g, j) holds the pre-activations of hidden units 16q + 4g + r for channel j (C layout) and keeps its own
// partial sums of the weight gradients in registers.  A workgroup owns one node n and all its inner edges (as
// target: side 1, as source: side 0), so d NE1[n] is accumulated in registers in a fixed order - no atomics.
#pragma once
#include "oard_kernels.h"
#include "oard_edge_bwd.h"

template <class D>
struct ScalarizeBwd {
    static constexpr int H4 = D::H4, HQ = (H4 + 15) / 16;
    static constexpr int NPART = 5 * H4 + 1;      // per-node partial: dw0 [H4][3] | db0 [H4] | dw2 [H4] | db2
    // rows per lane of hidden block q (a lane (g, j) holds rows 16q + 4g + r): 4, except in a last block with fewer than 4 real
    // rows (H4 = 49: one row) - its padding rows are not evaluated (13 instead of 16 evaluations per item)
    static constexpr int R_LAST = (H4 - 16 * (HQ - 1)) < 4 ? (H4 - 16 * (HQ - 1)) : 4;
    OARD_DEV static constexpr int rows(int q) { return q == HQ - 1 ? R_LAST : 4; }
};

// ne1: [N][3][ld] (ld >= H), gdew: gradient of the initial edge state, rows = inner edges, row stride WP, columns [0, 2H)
// dne1: [N][3][ld] out;  part: [N][NPART] out (summed over the node's edges and channels; the caller adds the nodes up)
//
// Round 4 (second version).  The first version spent 33 instructions per hidden-unit evaluation (the compiler packed the four rows of
// an MFMA result into v_pk_* operands through register moves), waited for three dependent loads per item and dealt 13 channel tiles
// to 4 waves (4 + 3 + 3 + 3).  Now: (a) the rows of a lane are processed as the register PAIRS the MFMA result already consists of
// (sc_eval<f2>: 16 packed instructions + 4 transcendentals per two evaluations), (b) a unit of work is (channel tile, side) - 2 HT
// units, 13 per wave at 2 waves per node - and the per-lane partial sums of d NE1 are only reduced over the k-groups once per unit,
// (c) the geometry row and the gradient element of the NEXT item are loaded before the current item is evaluated.
#ifndef OARD_SCAL_BWD_MINW
#define OARD_SCAL_BWD_MINW 2
#endif
template <class T> OARD_DEV T sc_exp2(T v);
template <> OARD_DEV float sc_exp2<float>(float v) { return __builtin_amdgcn_exp2f(v); }
template <> OARD_DEV f2 sc_exp2<f2>(f2 v) { return (f2){__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)}; }
template <class T> OARD_DEV T sc_rcp(T v);
template <> OARD_DEV float sc_rcp<float>(float v) { return __builtin_amdgcn_rcpf(v); }
template <> OARD_DEV f2 sc_rcp<f2>(f2 v) { return (f2){__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)}; }
// one (T = float) or two (T = f2) hidden rows of one item: z = pre-activation, w2 / w0 = the rows' constants
template <class T>
OARD_DEV void sc_eval(T z, float dout, float S0, float S1, float S2, T w2, const T (&w0)[3], T& aw2, T& ab0, T (&aw0)[3], T (&p)[3]) {
    const T sg = sc_rcp<T>(sc_exp2<T>(z * -1.44269504088896340736f) + 1.0f);
    aw2 += (z * sg) * dout;
    const T dz = (w2 * dout) * (sg * (z * (1.0f - sg) + 1.0f));
    ab0 += dz;
    aw0[0] += dz * S0; aw0[1] += dz * S1; aw0[2] += dz * S2;
    p[0] += dz * w0[0]; p[1] += dz * w0[1]; p[2] += dz * w0[2];
}

template <class D, int WAVES, bool SIGNED = false>
__global__ __launch_bounds__(WAVES * 64, OARD_SCAL_BWD_MINW) void k_scalarize_bwd(TopoDev tp, const float* __restrict__ l3, const float* __restrict__ ne1,
                                                              int ld, const float* __restrict__ geo, const float* __restrict__ gdew,
                                                              float* __restrict__ dne1, float* __restrict__ part) {
    constexpr bool signed_scal = SIGNED;
    using SB = ScalarizeBwd<D>;
    constexpr int H4 = SB::H4, HQ = SB::HQ, HT = D::HT;
    __shared__ float red[WAVES][4 * HQ * 4 * 5 + 4];
    __shared__ float dpart[2 * HT][3][16];
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform for the compiler: scalar address arithmetic
    const int n = blockIdx.x;
    const float* w0 = l3;
    const float* b0 = l3 + H4 * 3;
    const float* w2 = b0 + H4;
    // A operand: [W0 | b0] rows 16q + j, column g
    float aw[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        const int row = 16 * q + j;
        aw[q] = row < H4 ? (g < 3 ? w0[row * 3 + g] : b0[row]) : 0.f;
    }
    // constants and accumulators of this lane's rows 16q + 4g + r, as pairs (r = 0, 1), (r = 2, 3)
    f2 w2r[HQ][2], w0r[HQ][2][3], aw0[HQ][2][3], ab0[HQ][2], aw2[HQ][2];
    float ab2 = 0.f;
#pragma unroll
    for (int q = 0; q < HQ; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * q + 4 * g + r;
            const bool ok = r < SB::rows(q) && row < H4;
            w2r[q][r >> 1][r & 1] = ok ? w2[row] : 0.f;
            ab0[q][r >> 1][r & 1] = 0.f; aw2[q][r >> 1][r & 1] = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) { w0r[q][r >> 1][c][r & 1] = ok ? w0[row * 3 + c] : 0.f; aw0[q][r >> 1][c][r & 1] = 0.f; }
        }

    const int q_grp = tp.node_sample[n] * tp.n_obj + tp.node_obj[n];
    const int g0 = tp.grp_ptr[q_grp], ng = tp.grp_ptr[q_grp + 1] - g0, self = n - g0;
    const int a_n = tp.act_ptr[n];
    const int ni = ng - 1;                                    // items of a unit: the other members of the group
    for (int u = wave; u < 2 * HT; u += WAVES) {
        const int t = u >> 1, side = u & 1;
        const int ch = 16 * t + j;
        const bool chok = ch < D::H;
        const float n0 = chok ? ne1[((size_t)n * 3 + 0) * ld + ch] : 0.f;
        const float n1 = chok ? ne1[((size_t)n * 3 + 1) * ld + ch] : 0.f;
        const float n2 = chok ? ne1[((size_t)n * 3 + 2) * ld + ch] : 0.f;
        // side 1: edge (m -> n), n is the target;  side 0: edge (n -> m), n is the source.  Item i is group member kk = i + (i >= self);
        // its row is a_n + i (side 1) or act_ptr[member] + (position of n in the member's list) (side 0).
        // zv is 0 in every lane but opaque to the compiler: rows count as per-lane values, so the geometry rows come by VECTOR loads
        // (every lane the same address: one transaction), which return in order - the wait for item i + 1 is a vmcnt at the end of
        // item i, where a scalar load would be waited for with lgkmcnt(0) wherever the next scalar load result is needed - and the
        // loop body has no branch for hipcc to sink the loads behind.
        const int zv = (int)__builtin_amdgcn_mbcnt_lo(0u, 0u);
        auto member = [&](int i) { const int ii = (i < ni ? i : ni - 1) + zv; return g0 + ii + (ii >= self ? 1 : 0); };
        auto row_of = [&](int i, int act) {
            const int ii = (i < ni ? i : ni - 1) + zv, kk = ii + (ii >= self ? 1 : 0);
            return side == 1 ? a_n + ii : act + self - (self > kk ? 1 : 0);
        };
        const int chc = side * D::H + (chok ? ch : 0);          // lanes beyond H read a valid element and discard it
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
        if (ni > 0) {
            int a_nx = row_of(0, tp.act_ptr[member(0)]);
            // env u[3] c[3] v[3] of the row, and nothing else: a loaded register that is never read gets reused by the compiler, which
            // then has to wait for the load to land first (write-after-write)
            const float* gp = geo + (size_t)a_nx * GEO_STRIDE + 1;
            float gc[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) gc[k] = gp[k];
            float go = gdew[(size_t)a_nx * D::WP + chc];
            a_nx = row_of(1, tp.act_ptr[member(1)]);
            for (int i = 0; i < ni; ++i) {
                // operands of item i + 1 and the list pointer of item i + 2 travel while item i is evaluated (the last iterations re-read
                // the last item)
                gp = geo + (size_t)a_nx * GEO_STRIDE + 1;
                float gn[10];
#pragma unroll
                for (int k = 0; k < 10; ++k) gn[k] = gp[k];
                const float gon = gdew[(size_t)a_nx * D::WP + chc];
                const int act_nx = tp.act_ptr[member(i + 2)];
                __builtin_amdgcn_sched_barrier(0);              // the loads stay up here (hipcc otherwise sinks them to the end of the body)
                const float env = gc[0];
                const float ux = gc[1], uy = gc[2], uz = gc[3], cx = gc[4], cy = gc[5], cz = gc[6], vx = gc[7], vy = gc[8], vz = gc[9];
                const float S0 = n0 * ux + n1 * uy + n2 * uz;
                const float S1r = n0 * cx + n1 * cy + n2 * cz;
                const float S2 = n0 * vx + n1 * vy + n2 * vz;
                const float S1 = signed_scal ? S1r : fabsf(S1r);
                const float bval = g == 0 ? S0 : (g == 1 ? S1 : (g == 2 ? S2 : 1.0f));
                const float dout = chok ? go * env : 0.f;
                f2 pp[3] = {(f2){0.f, 0.f}, (f2){0.f, 0.f}, (f2){0.f, 0.f}};
                float ps[3] = {0.f, 0.f, 0.f};
                f4 zq[HQ];
#pragma unroll
                for (int q = 0; q < HQ; ++q) zq[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[q], bval, f4zero(), 0, 0, 0);
#pragma unroll
                for (int q = 0; q < HQ; ++q) {
                    const f4 z = zq[q];
                    const int rows = SB::rows(q);
                    if (rows >= 2) sc_eval<f2>((f2){z.x, z.y}, dout, S0, S1, S2, w2r[q][0], w0r[q][0], aw2[q][0], ab0[q][0], aw0[q][0], pp);
                    if (rows == 4) sc_eval<f2>((f2){z.z, z.w}, dout, S0, S1, S2, w2r[q][1], w0r[q][1], aw2[q][1], ab0[q][1], aw0[q][1], pp);
                    if (rows == 1 || rows == 3) {              // an odd last row: scalar
                        const int pi = rows >> 1;
                        float w0s[3] = {w0r[q][pi][0].x, w0r[q][pi][1].x, w0r[q][pi][2].x};
                        float a0s[3] = {aw0[q][pi][0].x, aw0[q][pi][1].x, aw0[q][pi][2].x};
                        float a2s = aw2[q][pi].x, b0s = ab0[q][pi].x;
                        sc_eval<float>(rows == 1 ? z.x : z.z, dout, S0, S1, S2, w2r[q][pi].x, w0s, a2s, b0s, a0s, ps);
                        aw0[q][pi][0].x = a0s[0]; aw0[q][pi][1].x = a0s[1]; aw0[q][pi][2].x = a0s[2];
                        aw2[q][pi].x = a2s; ab0[q][pi].x = b0s;
                    }
                }
                // this lane's share (its hidden rows) of d S; the k-groups are added up once per unit
                const float p0 = pp[0].x + pp[0].y + ps[0], p1 = pp[1].x + pp[1].y + ps[1], p2 = pp[2].x + pp[2].y + ps[2];
                const float dS0 = p0 + (g == 0 ? dout : 0.f);
                const float dS1 = signed_scal ? p1 : (S1r < 0.f ? -p1 : (S1r > 0.f ? p1 : 0.f));
                d0 += dS0 * ux + dS1 * cx + p2 * vx;
                d1 += dS0 * uy + dS1 * cy + p2 * vy;
                d2 += dS0 * uz + dS1 * cz + p2 * vz;
                if (g == 0) ab2 += dout;
                __builtin_amdgcn_sched_barrier(0);
                a_nx = row_of(i + 2, act_nx);
#pragma unroll
                for (int k = 0; k < 10; ++k) gc[k] = gn[k];
                go = gon;
            }
        }
        d0 = col_reduce(d0); d1 = col_reduce(d1); d2 = col_reduce(d2);
        if (g == 0) { dpart[u][0][j] = d0; dpart[u][1][j] = d1; dpart[u][2][j] = d2; }
    }
    // reduce the weight-gradient partials over the 16 channel lanes, then over the waves
    auto red16 = [](float v) {
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        return v;
    };
#pragma unroll
    for (int q = 0; q < HQ; ++q)
#pragma unroll
        for (int r = 0; r < SB::rows(q); ++r) {
            const float s0 = red16(aw0[q][r >> 1][0][r & 1]), s1 = red16(aw0[q][r >> 1][1][r & 1]), s2 = red16(aw0[q][r >> 1][2][r & 1]);
            const float sb = red16(ab0[q][r >> 1][r & 1]), sw = red16(aw2[q][r >> 1][r & 1]);
            if (j == 0) {
                float* dst = &red[wave][((q * 4 + g) * 4 + r) * 5];        // row 16q + 4g + r
                dst[0] = s0; dst[1] = s1; dst[2] = s2; dst[3] = sb; dst[4] = sw;
            }
        }
    ab2 = red16(ab2);
    if (lane == 0) red[wave][4 * HQ * 4 * 5] = ab2;
    __syncthreads();
    // d NE1[n]: side 0 + side 1 of every channel tile
    for (int i = threadIdx.x; i < HT * 3 * 16; i += WAVES * 64) {
        const int t = i / 48, c = (i % 48) >> 4, jj = i & 15, ch = 16 * t + jj;
        if (ch < D::H) dne1[((size_t)n * 3 + c) * ld + ch] = dpart[2 * t][c][jj] + dpart[2 * t + 1][c][jj];
    }
    float* out = part + (size_t)n * SB::NPART;
    for (int i = threadIdx.x; i < SB::NPART; i += WAVES * 64) {
        float s = 0.f;
        int slot;
        if (i < 3 * H4) { const int row = i / 3, c = i % 3; slot = row * 5 + c; }
        else if (i < 4 * H4) slot = (i - 3 * H4) * 5 + 3;
        else if (i < 5 * H4) slot = (i - 4 * H4) * 5 + 4;
        else slot = 4 * HQ * 4 * 5;
        for (int w = 0; w < WAVES; ++w) s += red[w][slot];
        out[i] = s;
    }
}

// =====================================================================================================
// k_equi_msg_bwd: adjoint of the message formation + aggregation of EquiMessage (leftnet.py:264-283, the first part of
// k_equi_node_v1):  per inner edge a = (m -> n), per channel
//     xs_k = xq[m][k] + xq[n][k],  q_k = cd_k cr_k,   dx[n] += xs_0 q_0,   a2 = xs_1 q_1 / sqrt3,   a3 = xs_2 q_2,
//     dvec[n][x] += (vec[m][x] a2 + a3 u_a[x]) / sqrt(H)
// given gX = d/d(dx) and gV = d/d(dvec).  One workgroup per node, one thread per channel; the node is visited in both
// of its roles - as the target of its incoming edges (writes the per-edge gradients d cd, d cr, accumulates d xq[n]) and
// as the source of its outgoing edges (accumulates d xq[n] and d vec[n]) - so every sum runs in a fixed order in
// registers and nothing is scattered with atomics.  The [A, 3H]-sized gather / product / scatter chain this replaces is
// what an eager formulation spends most of its node-stage time on.
// =====================================================================================================
struct Strided3 {            // element (row, k, ch) of a [rows][3][channels] tensor with arbitrary strides (multiples of 4 floats)
    const float* p;
    int row, comp;
    OARD_DEV float at(size_t r, int k, int ch) const { return p[r * (size_t)row + (size_t)k * comp + ch]; }
    OARD_DEV f4 at4(size_t r, int k, int ch) const { return ld_f4(p + r * (size_t)row + (size_t)k * comp + ch); }
};

// Round 4 (second version): four channels per lane (float4 accesses: 49 lanes cover H = 196), the four waves of the workgroup take
// the group members j = wave (mod 4), both roles of a member in one iteration with all 22 float4 operands requested before the
// first use; the waves' partial sums of d xq[n] / d vec[n] are added in wave order.  Was: one channel per thread (the fourth wave
// of a node held 4 channels), one role at a time, 2 (ng - 1) dependent round trips per thread: 307 us per layer at B = 64.
template <class D>
__global__ __launch_bounds__(256) void k_equi_msg_bwd(TopoDev tp, const float* __restrict__ geo, Strided3 xq, Strided3 vec,
                                                      const float* __restrict__ cd, Strided3 cr, const float* __restrict__ gX,
                                                      int gx_row, Strided3 gV, float* __restrict__ dcd, float* __restrict__ dcr,
                                                      float* __restrict__ dxq, float* __restrict__ dvec, int o_comp, int xcross, int zero_spare = 0) {
    // zero_spare: dcd / dcr have a spare row A (padding columns of the consumer's last MFMA tile read it): written as zeros here
    // xcross: reflect_equiv = False (the message's x (x) coord_cross term, leftnet.py:268-272)
    // outputs d xq / d vec: [N][3][o_comp] (o_comp = H: dense; o_comp = HP: padded, the pads are written as zeros)
    static_assert(D::H % 4 == 0 && D::HP % 4 == 0 && D::HP <= 256, "four channels per lane, one wave per row");
    __shared__ f4 part[3][6][64];
    const int n = OARD_XCD_RUNS ? xcd_run32(blockIdx.x, gridDim.x) : (int)blockIdx.x, lane = threadIdx.x & 63, ch = 4 * lane;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool real = ch < D::H, pad = !real && ch < o_comp;
    const float inv_sqrt3 = 0.57735026918962576f, inv_sqrt_h = 1.0f / sqrtf((float)D::H);
    const int q_grp = tp.node_sample[n] * tp.n_obj + tp.node_obj[n];
    const int g0 = tp.grp_ptr[q_grp], ng = tp.grp_ptr[q_grp + 1] - g0, self = n - g0;
    const int a_n = tp.act_ptr[n];
    constexpr size_t ES = (size_t)3 * D::HP;              // row stride of cd / dcd / dcr
    if (zero_spare && blockIdx.x == 0)
        for (int i = threadIdx.x; i < (int)ES; i += 256) { dcd[(size_t)tp.A * ES + i] = 0.f; dcr[(size_t)tp.A * ES + i] = 0.f; }
    f4 ax[3] = {f4zero(), f4zero(), f4zero()};           // d xq[n]
    f4 av[3] = {f4zero(), f4zero(), f4zero()};           // d vec_in[n]
    if (real) {
        const f4 xn[3] = {xq.at4(n, 0, ch), xq.at4(n, 1, ch), xq.at4(n, 2, ch)};
        const f4 wn[3] = {vec.at4(n, 0, ch), vec.at4(n, 1, ch), vec.at4(n, 2, ch)};
        const f4 gxn = ld_f4(gX + (size_t)n * gx_row + ch);
        const f4 gvn[3] = {gV.at4(n, 0, ch), gV.at4(n, 1, ch), gV.at4(n, 2, ch)};
        if (wave == 0) { av[0] = gvn[0]; av[1] = gvn[1]; av[2] = gvn[2]; }      // identity path of vec_a = vec_in + dvec
        for (int j = wave; j < ng - 1; j += 4) {
            const int kk = j + (j >= self ? 1 : 0), m = g0 + kk;
            // edge t = (m -> n): n is the target;  edge s = (n -> m): n is the source
            const size_t at = (size_t)a_n + j, as = (size_t)tp.act_ptr[m] + self - (self > kk ? 1 : 0);
            const float* get = geo + at * GEO_STRIDE;
            const float* ges = geo + as * GEO_STRIDE;
            const float* cdt = cd + at * ES + ch;
            const float* cds = cd + as * ES + ch;
            const f4 ct[3] = {ld_f4(cdt), ld_f4(cdt + D::HP), ld_f4(cdt + 2 * D::HP)};
            const f4 rt[3] = {cr.at4(at, 0, ch), cr.at4(at, 1, ch), cr.at4(at, 2, ch)};
            const f4 cs[3] = {ld_f4(cds), ld_f4(cds + D::HP), ld_f4(cds + 2 * D::HP)};
            const f4 rs[3] = {cr.at4(as, 0, ch), cr.at4(as, 1, ch), cr.at4(as, 2, ch)};
            const f4 xm[3] = {xq.at4(m, 0, ch), xq.at4(m, 1, ch), xq.at4(m, 2, ch)};
            const f4 wm[3] = {vec.at4(m, 0, ch), vec.at4(m, 1, ch), vec.at4(m, 2, ch)};
            const f4 gxm = ld_f4(gX + (size_t)m * gx_row + ch);
            const f4 gvm[3] = {gV.at4(m, 0, ch), gV.at4(m, 1, ch), gV.at4(m, 2, ch)};
            const f4 xs[3] = {xm[0] + xn[0], xm[1] + xn[1], xm[2] + xn[2]};     // xq[source] + xq[target], the same for both edges
            // ---- edge t: cotangents of the target n, source-side vectors of m ----
            {
                const f4 q[3] = {ct[0] * rt[0], ct[1] * rt[1], ct[2] * rt[2]};
                const f4 ga2 = (gvn[0] * wm[0] + gvn[1] * wm[1] + gvn[2] * wm[2]) * inv_sqrt_h;
                const f4 ga3 = (gvn[0] * get[2] + gvn[1] * get[3] + gvn[2] * get[4]) * inv_sqrt_h;
                // reflect_equiv = False: the vector message holds x (x) coord_cross as well - its cotangent joins the scalar message's
                const f4 gxc = xcross ? gxn + (gvn[0] * get[5] + gvn[1] * get[6] + gvn[2] * get[7]) * inv_sqrt_h : gxn;
                const f4 dm[3] = {gxc, ga2 * inv_sqrt3, ga3};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    ax[k] += dm[k] * q[k];                                      // d xs -> d xq of BOTH end points; this is n's share
                    const f4 dq = dm[k] * xs[k];
                    st_f4(dcd + at * ES + (size_t)k * D::HP + ch, dq * rt[k]);
                    st_f4(dcr + at * ES + (size_t)k * D::HP + ch, dq * ct[k]);
                }
            }
            // ---- edge s: cotangents of the target m, source-side vectors of n ----
            {
                const f4 q[3] = {cs[0] * rs[0], cs[1] * rs[1], cs[2] * rs[2]};
                const f4 a2 = xs[1] * q[1] * inv_sqrt3;
                const f4 ga2 = (gvm[0] * wn[0] + gvm[1] * wn[1] + gvm[2] * wn[2]) * inv_sqrt_h;
                const f4 ga3 = (gvm[0] * ges[2] + gvm[1] * ges[3] + gvm[2] * ges[4]) * inv_sqrt_h;
                const f4 gxc = xcross ? gxm + (gvm[0] * ges[5] + gvm[1] * ges[6] + gvm[2] * ges[7]) * inv_sqrt_h : gxm;
                ax[0] += gxc * q[0]; ax[1] += (ga2 * inv_sqrt3) * q[1]; ax[2] += ga3 * q[2];
#pragma unroll
                for (int x = 0; x < 3; ++x) av[x] += gvm[x] * a2 * inv_sqrt_h;  // d vec[source = n]
            }
        }
    } else if (pad) {
        // padded mode: the per-edge gradients feed an MFMA kernel next (W^T rows of the pads are zero, but 0 x NaN is NaN)
        for (int j = wave; j < ng - 1; j += 4) {
            const size_t at = (size_t)a_n + j;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                st_f4(dcd + at * ES + (size_t)k * D::HP + ch, f4zero());
                st_f4(dcr + at * ES + (size_t)k * D::HP + ch, f4zero());
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { part[wave - 1][k][lane] = ax[k]; part[wave - 1][3 + k][lane] = av[k]; }
    }
    __syncthreads();
    if (wave == 0 && (real || pad)) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int k = 0; k < 3; ++k) { ax[k] += part[w][k][lane]; av[k] += part[w][3 + k][lane]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            st_f4(dxq + ((size_t)n * 3 + k) * o_comp + ch, ax[k]);
            st_f4(dvec + ((size_t)n * 3 + k) * o_comp + ch, av[k]);
        }
    }
}

// =====================================================================================================
// EquiUpdate's frame-scalar MLP (leftnet.py:304-310, 333) as its own differentiable op (training):
//     out = lin3.4( SiLU( lin3.2( SiLU( lin3.0 (x, 0, 0) ) ) ) )          per (node, channel) item, x = <vec1, x1>
// (the node frame is [x1, 0, 0] exactly, so rows 1, 2 of the scalarisation are zero and |0| = 0: only column 0 of
// lin3.0.weight takes part).  N*H ~ 8.7e5 items of a 1 -> 48 -> 8 -> 1 MLP: in eager torch the K = 3 / N = 3 GEMMs and the
// [items, 48] reductions cost ~3 ms per layer; here one thread per item, weights through scalar loads.
// The backward kernel writes, per item, the operands of the weight gradients in row-major arrays and oard_wgrad (the same
// deterministic GEMM the edge stages use) reduces them:  xa = (x, 1, 0, 0), h1 [48], dz1 [48], dz2 [8] and
// h2a = dout * (h2[8], 1, 0, 0, 0), whose column sums are the gradients of lin3.4's weight and bias.
// p: raw parameter block  w0[48][3] b0[48] w2[8][48] b2[8] w4[8] b4[1]  (LayerOff::l3u)
// =====================================================================================================
OARD_KERNEL __global__ __launch_bounds__(256) void k_lin3u_fwd(const float* __restrict__ p, const float* __restrict__ x, long long n,
                                                   float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = lin3u(p, x[i]);
}

OARD_KERNEL __global__ __launch_bounds__(256) void k_lin3u_bwd(const float* __restrict__ p, const float* __restrict__ x,
                                                   const float* __restrict__ dout, long long n, float* __restrict__ dx,
                                                   float* __restrict__ xa, float* __restrict__ h1o, float* __restrict__ dz1o,
                                                   float* __restrict__ h2a, float* __restrict__ dz2o) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* w0 = p;
    const float* b0 = p + 144;
    const float* w2 = p + 192;
    const float* b2 = p + 576;
    const float* w4 = p + 584;
    const float xv = x[i], g = dout[i];
    float z2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) z2[j] = b2[j];
    float* h1row = h1o + i * 48;
    for (int k0 = 0; k0 < 48; k0 += 4) {
        f4 h;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = k0 + c;
            const float hk = silu1(w0[3 * k] * xv + b0[k]);
            h[c] = hk;
#pragma unroll
            for (int j = 0; j < 8; ++j) z2[j] += w2[j * 48 + k] * hk;
        }
        st_f4(h1row + k0, h);
    }
    float dz2[8];
    f4 ha = f4zero(), hb = f4zero();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float h2 = silu1(z2[j]);
        dz2[j] = g * w4[j] * dsilu1(z2[j]);
        if (j < 4) ha[j] = g * h2; else hb[j - 4] = g * h2;
    }
    st_f4(h2a + i * 12, ha); st_f4(h2a + i * 12 + 4, hb); st_f4(h2a + i * 12 + 8, (f4){g, 0.f, 0.f, 0.f});
    st_f4(dz2o + i * 8, (f4){dz2[0], dz2[1], dz2[2], dz2[3]}); st_f4(dz2o + i * 8 + 4, (f4){dz2[4], dz2[5], dz2[6], dz2[7]});
    st_f4(xa + i * 4, (f4){xv, 1.f, 0.f, 0.f});
    float dxv = 0.f;
    float* dz1row = dz1o + i * 48;
    for (int k0 = 0; k0 < 48; k0 += 4) {
        f4 d;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = k0 + c;
            float dh = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) dh += w2[j * 48 + k] * dz2[j];
            const float dz = dh * dsilu1(w0[3 * k] * xv + b0[k]);
            d[c] = dz;
            dxv += w0[3 * k] * dz;
        }
        st_f4(dz1row + k0, d);
    }
    dx[i] = dxv;
}


// =====================================================================================================
// k_lin3u_bwd_fused (round 3): the same adjoint WITH its weight gradients.  k_lin3u_bwd leaves 120 floats per item in HBM
// (h1, dz1, dz2, h2a, xa: 440 MB per layer at B = 64) for two k_wgrad_small passes and two column sums to read back - ~0.45 ms per
// layer for a 593-parameter MLP.  Here a wave keeps the operands of its 64 items in LDS and contracts them over the items on the
// matrix cores (the items are the MFMA K index, 16 steps of v_mfma_f32_16x16x4_f32 per 64 items):
//     Y2^T [16 x items] x [h1 | 1] [items x (48 + 1)]     Y2 = (dz2 [8] | dout h2 [8])  ->  d lin3.2.weight, d lin3.2.bias, d lin3.4.weight
//     dz1^T [48 x items] x (x, 1) [items x 2]                                             ->  d lin3.0.weight[:, 0], d lin3.0.bias
// The accumulators live in registers over all the groups a wave walks; one partial block per wave at the end, reduced in a fixed
// order by k_lin3u_reduce (deterministic; d lin3.4.bias = sum of dout by a wave butterfly).
// LDS per wave: operand rows [64][52] (h1, then dz1 in the same place), Y2 [64][20], x [64]; the strides (52, 20) make the float4
// row writes of the 64 lanes conflict-free (208- / 80-byte row pitch = odd multiples of 16 bytes).
// =====================================================================================================
#define L3F_S1 52
#define L3F_S2 20
#define L3F_WAVE_FLOATS (64 * L3F_S1 + 64 * L3F_S2 + 64)
#define L3F_PART 1856            // per wave: acc W2 [4 tiles][64 lanes][4] | acc W0 [3][64][4] | sum dout (1, padded to 64)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_lin3u_bwd_fused(const float* __restrict__ p, const float* __restrict__ x,
                                                                 const float* __restrict__ dout, long long n, float* __restrict__ dx,
                                                                 float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float l3f_sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    float* OP = l3f_sm + (size_t)wave * L3F_WAVE_FLOATS;          // [64][L3F_S1]
    float* Y2 = OP + 64 * L3F_S1;                                  // [64][L3F_S2]
    float* XS = Y2 + 64 * L3F_S2;                                  // [64]
    const float* w0 = p;
    const float* b0 = p + 144;
    const float* w2 = p + 192;
    const float* b2 = p + 576;
    const float* w4 = p + 584;
    f4 aw2[4], aw0[3];
#pragma unroll
    for (int t = 0; t < 4; ++t) aw2[t] = f4zero();
#pragma unroll
    for (int t = 0; t < 3; ++t) aw0[t] = f4zero();
    float sg = 0.f;
    const long long ngroups = (n + 63) / 64;
    for (long long grp = (long long)blockIdx.x * WAVES + wave; grp < ngroups; grp += (long long)gridDim.x * WAVES) {
        const long long it = grp * 64 + lane;
        const bool ok = it < n;
        const float xv = ok ? x[it] : 0.f, gv = ok ? dout[it] : 0.f;     // items beyond n: dout = 0 -> every gradient operand is 0
        sg += gv;
        float z2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) z2[j] = b2[j];
        for (int k0 = 0; k0 < 48; k0 += 4) {
            f4 h;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = k0 + c;
                const float hk = silu1(w0[3 * k] * xv + b0[k]);
                h[c] = hk;
#pragma unroll
                for (int j = 0; j < 8; ++j) z2[j] += w2[j * 48 + k] * hk;
            }
            st_f4(OP + lane * L3F_S1 + k0, h);
        }
        float dz2[8];
        f4 ya = f4zero(), yb = f4zero(), yc = f4zero(), yd = f4zero();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float h2 = silu1(z2[j]);
            dz2[j] = gv * w4[j] * dsilu1(z2[j]);
            if (j < 4) { ya[j] = dz2[j]; yc[j] = gv * h2; } else { yb[j - 4] = dz2[j]; yd[j - 4] = gv * h2; }
        }
        st_f4(Y2 + lane * L3F_S2, ya); st_f4(Y2 + lane * L3F_S2 + 4, yb); st_f4(Y2 + lane * L3F_S2 + 8, yc); st_f4(Y2 + lane * L3F_S2 + 12, yd);
        XS[lane] = xv;
        __builtin_amdgcn_wave_barrier();                                  // same-wave LDS traffic: the queue is in order, keep the compiler in order too
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {                                    // Y2^T x [h1 | 1]
            const int r = 4 * s + g;
            const float a2 = Y2[r * L3F_S2 + i];
#pragma unroll
            for (int t = 0; t < 3; ++t) aw2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, OP[r * L3F_S1 + 16 * t + i], aw2[t], 0, 0, 0);
            aw2[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, i == 0 ? 1.0f : 0.f, aw2[3], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float dxv = 0.f;
        for (int k0 = 0; k0 < 48; k0 += 4) {                              // dz1 over the h1 rows
            f4 d;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = k0 + c;
                float dh = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) dh += w2[j * 48 + k] * dz2[j];
                const float dz = dh * dsilu1(w0[3 * k] * xv + b0[k]);
                d[c] = dz;
                dxv += w0[3 * k] * dz;
            }
            st_f4(OP + lane * L3F_S1 + k0, d);
        }
        if (ok) dx[it] = dxv;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {                                    // dz1^T x (x, 1)
            const int r = 4 * s + g;
            const float bx = i == 0 ? XS[r] : (i == 1 ? 1.0f : 0.f);
#pragma unroll
            for (int t = 0; t < 3; ++t) aw0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(OP[r * L3F_S1 + 16 * t + i], bx, aw0[t], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sg += __shfl_xor(sg, d, 64);
    float* out = partial + ((size_t)blockIdx.x * WAVES + wave) * L3F_PART;
#pragma unroll
    for (int t = 0; t < 4; ++t) st_f4(out + (t * 64 + lane) * 4, aw2[t]);
#pragma unroll
    for (int t = 0; t < 3; ++t) st_f4(out + 1024 + (t * 64 + lane) * 4, aw0[t]);
    if (lane == 0) out[1792] = sg;
}
// one wave per parameter-gradient entry: fixed-order sum over the per-wave partial blocks, ADDED to the destination
//   entries: [0, 384) lin3.2.weight [8][48] | [384, 392) lin3.2.bias | [392, 400) lin3.4.weight | [400, 448) lin3.0.weight[:, 0]
//            | [448, 496) lin3.0.bias | 496 lin3.4.bias
OARD_KERNEL __global__ __launch_bounds__(256) void k_lin3u_reduce(const float* __restrict__ partial, int n_waves, float* __restrict__ gw0,
                                                      float* __restrict__ gb0, float* __restrict__ gw2, float* __restrict__ gb2,
                                                      float* __restrict__ gw4, float* __restrict__ gb4) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= 497) return;
    int idx;                       // position inside a partial block
    float* dst;
    auto c_idx = [](int base, int t, int row, int col) { return base + (t * 64 + 16 * (row >> 2) + col) * 4 + (row & 3); };   // C layout
    if (e < 384) { const int row = e / 48, col = e % 48; idx = c_idx(0, col >> 4, row, col & 15); dst = gw2 ? gw2 + e : nullptr; }
    else if (e < 392) { idx = c_idx(0, 3, e - 384, 0); dst = gb2 ? gb2 + (e - 384) : nullptr; }
    else if (e < 400) { idx = c_idx(0, 3, 8 + (e - 392), 0); dst = gw4 ? gw4 + (e - 392) : nullptr; }
    else if (e < 448) { const int h = e - 400; idx = c_idx(1024, h >> 4, h & 15, 0); dst = gw0 ? gw0 + 3 * h : nullptr; }
    else if (e < 496) { const int h = e - 448; idx = c_idx(1024, h >> 4, h & 15, 1); dst = gb0 ? gb0 + h : nullptr; }
    else { idx = 1792; dst = gb4; }
    const float s = chunk_sum_wave(partial + idx, (size_t)L3F_PART, n_waves, lane);
    if (lane == 0 && dst != nullptr) *dst += s;
}
