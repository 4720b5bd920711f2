// oard_edge_bwd.h — backward pass of the two per-layer edge stages (training).
//
// Reference: what torch autograd derives for GCLMessage (leftnet.py:157-183) and for the edge half of
// EquiMessage (leftnet.py:245-249) when the loss of en_diffusion.py:56-248 is back-propagated
// (pl_trainer.py:327-347).  The reference has no hand-written backward; these kernels are the adjoints of
// k_gcl_edge_v1 / k_equi_edge_v1 (oard_edge_v1.h) on the same column engine:
//
//   * data gradients ("dx" kernels): the Linear layers run transposed, dX = W^T dY.  The transposed weights are
//     packed once per weight update into per-layer streams in consumption order (oard_pack_weights_bwd) and
//     streamed through LDS by LDS-DMA exactly like the forward streams; SiLU' is evaluated from the stored
//     pre-activations of the training-mode forward (GclTape, zd1, cd).
//   * weight gradients (k_wgrad): dW[o][i] = sum_rows dY[row][o] * X[row][i] is a GEMM whose contraction index is
//     the edge, i.e. the SLOW index of both row-major operands.  A wave loads 4 rows x 64 features of each operand
//     as one float4 per lane (fully coalesced 256-byte row segments) and uses component c of dY against component
//     c' of X in 16 MFMAs on 16 accumulators: accumulator (c, c') holds dW[4i + c][4j + c'], so no transpose is
//     ever needed.  Row chunks give per-workgroup partial sums, reduced by a second pass in a fixed order
//     (deterministic, no atomics).
#pragma once
#include "oard_edge_v1.h"

OARD_DEV float dsilu1(float x) {           // d/dx [x sigmoid(x)] = s (1 + x (1 - s))
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-x));
    return s * (1.0f + x * (1.0f - s));
}
// four values with the plain arithmetic as packed two-float operations (as silu4: VALU instructions take MFMA issue time)
OARD_DEV f4 dsilu4(f4 v) {
    const f2 a = {v.x, v.y}, b = {v.z, v.w};
    const f2 ta = a * -1.44269504088896340736f, tb = b * -1.44269504088896340736f;
    const f2 da = (f2){__builtin_amdgcn_exp2f(ta.x), __builtin_amdgcn_exp2f(ta.y)} + 1.0f;
    const f2 db = (f2){__builtin_amdgcn_exp2f(tb.x), __builtin_amdgcn_exp2f(tb.y)} + 1.0f;
    const f2 sa = {__builtin_amdgcn_rcpf(da.x), __builtin_amdgcn_rcpf(da.y)}, sb = {__builtin_amdgcn_rcpf(db.x), __builtin_amdgcn_rcpf(db.y)};
    const f2 ra = sa * (a * (1.0f - sa) + 1.0f), rb = sb * (b * (1.0f - sb) + 1.0f);       // s (1 + x (1 - s))
    return (f4){ra.x, ra.y, rb.x, rb.y};
}

// SiLU'(v) and SiLU(v) from one sigmoid
OARD_DEV f4 dsilu4_silu(f4 v, f4& sil) {
    const f2 a = {v.x, v.y}, b = {v.z, v.w};
    const f2 ta = a * -1.44269504088896340736f, tb = b * -1.44269504088896340736f;
    const f2 da = (f2){__builtin_amdgcn_exp2f(ta.x), __builtin_amdgcn_exp2f(ta.y)} + 1.0f;
    const f2 db = (f2){__builtin_amdgcn_exp2f(tb.x), __builtin_amdgcn_exp2f(tb.y)} + 1.0f;
    const f2 sa = {__builtin_amdgcn_rcpf(da.x), __builtin_amdgcn_rcpf(da.y)}, sb = {__builtin_amdgcn_rcpf(db.x), __builtin_amdgcn_rcpf(db.y)};
    const f2 ra = sa * (a * (1.0f - sa) + 1.0f), rb = sb * (b * (1.0f - sb) + 1.0f);
    const f2 ya = a * sa, yb = b * sb;
    sil = (f4){ya.x, ya.y, yb.x, yb.y};
    return (f4){ra.x, ra.y, rb.x, rb.y};
}
// sum over the 16 lanes of a DPP row (the 16 columns of one k-slice g); every lane of the row gets the total.  Four v_add_f32 with a
// DPP operand: quad xor 1, quad xor 2, mirror inside the half row, mirror of the row
#define OARD_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, false))
OARD_DEV float row16_sum(float x) {
    x += OARD_DPP(x, 0xB1);       // quad_perm [1, 0, 3, 2]
    x += OARD_DPP(x, 0x4E);       // quad_perm [2, 3, 0, 1]
    x += OARD_DPP(x, 0x141);      // row_half_mirror
    x += OARD_DPP(x, 0x140);      // row_mirror
    return x;
}

// =====================================================================================================
// GCLMessage edge part, backward.  Stream (chunks, all groups HT wide, no bias chunks):
//   T3 = WB groups x HT [W3^T, K-outer over the blocks of dz3];  T2 = HT groups x HT [W2^T tile t];
//   T1 = WB groups x HT [W1c^T tile tb].
// Per column (edge):   dz3 = G * SiLU'(z3);  dm = W3^T dz3 + dagg[src] / deg(src)
//                      m0 = SiLU(z2), gate = SiLU(a);  da = <dm, m0> SiLU'(a);  dz2 = (dm gate + watt da) SiLU'(z2)
//                      dz1 = (W2^T dz2) SiLU'(z1);   G_out = G + W1c^T dz1
// G (the gradient of the NEW edge state) is read from `dew` and the gradient of the OLD state is written back in place.
// =====================================================================================================
template <class D, int GP>
struct GclBwdStream {
    static constexpr int HT = D::HT, WB = D::WB, G = HT;
    static constexpr int SLAB = GP * G;
    static constexpr int NP3 = (WB + GP - 1) / GP, NP2 = (HT + GP - 1) / GP, NP1 = NP3, NPH = NP3 + NP2 + NP1;
    static constexpr int C3 = WB * G, C2 = HT * G, C1 = WB * G, CHUNKS = C3 + C2 + C1;
    static constexpr size_t LDS_BYTES = (size_t)2 * SLAB * 1024;
    // round 3: the forward's padding trims (oard_edge_v1.h), same conditions, matching flags in oard_pack_weights_bwd:
    //   compact K tail - the 13th block of dz2 / dz1 holds <= 4 real features: T2 / T1 tiles end with 1 MFMA instead of 4;
    //   4-row tiles    - the 13th output tile of W3^T (T3, K-outer) and of W2^T (T2) has <= 4 real rows: v_mfma_f32_4x4x1.
    static constexpr bool TAIL1 = (D::H % 16) >= 1 && (D::H % 16) <= 4 && HT >= 3 && (HT & 1);
    static constexpr bool ROWS4 = TAIL1;
};

struct GclBwdArgs {
    const float *z1, *z2, *att, *z3;   // tape of the training-mode forward (physical rows)
    const float* dagg;                 // [N][HP]   gradient of the per-node mean message (pads zero)
    const float* watt;                 // [HP]      attention weight, padded
    float* dew;                        // [E+1][WP] in: gradient of ew_{l+1}; out: gradient of ew_l
    float* dz3;                        // [E+1][WP] out
    float* mout;                       // [E+1][HP] out: gated message m (recomputed), physical row order
    float* dz2;                        // [E+1][HP] out
    float* da;                         // [E+1]     out
    float* dz1;                        // [E+1][HP] out
    float* gate_part;                  // [waves of the launch][HP] out, or nullptr: per-wave sums over its 16 edges of da * SiLU(z2) (the
                                       // att_mlp weight gradient; feature H holds the sum of da = its bias gradient) - a fixed-order
                                       // k_colsum_fin over the rows finishes them.  Round 4: replaces a [E][H] column-sum pass per layer
};

// HAS_S3 = false: inter-object rows of the last layer - the forward skipped S3 there (nothing reads their new
// state), so G == 0, z3 was never stored, dz3 is not produced, and the old-state gradient is just W1c^T dz1.
// (Three LDS slabs with the barrier inside the phase - k_gcl_edge_v1's RING = 3 - were measured here: 13.03 against 12.83 ms per
// step; this kernel waits for memory, not at the barrier.  Not kept.)
// timing-only ablations of an experiment build (results are garbage): -DOARD_GB_ABL=1 no vmcnt wait at the phase barriers of
// k_gcl_edge_bwd, 2 no global stores, 3 no edge-row loads, 4 neither loads nor stores
#ifndef OARD_GB_ABL
#define OARD_GB_ABL 0
#endif
OARD_DEV void gb_barrier() {
    if (OARD_GB_ABL == 1) __syncthreads(); else phase_barrier();
}
OARD_DEVVAR __device__ int g_gb_never = 0;            // always 0: "stores" of the no-store ablations stay in the code (nothing is dead) but never execute
OARD_DEV void gb_st(float* p, f4 v) { if ((OARD_GB_ABL != 2 && OARD_GB_ABL != 4) || g_gb_never) st_f4(p, v); }
OARD_DEV f4 gb_ld(const float* p) { return (OARD_GB_ABL == 3 || OARD_GB_ABL == 4) ? (f4){0.25f, -0.5f, 0.125f, 1.0f} : ld_f4(p); }
template <class D, int WAVES, int GP, bool HAS_S3>
__global__ __launch_bounds__(WAVES * 64, 2) void k_gcl_edge_bwd(TopoDev tp, const float* __restrict__ stream,
                                                                long long r0, long long r1, GclBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using S = GclBwdStream<D, GP>;
    constexpr int HT = D::HT, WB = D::WB, G = S::G;
    constexpr bool TAIL1 = S::TAIL1, ROWS4 = S::ROWS4;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    SlabPrefetch<WAVES, S::SLAB> pf;
    pf.wave = wave;
    pf.lane_off = (unsigned)lane * 16u;              // LDS-DMA source = uniform chunk address (SGPR pair) + this 32-bit lane offset
    auto pf_begin = [&](int p) {
        int start = 0, n = 0;
        if (p < S::NP3) { start = p * GP * G; n = min(GP, WB - p * GP) * G; }
        else if (p < S::NP3 + S::NP2) { const int q = p - S::NP3; start = S::C3 + q * GP * G; n = min(GP, HT - q * GP) * G; }
        else if (p < S::NPH) { const int q = p - S::NP3 - S::NP2; start = S::C3 + S::C2 + q * GP * G; n = min(GP, WB - q * GP) * G; }
        pf.begin(stream, smem, p, start, n);
    };
    auto hook = [&]() { pf.tick(); };
    auto SL = [&](int p) -> const float* { return smem + (size_t)(p & 1) * S::SLAB * 256 + lane * 4; };

    const long long c = r0 + ((long long)blockIdx.x * WAVES + wave) * 16 + (lane & 15);
    const size_t e = (size_t)(c < r1 ? c : tp.E);              // padding columns work on the spare row
    float* grow = a.dew + e * D::WP + 4 * g;
    const float* z3row = a.z3 + e * D::WP + 4 * g;
    float* dz3row = a.dz3 + e * D::WP + 4 * g;

    f4 dm[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) dm[t] = f4zero();
    f4 gn[GP], zn[GP];
#pragma unroll
    for (int gg = 0; gg < GP; ++gg) {
        gn[gg] = (HAS_S3 && gg < WB) ? gb_ld(grow + 16 * gg) : f4zero();
        zn[gg] = (HAS_S3 && gg < WB) ? gb_ld(z3row + 16 * gg) : f4zero();
    }
    int p = HAS_S3 ? 0 : S::NP3;
    pf_begin(p);
    pf.flush();

    // ---- T3: dm += W3^T . dz3   (K-outer); the dz3 blocks are stored one phase late ---------------------
    f4 xs[GP];
    f4 dmx = f4zero();                                   // ROWS4: second accumulator of the 4-row tile (unreduced k-slice sums)
    for (int p3 = 0; HAS_S3 && p3 < S::NP3; ++p3, ++p) {
        gb_barrier();
        if (p3 > 0) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) gb_st(dz3row + 16 * ((p3 - 1) * GP + gg), xs[gg]);
        }
        f4 x[GP];
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) x[gg] = gn[gg] * dsilu4(zn[gg]);
        pf_begin(p + 1);
        if (p3 + 1 < S::NP3) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) {
                const int b = (p3 + 1) * GP + gg;
                if (b < WB) { gn[gg] = gb_ld(grow + 16 * b); zn[gg] = gb_ld(z3row + 16 * b); }
            }
        }
#pragma unroll
        for (int gg = 0; gg < GP; ++gg)
            if (p3 * GP + gg < WB) chain_kouter<HT, ROWS4>(SL(p), gg * G, x[gg], dm, dmx, hook);
        pf.flush();
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) xs[gg] = x[gg];
    }
    if (HAS_S3) {
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int b = (S::NP3 - 1) * GP + gg;
            if (b < WB) gb_st(dz3row + 16 * b, xs[gg]);
        }
        if (ROWS4) {                                     // k-slice sums of the 4-row tile -> block layout (real rows in lanes g = 0)
            const f4 v = reduce_g(dm[HT - 1] + dmx);
            dm[HT - 1] = g == 0 ? v : f4zero();
        }
    }

    // ---- mean-aggregation adjoint, gate and SiLU of stage 2 (element-wise) --------------------------------
    f4 dz2[HT];
    {
        const int src = tp.row_src[e];
        const int smp = tp.node_sample[src];
        const int deg = tp.sample_ptr[smp + 1] - tp.sample_ptr[smp] - 1;
        const float inv = 1.0f / (float)max(deg, 1);            // util_funcs.py:40-44
        const float att = a.att[e];
        const float gate = silu1(att);
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            dm[t] += ld_blk(a.dagg, src, D::HP, t, lane) * inv;
            dz2[t] = ld_blk(a.z2, e, D::HP, t, lane);              // z2 for now
            const f4 m0 = silu4(dz2[t]);
            part += dm[t].x * m0.x + dm[t].y * m0.y + dm[t].z * m0.z + dm[t].w * m0.w;
            if ((OARD_GB_ABL != 2 && OARD_GB_ABL != 4) || g_gb_never) st_blk(a.mout, e, D::HP, t, lane, m0 * gate);
        }
        const float dav = col_reduce(part) * dsilu1(att);
        if (g == 0) a.da[e] = dav;
        const float dav_w = c < r1 ? dav : 0.f;                      // padding columns (spare row) do not count
        float* gp = a.gate_part != nullptr ? a.gate_part + ((size_t)blockIdx.x * WAVES + wave) * D::HP + 4 * g : nullptr;
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            const f4 z = dz2[t];
            f4 m0;
            const f4 ds = dsilu4_silu(z, m0);
            dz2[t] = (dm[t] * gate + ld_vec(a.watt, t, lane) * dav) * ds;
            if ((OARD_GB_ABL != 2 && OARD_GB_ABL != 4) || g_gb_never) st_blk(a.dz2, e, D::HP, t, lane, dz2[t]);
            if (gp != nullptr) {                                     // wave-uniform
                f4 w = m0 * dav_w;
                if (16 * t + 4 * g == D::H) w.x = dav_w;             // the first pad feature carries sum(da): att_mlp's bias gradient
                w = (f4){row16_sum(w.x), row16_sum(w.y), row16_sum(w.z), row16_sum(w.w)};
                if ((lane & 15) == 0) st_f4(gp + 16 * t, w);
            }
        }
    }

    // ---- T2: dz1 = (W2^T dz2) * SiLU'(z1), one output tile per group ---------------------------------------
    f4 dz1[HT];
    f4 z1n[GP], on[GP];
    const float dz2_tail = TAIL1 ? tail_compact(dz2[HT - 1], lane) : 0.f;
#pragma unroll
    for (int gg = 0; gg < GP; ++gg) z1n[gg] = gg < HT ? ld_blk(a.z1, e, D::HP, gg, lane) : f4zero();
#pragma unroll
    for (int p2 = 0; p2 < S::NP2; ++p2, ++p) {
        gb_barrier();
        pf_begin(p + 1);
        f4 zc[GP];
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) zc[gg] = z1n[gg];
        if (p2 + 1 < S::NP2) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) {
                const int t = (p2 + 1) * GP + gg;
                if (t < HT) z1n[gg] = ld_blk(a.z1, e, D::HP, t, lane);
            }
        } else {                                       // prefetch the incoming-gradient tiles of T1's first phase
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) on[gg] = (HAS_S3 && gg < WB) ? gb_ld(grow + 16 * gg) : f4zero();
        }
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int t = p2 * GP + gg;                // compile-time after unrolling
            if (t < HT) {
                if (ROWS4 && t == HT - 1) {             // 13th tile: 4 real rows on 4x4x1 MFMAs (rows4 packing, K tail not compacted)
                    const f4 v = reduce_g(chain_tile4<HT>(SL(p), gg * G, dz2, f4zero(), hook));
                    dz1[t] = (g == 0 ? v : f4zero()) * dsilu4(zc[gg]);
                } else {
                    dz1[t] = chain_tile<HT, TAIL1>(SL(p), gg * G, dz2, f4zero(), dz2_tail, hook) * dsilu4(zc[gg]);
                }
                if ((OARD_GB_ABL != 2 && OARD_GB_ABL != 4) || g_gb_never) st_blk(a.dz1, e, D::HP, t, lane, dz1[t]);
            }
        }
        pf.flush();
    }

    // ---- T1: G_out = G + W1c^T dz1, one output tile per group; stores one phase late --------------------------
    f4 pend[GP];
    const float dz1_tail = TAIL1 ? tail_compact(dz1[HT - 1], lane) : 0.f;
    for (int p1 = 0; p1 < S::NP1; ++p1, ++p) {
        gb_barrier();
        if (p1 > 0) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) gb_st(grow + 16 * ((p1 - 1) * GP + gg), pend[gg]);
        }
        f4 o[GP];
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) o[gg] = on[gg];
        pf_begin(p + 1);
        if (HAS_S3 && p1 + 1 < S::NP1) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) {
                const int t = (p1 + 1) * GP + gg;
                if (t < WB) on[gg] = gb_ld(grow + 16 * t);
            }
        }
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int t = p1 * GP + gg;
            if (t < WB) pend[gg] = chain_tile<HT, TAIL1>(SL(p), gg * G, dz1, o[gg], dz1_tail, hook);
        }
        pf.flush();
    }
#pragma unroll
    for (int gg = 0; gg < GP; ++gg) {
        const int t = (S::NP1 - 1) * GP + gg;
        if (t < WB) gb_st(grow + 16 * t, pend[gg]);
    }
}

// =====================================================================================================
// EquiMessage edge part, backward, inner edges.  Stream: U2 = 3*HT groups x D1T [dir_proj.2^T, K-outer over the
// blocks of dcd in their stored order (third-major)];  U1 = WB groups x D1T [dir_proj.0^T tile tb].
//   dd1 = dir_proj.2^T dcd;  dzd1 = dd1 * SiLU'(zd1);  dew[a] += dir_proj.0^T dzd1
// (dcd = dq * cr, the rbf_proj factor and its gradient are element-wise / a plain [A,R] GEMM on the caller's side)
// =====================================================================================================
template <class D>
struct EquiBwdStream {
    static constexpr int WB = D::WB, D1T = D::D1T, HT = D::HT, G = D1T, NG2 = 3 * HT;
    static constexpr int SLAB = G, NPH = NG2 + WB, C2 = NG2 * G, CHUNKS = C2 + WB * G;
    static constexpr size_t LDS_BYTES = (size_t)2 * SLAB * 1024;
};

template <class D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_equi_edge_bwd(TopoDev tp, const float* __restrict__ stream,
                                                              const float* __restrict__ dcd, const float* __restrict__ zd1,
                                                              float* dew, float* __restrict__ dzd1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using S = EquiBwdStream<D>;
    constexpr int WB = D::WB, D1T = D::D1T, G = S::G;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    SlabPrefetch<WAVES, S::SLAB> pf;
    pf.wave = wave;
    pf.lane_off = (unsigned)lane * 16u;              // LDS-DMA source = uniform chunk address (SGPR pair) + this 32-bit lane offset
    auto pf_begin = [&](int p) { pf.begin(stream, smem, p, p * G, p >= S::NPH ? 0 : G); };
    auto hook = [&]() { pf.tick(); };
    auto SL = [&](int p) -> const float* { return smem + (size_t)(p & 1) * S::SLAB * 256 + lane * 4; };

    const long long c = ((long long)blockIdx.x * WAVES + wave) * 16 + (lane & 15);
    const size_t ai = (size_t)(c < tp.A ? c : tp.A);          // padding columns use the spare entry A
    float* grow = dew + (c < tp.A ? ai : (size_t)tp.E) * D::WP + 4 * g;      // inner entry a == physical row a
    const float* crow = dcd + ai * (size_t)(3 * D::HP) + 4 * g;

    f4 d1[D1T];
#pragma unroll
    for (int t = 0; t < D1T; ++t) d1[t] = f4zero();
    f4 xn = ld_f4(crow);
    pf_begin(0);
    pf.flush();
    int p = 0;
    for (int j = 0; j < S::NG2; ++j, ++p) {
        phase_barrier();
        const f4 x = xn;
        pf_begin(p + 1);
        if (j + 1 < S::NG2) xn = ld_f4(crow + 16 * (j + 1));
        chain_kouter<D1T>(SL(p), 0, x, d1, hook);
        pf.flush();
    }
#pragma unroll
    for (int t = 0; t < D1T; ++t) {
        d1[t] = d1[t] * dsilu4(ld_blk(zd1, ai, D::D1P, t, lane));
        st_blk(dzd1, ai, D::D1P, t, lane, d1[t]);
    }
    f4 on = ld_f4(grow), pend = f4zero();
    for (int tb = 0; tb < WB; ++tb, ++p) {
        phase_barrier();
        if (tb > 0) st_f4(grow + 16 * (tb - 1), pend);
        const f4 o = on;
        pf_begin(p + 1);
        if (tb + 1 < WB) on = ld_f4(grow + 16 * (tb + 1));
        pend = chain_tile<D1T>(SL(p), 0, d1, o, hook);
        pf.flush();
    }
    st_f4(grow + 16 * (WB - 1), pend);
}

// Workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD with its own L2.  Kernels with one workgroup per node that read rows
// shared by the nodes of one group (both end points of an edge read the edge's row) want the nodes of a group on ONE XCD: the
// workgroups of an XCD (b, b + 8, b + 16, ...) get runs of 32 consecutive node indices.  A bijection on [0, nblk); the last
// nblk % 256 workgroups keep their index.
#ifndef OARD_XCD_RUNS
#define OARD_XCD_RUNS 1
#endif
OARD_DEV int xcd_run32(int b, int nblk) {
    if (b >= (nblk & ~255)) return b;
    const int x = b & 7, s = b >> 3;
    return (((s >> 5) * 8 + x) << 5) + (s & 31);
}

// =====================================================================================================
// dP[n] = sum of dz1 over the edges whose SOURCE is n, dQ[n] = over the edges whose TARGET is n (the node terms
// P[src] + Q[tgt] of edge_mlp.0 were hoisted to the nodes in the forward).  One 256-thread block per node: wave w takes
// the list entries k = w (mod 4) of both lists, one float4 of the row per lane, four rows in flight per wave; the four
// partial sums are added in wave order (fixed summation order).  Round 4: was one wave per node walking both lists
// (index load -> row load, 2 (ns - 1) dependent round trips): 118 -> see profiles us per layer at B = 64.
// =====================================================================================================
OARD_KERNEL __global__ __launch_bounds__(256) void k_edge_node_sums(TopoDev tp, const float* __restrict__ dz1, int HP,
                                                        float* __restrict__ dP, float* __restrict__ dQ) {
    __shared__ f4 part[3][2][64];
    const int n = OARD_XCD_RUNS ? xcd_run32(blockIdx.x, gridDim.x) : (int)blockIdx.x, lane = threadIdx.x & 63, f = lane * 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool on = f < HP;
    const int smp = tp.node_sample[n], s0 = tp.sample_ptr[smp], ns = tp.sample_ptr[smp + 1] - s0, self = n - s0;
    f4 ap = f4zero(), aq = f4zero();
    const int e0 = tp.edge_ptr[n];
    const float* col = dz1 + (on ? f : 0);
    // list position j (0 .. ns - 2) of the target list is sample member k = j + (j >= self); n's own position in that member's list
    auto qrow = [&](int j) { const int k = j + (j >= self ? 1 : 0); return tp.edge_row[tp.edge_ptr[s0 + k] + self - (self > k ? 1 : 0)]; };
    int j = wave;
    for (; j + 12 < ns - 1; j += 16) {
        const int r0 = tp.edge_row[e0 + j], r1 = tp.edge_row[e0 + j + 4], r2 = tp.edge_row[e0 + j + 8], r3 = tp.edge_row[e0 + j + 12];
        const int q0 = qrow(j), q1 = qrow(j + 4), q2 = qrow(j + 8), q3 = qrow(j + 12);
        const f4 a0 = ld_f4(col + (size_t)r0 * HP), a1 = ld_f4(col + (size_t)r1 * HP), a2 = ld_f4(col + (size_t)r2 * HP), a3 = ld_f4(col + (size_t)r3 * HP);
        const f4 b0 = ld_f4(col + (size_t)q0 * HP), b1 = ld_f4(col + (size_t)q1 * HP), b2 = ld_f4(col + (size_t)q2 * HP), b3 = ld_f4(col + (size_t)q3 * HP);
        ap += (a0 + a1) + (a2 + a3);
        aq += (b0 + b1) + (b2 + b3);
    }
    for (; j < ns - 1; j += 4) {
        ap += ld_f4(col + (size_t)tp.edge_row[e0 + j] * HP);
        aq += ld_f4(col + (size_t)qrow(j) * HP);
    }
    if (wave > 0) { part[wave - 1][0][lane] = ap; part[wave - 1][1][lane] = aq; }
    __syncthreads();
    if (wave == 0 && on) {
#pragma unroll
        for (int w = 0; w < 3; ++w) { ap += part[w][0][lane]; aq += part[w][1][lane]; }
        st_f4(dP + (size_t)n * HP + f, ap);
        st_f4(dQ + (size_t)n * HP + f, aq);
    }
}

// =====================================================================================================
// weight gradient GEMM:  partial[chunk][pf][qf] = sum_{rows of the chunk} P[row][pf] * act(Q[row][qf])
// Both operands are row-major [rows][features]: the contraction index is their SLOW index.  The "wide" operand P is read as
// one float4 per lane (4 rows x 64 features, coalesced 256-byte segments): component c of lane (g, i) is P[row g][64 pb + 4 i + c],
// i.e. the A operand of an MFMA whose output row i stands for feature 4 i + c - four accumulators per P block, no transpose.
// The "narrow" operand Q is read as 16-feature tiles, one dword per lane (lane (g, j): Q[row g][16 t + j] = the B operand), so
// widths that are multiples of 16 but not of 64 (H = 196 -> 208 = 13 tiles, 3H -> 592 = 37 tiles) cost no padding.
// A wave owns one P block (64 features) x NT Q tiles: 4 NT accumulators, 4 NT MFMAs per 4 rows from 1 + NT loads.
// psum / qsum (optional): per-chunk column sums of P / Q (the bias gradient is the column sum of dY).
// =====================================================================================================
#define OARD_WG_PD 2                 // prefetch distance in 4-row steps (3 spills at the 256-register bound)
// Addressing (round 2): every load is "wave-uniform base (SGPR pair, advanced by scalar adds) + constant 32-bit lane offset", so the
// inner loop has no address arithmetic, no predicates and no zero-filling on the VALU: the first version spent 105 VALU instructions
// (62 v_mov, 22 64-bit adds) and 23 exec branches per 56 MFMAs, and VALU instructions take MFMA issue time on this hardware
// (tools/micro/mfma_valu.hip).  Lanes of a partial P block read the row's last valid float4 instead, invalid Q tiles of the last
// group re-read the group's first tile: their accumulators belong to padding outputs that no reduce pass reads.  Rows beyond the
// chunk are handled by a short masked epilogue.
template <bool QSILU, int NT>
OARD_DEV void wgrad_wave_body(const float* __restrict__ P, int ldP, int ncP, const float* __restrict__ Q, int ldQ,
                              int ncQ, long long r0, long long r1, long long rows_per_chunk, int nPB, int nQG,
                              float* __restrict__ partial, float* __restrict__ psum, float* __restrict__ qsum, const unsigned bid) {
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // grid.x = chunk * gy + task group (4 consecutive (P block, Q group) tasks per workgroup, Q group fastest): the workgroups
    // that stream the same row chunk are dispatched back to back.  Measured alternatives (profiles/round2_wgrad_notes.txt):
    // XCD-aware placement of a chunk's workgroups -15 %, 8-wave workgroups owning 4 P blocks x 2 Q groups -23 %.
    const int gy = (nPB * nQG + 3) / 4;
    const int chunk = bid / gy;
    const int task = (bid % gy) * 4 + wave;
    const long long rb = r0 + (long long)chunk * rows_per_chunk;
    if (task >= nPB * nQG || rb >= r1) return;
    const int pb = task / nQG, qg = task - pb * nQG;
    const int PP = nPB * 64, QP = nQG * NT * 16;
    const long long re = rb + rows_per_chunk < r1 ? rb + rows_per_chunk : r1;
    const int cp = 64 * pb + 4 * i;
    const unsigned voffP = (unsigned)(g * ldP + (cp < ncP ? cp : ncP - 4)) * 4u;      // constant per lane (bytes)
    const unsigned voffQ = (unsigned)(g * ldQ + 16 * qg * NT + i) * 4u;
    int qtile[NT];                                                                    // wave-uniform: tile u, or 0 if it lies beyond ncQ
#pragma unroll
    for (int u = 0; u < NT; ++u) qtile[u] = 16 * (qg * NT + u) < ncQ ? 64 * u : 0;
    const char* bp = reinterpret_cast<const char*>(P + (size_t)rb * ldP);             // wave-uniform bases, advanced per step
    const char* bq = reinterpret_cast<const char*>(Q + (size_t)rb * ldQ);
    const size_t stepP = (size_t)16 * ldP, stepQ = (size_t)16 * ldQ;                 // 4 rows, bytes

    f4 acc[4][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[c][u] = f4zero();
    f4 ps = f4zero();
    float qs[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) qs[u] = 0.f;
    const bool sums = (psum != nullptr && qg == 0) || (qsum != nullptr && pb == 0);        // wave-uniform
    const long long nrows = re - rb, nfull = nrows / 4;        // nfull: steps whose 4 rows are all inside the chunk
    f4 ra[OARD_WG_PD];
    float rq[OARD_WG_PD][NT];
    auto load = [&](int slot) {                               // unconditional: the caller guarantees that the 4 rows exist
        ra[slot] = *reinterpret_cast<const f4*>(bp + voffP);
#pragma unroll
        for (int u = 0; u < NT; ++u) rq[slot][u] = *reinterpret_cast<const float*>(bq + qtile[u] + voffQ);
        bp += stepP; bq += stepQ;
    };
    auto mma_step = [&](const f4 a, const float (&b)[NT], bool with_sums) {
        if (with_sums) {
            ps += a;
#pragma unroll
            for (int u = 0; u < NT; ++u) qs[u] += b[u];
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[u], acc[0][u], 0, 0, 0);
            acc[1][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[u], acc[1][u], 0, 0, 0);
            acc[2][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[u], acc[2][u], 0, 0, 0);
            acc[3][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[u], acc[3][u], 0, 0, 0);
        }
    };
    long long st = 0;                                         // steps consumed
    if (nfull >= OARD_WG_PD) {
#pragma unroll
        for (int u = 0; u < OARD_WG_PD; ++u) load(u);
        // main loop: the step loaded in place of a consumed one (OARD_WG_PD steps ahead) is a full step as well
        for (; st + 2 * OARD_WG_PD <= nfull; st += OARD_WG_PD) {
#pragma unroll
            for (int sl = 0; sl < OARD_WG_PD; ++sl) {
                const f4 a = ra[sl];
                float b[NT];
#pragma unroll
                for (int u = 0; u < NT; ++u) b[u] = QSILU ? silu1(rq[sl][u]) : rq[sl][u];
                load(sl);
                mma_step(a, b, sums);
            }
        }
#pragma unroll
        for (int sl = 0; sl < OARD_WG_PD; ++sl) {              // drain the loaded steps
            float b[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) b[u] = QSILU ? silu1(rq[sl][u]) : rq[sl][u];
            mma_step(ra[sl], b, sums);
        }
        st += OARD_WG_PD;
    }
    // what is left (fewer than 2 OARD_WG_PD full steps and the partial one): rows beyond the chunk read row re - 1 and count as zero
    for (; st * 4 < nrows; ++st) {
        const long long row = rb + 4 * st + g;
        const bool v = row < re;
        const size_t rr = (size_t)((v ? row : re - 1) - rb - g);        // row offset relative to the lane's row rb + g
        f4 a = *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(P + (size_t)rb * ldP) + rr * ldP * 4 + voffP);
        float b[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Q + (size_t)rb * ldQ) + rr * ldQ * 4 + qtile[u] + voffQ);
            b[u] = v ? (QSILU ? silu1(x) : x) : 0.f;
        }
        if (!v) a = f4zero();
        mma_step(a, b, sums);
    }
    // accumulator (c, u), component q of lane (g, j):  out[64 pb + 4 (4g + q) + c][16 (qg NT + u) + j]
    float* out = partial + (size_t)chunk * PP * QP;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                out[(size_t)(64 * pb + 4 * (4 * g + q) + c) * QP + 16 * (qg * NT + u) + i] = acc[c][u][q];
    if (psum != nullptr && qg == 0) {
        ps.x = col_reduce(ps.x); ps.y = col_reduce(ps.y); ps.z = col_reduce(ps.z); ps.w = col_reduce(ps.w);
        if (g == 0) st_f4(psum + (size_t)chunk * PP + 64 * pb + 4 * i, ps);
    }
    if (qsum != nullptr && pb == 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float v = col_reduce(qs[u]);
            if (g == 0) qsum[(size_t)chunk * QP + 16 * (qg * NT + u) + i] = v;
        }
    }
}

template <bool QSILU, int NT>
__global__ __launch_bounds__(256, 2) void k_wgrad(const float* __restrict__ P, int ldP, int ncP, const float* __restrict__ Q, int ldQ,
                                                  int ncQ, long long r0, long long r1, long long rows_per_chunk, int nPB, int nQG,
                                                  float* __restrict__ partial, float* __restrict__ psum, float* __restrict__ qsum) {
    wgrad_wave_body<QSILU, NT>(P, ldP, ncP, Q, ldQ, ncQ, r0, r1, rows_per_chunk, nPB, nQG, partial, psum, qsum, blockIdx.x);
}

// ---- grouped launches (round 4) ---------------------------------------------------------------------------------------------------------
// The node-level Linear layers of a layer's reverse sweep (x_proj, node_mlp, xvec_proj, vec_proj, the node halves of edge_mlp.0,
// pos_expansion: 12 products over N or 3 N rows) are 0.3 ... 1 GFLOP each: as 12 + 12 + 8 launches of k_wgrad / k_wgrad_reduce /
// k_bgrad_reduce they were latency-bound (86 launches of 34 us per training step).  The stages now QUEUE them (oard_train_stages.h) and
// a layer ends with ONE launch of each kind over a job table: same kernels, same chunking, same summation order - bit-identical
// results.
struct WgqJob {
    const float* P; const float* Q; float* partial; float* bsum; float* dW; float* db;
    long long rows, rpc;
    int ldP, ncP, ldQ, ncQ, nPB, nQG, n_chunks, transposed;
    int o_len, o_pad, MO, i_len, i_pad, MI, ldW, acc, PP, QP;
    unsigned blk0, rblk0, rblk_w, rblk_b;      // first block in the GEMM launch; first block in the reduce launch, its dW / db block counts
};
OARD_DEV int wgq_find(const WgqJob* __restrict__ jobs, int n_jobs, unsigned bid, bool reduce) {
    int cur = 0;
    for (int j = 1; j < n_jobs; ++j)
        if (bid >= (reduce ? jobs[j].rblk0 : jobs[j].blk0)) cur = j;
    return cur;
}
template <int NT>
__global__ __launch_bounds__(256, 2) void k_wgrad_q(const WgqJob* __restrict__ jobs, int n_jobs) {
    const WgqJob& J = jobs[wgq_find(jobs, n_jobs, blockIdx.x, false)];
    float* psum = (J.bsum != nullptr && !J.transposed) ? J.bsum : nullptr;
    float* qsum = (J.bsum != nullptr && J.transposed) ? J.bsum : nullptr;
    wgrad_wave_body<false, NT>(J.P, J.ldP, J.ncP, J.Q, J.ldQ, J.ncQ, 0LL, J.rows, J.rpc, J.nPB, J.nQG, J.partial, psum, qsum,
                               blockIdx.x - J.blk0);
}

// =====================================================================================================
// k_wgrad_lds (round 3): the same contraction with a row panel SHARED through LDS.  k_wgrad's per-wave tile (64 x 16 NT outputs)
// has an arithmetic intensity of ~20 FLOP per operand byte, i.e. it sits on the HBM ridge and depends on L2 hits for the operand
// re-reads of the other tasks (measured round 2: FETCH x 2 = 68.6 GB per training step for 23 GB of unique operands, 55 % MFMA-busy).
// Here a workgroup of 8 waves owns PBW P blocks x QGW Q groups (PBW QGW = 8 tasks, 256 x 16 NT QGW outputs) of one row chunk and
// stages GROUPS of 32 rows of exactly those columns in LDS, double-buffered: every operand element is fetched once per workgroup
// (~60 FLOP per byte), as coalesced 1-KiB row segments, SiLU-on-load is applied once per element instead of once per task, and the
// next group's loads are in flight during the 8 steps (8 x 4 NT MFMAs per wave) of the current one.  One barrier per group.
//   P panel [32][256] floats (a wave's A operand = one conflict-free ds_read_b128);  Q panel [32][QLD = 16 NT QGW + 16] floats: the
//   +16 shifts consecutive rows by 16 banks, so the four rows a ds_read_b32 of the B operand touches hit 64 distinct banks.
// Same partial / psum / qsum layout as k_wgrad (the reduce passes are shared); rows beyond the chunk count as zero.
// =====================================================================================================
#define WGL_ROWS 32
template <bool QSILU, int NT, int PBW, int QGW>
__global__ __launch_bounds__(512, 1) void k_wgrad_lds(const float* __restrict__ P, int ldP, int ncP, const float* __restrict__ Q, int ldQ,
                                                       int ncQ, long long r0, long long r1, long long rows_per_chunk, int nPB, int nQG,
                                                       float* __restrict__ partial, float* __restrict__ psum, float* __restrict__ qsum) {
    static_assert(PBW * QGW == 8, "one task per wave");
    constexpr int PW = 64 * PBW, QW = 16 * NT * QGW, QLD = QW + 16;
    constexpr int PF4 = WGL_ROWS * PW / 4 / 512, QF4 = (WGL_ROWS * QW / 4 + 511) / 512;      // float4 loads per thread and group
    extern __shared__ __attribute__((aligned(16))) float wgl_sm[];
    float* sP = wgl_sm;                                  // [2][WGL_ROWS][PW]
    float* sQ = wgl_sm + 2 * WGL_ROWS * PW;              // [2][WGL_ROWS][QLD]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nPT = (nPB + PBW - 1) / PBW, nQT_ = (nQG + QGW - 1) / QGW;
    const int chunk = blockIdx.x / (nPT * nQT_), tile = blockIdx.x % (nPT * nQT_);
    const int pt = tile / nQT_, qt = tile % nQT_;
    const int pb = pt * PBW + wave / QGW, qg = qt * QGW + wave % QGW;
    const bool active = pb < nPB && qg < nQG;
    const long long rb = r0 + (long long)chunk * rows_per_chunk;
    const long long re = rb + rows_per_chunk < r1 ? rb + rows_per_chunk : r1;
    if (rb >= r1) return;                                 // whole workgroup (uniform)
    const long long nrows = re - rb;
    const int ngroups = (int)((nrows + WGL_ROWS - 1) / WGL_ROWS);
    const int p_col0 = 64 * PBW * pt, q_col0 = 16 * NT * QGW * qt;

    f4 stP[PF4], stQ[QF4];
    auto fetch = [&](int grp) {                           // global -> registers (clamped addresses, invalid rows / columns become 0)
        const long long gr0 = rb + (long long)grp * WGL_ROWS;
#pragma unroll
        for (int k = 0; k < PF4; ++k) {
            const int idx = tid + 512 * k, row = idx / (PW / 4), c = p_col0 + 4 * (idx % (PW / 4));
            const long long r = gr0 + row;
            const bool ok = r < re && c < ncP;
            const f4 v = *reinterpret_cast<const f4*>(P + (size_t)(r < re ? r : re - 1) * ldP + (c < ncP ? c : ncP - 4));
            stP[k] = ok ? v : f4zero();
        }
#pragma unroll
        for (int k = 0; k < QF4; ++k) {
            const int idx = tid + 512 * k, row = idx / (QW / 4), c = q_col0 + 4 * (idx % (QW / 4));
            const long long r = gr0 + row;
            const bool ok = row < WGL_ROWS && r < re && c < ncQ;
            f4 v = *reinterpret_cast<const f4*>(Q + (size_t)((row < WGL_ROWS && r < re) ? r : re - 1) * ldQ + (c < ncQ ? c : ncQ - 4));
            if (QSILU) v = silu4(v);
            stQ[k] = ok ? v : f4zero();
        }
    };
    auto stash = [&](int buf) {                           // registers -> LDS
#pragma unroll
        for (int k = 0; k < PF4; ++k) {
            const int idx = tid + 512 * k;
            *reinterpret_cast<f4*>(sP + (size_t)buf * WGL_ROWS * PW + 4 * idx) = stP[k];
        }
#pragma unroll
        for (int k = 0; k < QF4; ++k) {
            const int idx = tid + 512 * k, row = idx / (QW / 4), c4 = idx % (QW / 4);
            if (row < WGL_ROWS) *reinterpret_cast<f4*>(sQ + ((size_t)buf * WGL_ROWS + row) * QLD + 4 * c4) = stQ[k];
        }
    };

    f4 acc[4][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[c][u] = f4zero();
    f4 ps = f4zero();
    float qs[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) qs[u] = 0.f;
    const bool sums = active && ((psum != nullptr && qg == 0) || (qsum != nullptr && pb == 0));
    const int pl = wave / QGW, ql = wave % QGW;           // this wave's block / group inside the panel

    fetch(0);
    stash(0);
    __syncthreads();
    for (int grp = 0; grp < ngroups; ++grp) {
        const int buf = grp & 1;
        if (grp + 1 < ngroups) fetch(grp + 1);
        if (active) {
            const float* bp = sP + (size_t)buf * WGL_ROWS * PW + 64 * pl + 4 * i;
            const float* bq = sQ + (size_t)buf * WGL_ROWS * QLD + 16 * NT * ql + i;
            bp += (size_t)g * PW; bq += (size_t)g * QLD;
#ifndef OARD_WGL_PIPE
#ifndef OARD_WGL_UNROLL
#define OARD_WGL_UNROLL 2
#endif
#pragma unroll OARD_WGL_UNROLL
            for (int stp = 0; stp < WGL_ROWS / 4; ++stp, bp += 4 * PW, bq += 4 * QLD) {
                const f4 a = *reinterpret_cast<const f4*>(bp);
                float b[NT];
#pragma unroll
                for (int u = 0; u < NT; ++u) b[u] = bq[16 * u];
                if (sums) {
                    ps += a;
#pragma unroll
                    for (int u = 0; u < NT; ++u) qs[u] += b[u];
                }
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[u], acc[0][u], 0, 0, 0);
                    acc[1][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[u], acc[1][u], 0, 0, 0);
                    acc[2][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[u], acc[2][u], 0, 0, 0);
                    acc[3][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[u], acc[3][u], 0, 0, 0);
                }
            }
#else       // the LDS operands of step s + 1 are requested before the MFMAs of step s are issued
            f4 a = *reinterpret_cast<const f4*>(bp);
            float b[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) b[u] = bq[16 * u];
#pragma unroll
            for (int stp = 0; stp < WGL_ROWS / 4; ++stp) {
                f4 an = a;
                float bn[NT];
#pragma unroll
                for (int u = 0; u < NT; ++u) bn[u] = b[u];
                if (stp + 1 < WGL_ROWS / 4) {
                    an = *reinterpret_cast<const f4*>(bp + (size_t)(stp + 1) * 4 * PW);
#pragma unroll
                    for (int u = 0; u < NT; ++u) bn[u] = bq[(size_t)(stp + 1) * 4 * QLD + 16 * u];
                }
                if (sums) {
                    ps += a;
#pragma unroll
                    for (int u = 0; u < NT; ++u) qs[u] += b[u];
                }
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[u], acc[0][u], 0, 0, 0);
                    acc[1][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[u], acc[1][u], 0, 0, 0);
                    acc[2][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[u], acc[2][u], 0, 0, 0);
                    acc[3][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[u], acc[3][u], 0, 0, 0);
                }
                a = an;
#pragma unroll
                for (int u = 0; u < NT; ++u) b[u] = bn[u];
            }
#endif
        }
        if (grp + 1 < ngroups) stash(buf ^ 1);
        __syncthreads();
    }
    if (!active) return;
    const int PP = nPB * 64, QP = nQG * NT * 16;
    float* out = partial + (size_t)chunk * PP * QP;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                out[(size_t)(64 * pb + 4 * (4 * g + q) + c) * QP + 16 * (qg * NT + u) + i] = acc[c][u][q];
    if (psum != nullptr && qg == 0) {
        ps.x = col_reduce(ps.x); ps.y = col_reduce(ps.y); ps.z = col_reduce(ps.z); ps.w = col_reduce(ps.w);
        if (g == 0) st_f4(psum + (size_t)chunk * PP + 64 * pb + 4 * i, ps);
    }
    if (qsum != nullptr && pb == 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float v = col_reduce(qs[u]);
            if (g == 0) qsum[(size_t)chunk * QP + 16 * (qg * NT + u) + i] = v;
        }
    }
}

// Small outputs (MO * (MI + 1) <= 1024, e.g. the 8 x 48 / 48 x 2 / 1 x 9 layers of the frame-scalar MLPs over ~1e6 rows): the MFMA
// kernel above would run 128 x 256 tiles that are almost all padding and is latency-bound there.  Here a workgroup stages 64
// rows of both operands in LDS and every thread owns up to four outputs (o, i); column i == MI is the bias (X = 1).
// partial layout: [chunk][MO][MI + 1].
OARD_KERNEL __global__ __launch_bounds__(256) void k_wgrad_small(const float* __restrict__ dY, int ldY, int MO, const float* __restrict__ X, int ldX,
                                                     int MI, int x_silu, long long rows, long long rows_per_chunk,
                                                     float* __restrict__ partial) {
    __shared__ float sy[64 * 65], sx[64 * 65];
    const int tid = threadIdx.x, nout = MO * (MI + 1);
    const long long rb = (long long)blockIdx.x * rows_per_chunk;
    const long long re = rb + rows_per_chunk < rows ? rb + rows_per_chunk : rows;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int oo[4], ii[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int idx = min(tid + 256 * q, nout - 1); oo[q] = idx / (MI + 1); ii[q] = idx % (MI + 1); }
    for (long long r0 = rb; r0 < re; r0 += 64) {
        const int nr = (int)(re - r0 < 64 ? re - r0 : 64);
        __syncthreads();
        for (int k = tid; k < 64 * MO; k += 256) { const int r = k / MO, c = k % MO; sy[r * 65 + c] = r < nr ? dY[(size_t)(r0 + r) * ldY + c] : 0.f; }
        for (int k = tid; k < 64 * (MI + 1); k += 256) {
            const int r = k / (MI + 1), c = k % (MI + 1);
            float v = c == MI ? 1.0f : (r < nr ? X[(size_t)(r0 + r) * ldX + c] : 0.f);
            if (x_silu && c < MI) v = silu1(v);
            sx[r * 65 + c] = r < nr ? v : 0.f;
        }
        __syncthreads();
        for (int r = 0; r < 64; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += sy[r * 65 + oo[q]] * sx[r * 65 + ii[q]];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (tid + 256 * q < nout) partial[(size_t)blockIdx.x * nout + tid + 256 * q] = acc[q];
}
// Fixed-order sum of one value over the chunks by ONE WAVE: lane l adds chunks l, l + 64, ... in ascending order, then the 64
// lane sums are combined by a fixed butterfly - deterministic, and ~n_chunks / 64 dependent loads per thread instead of n_chunks
// (a single thread walking 1024 partials took 200 us per call: 18 + 24 such calls per training step).
// Accumulated in float64 (round 5): these are sums of up to ~19 000 partials per output, and for cancellation-heavy outputs - the att_mlp
// bias gradient is ONE scalar, the sum of a signed value over every edge - a float32 accumulator left 8e-6 ... 1.1e-5 of relative error
// by itself (tests/test_grad_stages.py sat on its 1e-5 gate).  A few thousand float64 adds per training step.
OARD_DEV float chunk_sum_wave(const float* __restrict__ p, size_t stride, int n_chunks, int lane) {
    double s = 0.0;
    for (int ch = lane; ch < n_chunks; ch += 64) s += (double)p[(size_t)ch * stride];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    return (float)s;
}
// dW [MO][MI] and db [MO] from the small kernel's partials: one wave per output
// ldW: row stride of the destination (a column slice of a wider nn.Linear weight); acc: add to the destination instead of overwriting
OARD_KERNEL __global__ __launch_bounds__(256) void k_wgrad_small_reduce(const float* __restrict__ partial, int n_chunks, int MO, int MI,
                                                            float* __restrict__ dW, int ldW, float* __restrict__ db, int acc) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), nout = MO * (MI + 1);
    if (idx >= nout) return;
    const float s = chunk_sum_wave(partial + idx, (size_t)nout, n_chunks, threadIdx.x & 63);
    if ((threadIdx.x & 63) != 0) return;
    const int o = idx / (MI + 1), i = idx % (MI + 1);
    if (i < MI) { if (dW != nullptr) dW[(size_t)o * ldW + i] = acc ? dW[(size_t)o * ldW + i] + s : s; }
    else if (db != nullptr) db[o] = acc ? db[o] + s : s;
}

// second pass: dW[o][i] (dense, logical nn.Linear shape) = sum over chunks in ascending order.  Logical index -> padded index by
// sections (o = s * len + w -> s * pad + w), which undoes the 196 -> 208 padding of split projections.  `transposed`: the
// partials are [x feature][dY feature] (the kernel ran with P = X, Q = dY).
// 256 threads = 32 consecutive outputs x 8 chunk slices: slice s adds chunks s, s + 8, ... in ascending order (eight loads in flight
// per output instead of one thread walking all chunks), the eight slice sums are then added in slice order - a fixed order again.
OARD_DEV void wgrad_reduce_body(float (*red)[33], const float* __restrict__ partial, int n_chunks, int PP, int QP, int transposed, int o_len, int o_pad,
                                int MO, int i_len, int i_pad, int MI, float* __restrict__ out, int ldW, int acc, const unsigned bid) {
    const int oi = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const long long idx = (long long)bid * 32 + oi;
    const bool ok = idx < (long long)MO * MI;
    float s = 0.f;
    int o = 0, i = 0;
    if (ok) {
        o = (int)(idx / MI); i = (int)(idx % MI);
        const int op = (o / o_len) * o_pad + o % o_len, ip = (i / i_len) * i_pad + i % i_len;
        const float* p = partial + (transposed ? (size_t)ip * QP + op : (size_t)op * QP + ip);
        for (int ch = sl; ch < n_chunks; ch += 8) s += p[(size_t)ch * PP * QP];
    }
    red[sl][oi] = s;
    __syncthreads();
    if (sl == 0 && ok) {
        float t = red[0][oi];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][oi];
        float* dst = out + (size_t)o * ldW + i;
        *dst = acc ? *dst + t : t;
    }
}
OARD_KERNEL __global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial, int n_chunks, int PP, int QP, int transposed, int o_len, int o_pad,
                               int MO, int i_len, int i_pad, int MI, float* __restrict__ out, int ldW, int acc) {
    __shared__ float red[8][33];
    wgrad_reduce_body(red, partial, n_chunks, PP, QP, transposed, o_len, o_pad, MO, i_len, i_pad, MI, out, ldW, acc, blockIdx.x);
}
OARD_DEV void bgrad_reduce_body(const float* __restrict__ bpartial, int n_chunks, int stride, int o_len, int o_pad,
                                int MO, float* __restrict__ out, int acc, const unsigned bid) {  // one wave per output
    const int o = bid * 4 + (threadIdx.x >> 6);
    if (o >= MO) return;
    const int op = (o / o_len) * o_pad + o % o_len;
    const float s = chunk_sum_wave(bpartial + op, (size_t)stride, n_chunks, threadIdx.x & 63);
    if ((threadIdx.x & 63) == 0) out[o] = acc ? out[o] + s : s;
}
OARD_KERNEL __global__ __launch_bounds__(256) void k_bgrad_reduce(const float* __restrict__ bpartial, int n_chunks, int stride, int o_len, int o_pad,
                                                      int MO, float* __restrict__ out, int acc) {
    bgrad_reduce_body(bpartial, n_chunks, stride, o_len, o_pad, MO, out, acc, blockIdx.x);
}
// the reduce passes of a job table in one launch: per job rblk_w blocks of the dW reduce, then rblk_b blocks of the bias reduce
OARD_KERNEL __global__ __launch_bounds__(256) void k_wgrad_reduce_q(const WgqJob* __restrict__ jobs, int n_jobs) {
    __shared__ float red[8][33];
    const WgqJob& J = jobs[wgq_find(jobs, n_jobs, blockIdx.x, true)];
    const unsigned lb = blockIdx.x - J.rblk0;
    if (lb < J.rblk_w)
        wgrad_reduce_body(red, J.partial, J.n_chunks, J.PP, J.QP, J.transposed, J.o_len, J.o_pad, J.MO, J.i_len, J.i_pad, J.MI, J.dW, J.ldW, J.acc, lb);
    else
        bgrad_reduce_body(J.bsum, J.n_chunks, J.transposed ? J.QP : J.PP, J.o_len, J.o_pad, J.MO, J.db, J.acc, lb - J.rblk_w);
}
