// oard_edge_b3.h — the GCLMessage edge kernel in SPLIT PRECISION: every fp32 value as three bf16 terms (h, m, l), six products per
// K block (h h, h m, m h, h l, l h, m m) on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  The dropped terms (m l, l m, l l)
// are below 2^-24 of the product, i.e. the result carries fp32 accuracy (emulated on the 684 x 196 product: 2.7e-7 of the largest
// output, plain fp32 6.5e-7) - at 6 x 16 cycles per 16 x 16 x 32 block instead of 8 x 32 cycles on
// v_mfma_f32_16x16x4_f32.  This is an OPTIONAL second formulation (debug option gcl_b3, bench.py --precision bf16x3); the fp32
// kernel of oard_edge_v1.h stays the default and the headline.  <TRAIN> variants write the tape of the training-mode forward
// (debug option train_b3); the backward kernels are fp32.
//
// Same structure as k_gcl_edge_v1<.., RING = 3> (8 waves x 16 edges, weights streamed through three LDS slabs by LDS-DMA, one barrier
// per phase behind the second hook of the phase, edge-state blocks prefetched one phase ahead, stores one phase late).  What changes:
//   * a K block is 32 features = TWO of the fp32 kernel's 16-feature blocks; its A operand is three 1-KiB bf16 chunks.  Lane (g, i) of
//     a chunk holds row 16 t + i and the 8 k values [16 (2 kb) + 4 g + 0..3 | 16 (2 kb + 1) + 4 g + 0..3] - exactly the features lane
//     (g, e) of the column engine holds for its edge e in the two f4 blocks 2 kb and 2 kb + 1 (C layout of the producing MFMA), so
//     the B operand is built from a lane's own registers: no cross-lane traffic.  The K order inside a block is a permutation the
//     packer (k_pack_b3) applies to the weights.
//   * activations are split into (h, m, l) once per use as a B operand: the edge-state blocks of S1, h1 after S1, m after S2.
//   * the gate multiplies S3's accumulator instead of m (W3 (g m) = g (W3 m)), so m is split once.
//   * no 4 x 4 tiles and no compact K tail: the padding of H = 196 to 7 x 32 is carried (12.5 % of the S2 / S3 MFMAs).
// Stream (1-KiB chunks): S1 = NBW K blocks x HT tiles x 3;  S2 = (HT + 1) groups x (1 bias chunk (fp32) + NBH x 3);  S3 = WB groups x
// (1 + NBH x 3).
#pragma once
#include "oard_edge_v1.h"

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#ifndef OARD_B3_HALF
#define OARD_B3_HALF 1          // LDS-DMA issue policy of k_gcl_edge_b3 (see SlabPrefetch): waves 0..3 only / every hook
#endif
#ifndef OARD_B3_PERIOD
#define OARD_B3_PERIOD 1
#endif

template <class D>
struct GclB3Stream {
    static constexpr int HT = D::HT, WB = D::WB, NBH = (HT + 1) / 2, NBW = (WB + 1) / 2;
    static constexpr int G1 = 3 * HT, G2 = 1 + 3 * NBH, NG2 = HT + 1, GP = 2;
    static constexpr int SLAB = (GP * G2 > G1) ? GP * G2 : G1;
    static constexpr int NP1 = NBW, NP2 = (NG2 + GP - 1) / GP, NP3 = (WB + GP - 1) / GP, NPH = NP1 + NP2 + NP3;
    static constexpr int C1 = NBW * G1, C2 = NG2 * G2, C3 = WB * G2, CHUNKS = C1 + C2 + C3;
    static constexpr size_t LDS_BYTES = (size_t)3 * SLAB * 1024;
};

OARD_DEV bf8 lds_b3(const float* sl, int j) { return *reinterpret_cast<const bf8*>(sl + j * 256); }
OARD_DEV f4 mf_b3(bf8 a, bf8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// (a, b) = the lane's two f4 blocks of one K block -> the three bf16 terms of its 8 values
OARD_DEV void split3(f4 a, f4 b, bf8& h, bf8& m, bf8& l) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 hj = (__bf16)x[j];
        const float r = x[j] - (float)hj;
        const __bf16 mj = (__bf16)r;
        const __bf16 lj = (__bf16)(r - (float)mj);
        h[j] = hj; m[j] = mj; l[j] = lj;
    }
}

// M-outer: one output tile = sum over NB K blocks (3 chunks each, slots j0 ..) of the six products; two accumulators alternate
template <int NB, class Hook>
OARD_DEV f4 chain_tile_b3(const float* sl, int j0, const bf8 (&Bh)[NB], const bf8 (&Bm)[NB], const bf8 (&Bl)[NB], f4 init, Hook hook) {
    f4 c0 = init, c1 = f4zero();
    bf8 ah = lds_b3(sl, j0), am = lds_b3(sl, j0 + 1), al = lds_b3(sl, j0 + 2);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        bf8 nh = ah, nm = am, nl = al;
        if (b + 1 < NB) { nh = lds_b3(sl, j0 + 3 * b + 3); nm = lds_b3(sl, j0 + 3 * b + 4); nl = lds_b3(sl, j0 + 3 * b + 5); }
        c0 = mf_b3(ah, Bh[b], c0);
        c1 = mf_b3(ah, Bm[b], c1);
        c0 = mf_b3(am, Bh[b], c0);
        c1 = mf_b3(ah, Bl[b], c1);
        c0 = mf_b3(al, Bh[b], c0);
        c1 = mf_b3(am, Bm[b], c1);
        hook();
        ah = nh; am = nm; al = nl;
    }
    return c0 + c1;
}
// K-outer: acc[t] += the six products of tile t's chunks (slots j0 + 3 t ..) with one K block of the B operand; tiles in pairs so
// that consecutive MFMAs never depend on each other
template <int MT, class Hook>
OARD_DEV void chain_kouter_b3(const float* sl, int j0, bf8 xh, bf8 xm, bf8 xl, f4* acc, Hook hook) {
    bf8 ah0 = lds_b3(sl, j0), am0 = lds_b3(sl, j0 + 1), al0 = lds_b3(sl, j0 + 2);
    bf8 ah1 = ah0, am1 = am0, al1 = al0;
    if (MT > 1) { ah1 = lds_b3(sl, j0 + 3); am1 = lds_b3(sl, j0 + 4); al1 = lds_b3(sl, j0 + 5); }
#pragma unroll
    for (int t = 0; t + 1 < MT; t += 2) {
        bf8 nh0 = ah0, nm0 = am0, nl0 = al0, nh1 = ah1, nm1 = am1, nl1 = al1;
        if (t + 2 < MT) { nh0 = lds_b3(sl, j0 + 3 * (t + 2)); nm0 = lds_b3(sl, j0 + 3 * (t + 2) + 1); nl0 = lds_b3(sl, j0 + 3 * (t + 2) + 2); }
        if (t + 3 < MT) { nh1 = lds_b3(sl, j0 + 3 * (t + 3)); nm1 = lds_b3(sl, j0 + 3 * (t + 3) + 1); nl1 = lds_b3(sl, j0 + 3 * (t + 3) + 2); }
        acc[t] = mf_b3(ah0, xh, acc[t]);     acc[t + 1] = mf_b3(ah1, xh, acc[t + 1]);
        acc[t] = mf_b3(ah0, xm, acc[t]);     acc[t + 1] = mf_b3(ah1, xm, acc[t + 1]);
        acc[t] = mf_b3(am0, xh, acc[t]);     acc[t + 1] = mf_b3(am1, xh, acc[t + 1]);
        hook();
        acc[t] = mf_b3(ah0, xl, acc[t]);     acc[t + 1] = mf_b3(ah1, xl, acc[t + 1]);
        acc[t] = mf_b3(al0, xh, acc[t]);     acc[t + 1] = mf_b3(al1, xh, acc[t + 1]);
        acc[t] = mf_b3(am0, xm, acc[t]);     acc[t + 1] = mf_b3(am1, xm, acc[t + 1]);
        hook();
        ah0 = nh0; am0 = nm0; al0 = nl0; ah1 = nh1; am1 = nm1; al1 = nl1;
    }
    if (MT & 1) {
        f4 c1 = f4zero();
        acc[MT - 1] = mf_b3(ah0, xh, acc[MT - 1]);   c1 = mf_b3(ah0, xm, c1);
        acc[MT - 1] = mf_b3(am0, xh, acc[MT - 1]);   c1 = mf_b3(ah0, xl, c1);
        acc[MT - 1] = mf_b3(al0, xh, acc[MT - 1]);   c1 = mf_b3(am0, xm, c1);
        acc[MT - 1] += c1;
        hook();
    }
}

// Columns are the physical rows [r0, r1); DO_S1 / DO_S3 / TRAIN (new state to ew_out != ew_in, pre-activations to the tape) as in
// k_gcl_edge_v1.
template <class D, bool DO_S1, bool DO_S3, bool TRAIN = false>
__global__ __launch_bounds__(512, 2) void k_gcl_edge_b3(TopoDev tp, const float* __restrict__ stream, const float* __restrict__ P,
                                                        const float* __restrict__ Q, const float* __restrict__ u0,
                                                        const float* __restrict__ c0, long long r0, long long r1, const float* ew_in,
                                                        float* ew_out, float* __restrict__ mbuf, GclTape tape) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using S = GclB3Stream<D>;
    constexpr int HT = D::HT, WB = D::WB, NBH = S::NBH, G1 = S::G1, G2 = S::G2, GP = S::GP, WAVES = 8;
    static_assert(HT >= 2, "the phase barrier sits inside the first chain of a phase");
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SlabPrefetch<WAVES, S::SLAB, OARD_B3_HALF, OARD_B3_PERIOD, 3> pf;
    pf.wave = wave;
    pf.lane_off = (unsigned)lane * 16u;
    auto pf_begin = [&](int p) {
        int start = 0, n = 0;
        if (p < S::NP1) { start = p * G1; n = G1; }
        else if (p < S::NP1 + S::NP2) { const int q = p - S::NP1; start = S::C1 + q * GP * G2; n = min(GP, S::NG2 - q * GP) * G2; }
        else if (p < S::NPH && DO_S3) { const int q = p - S::NP1 - S::NP2; start = S::C1 + S::C2 + q * GP * G2; n = min(GP, WB - q * GP) * G2; }
        pf.begin(stream, smem, p, start, n);
    };
    int bar_left = 0;
    auto SL = [&](int p) -> const float* { return smem + (size_t)(p % 3) * S::SLAB * 256 + lane * 4; };
    auto A = [&](int p, int j) -> f4 { return *reinterpret_cast<const f4*>(SL(p) + j * 256); };
    constexpr int BAR_AT = 2, BAR_AT23 = NBH < 2 ? 1 : 2;       // hook calls in the first chain of a phase: S1 >= 2, S2 / S3 NBH

    const long long c = r0 + ((long long)blockIdx.x * WAVES + wave) * 16 + (lane & 15);
    const size_t e = (size_t)(c < r1 ? c : tp.E);
    const float* erow = ew_in + e * D::WP + 4 * g;
    float* orow = ew_out + e * D::WP + 4 * g;
    f4 h1[HT];
    const size_t eid = (size_t)tp.row_eid[e];
    {
        const int src = tp.row_src[e], tgt = tp.row_tgt[e];
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            h1[t] = ld_blk(P, src, D::HP, t, lane) + ld_blk(Q, tgt, D::HP, t, lane);
            if (!DO_S1) h1[t] += ld_vec(u0, t, lane);
        }
    }
    // the two f4 blocks of K block kb of the edge state (the second may lie beyond WB: zero)
    auto ld_pair = [&](int kb, f4& a, f4& b) {
        a = ld_f4(erow + 32 * kb);
        b = (2 * kb + 1 < WB) ? ld_f4(erow + 32 * kb + 16) : f4zero();
    };
    f4 xa = f4zero(), xb = f4zero();
    if (DO_S1) ld_pair(0, xa, xb);
    int p = DO_S1 ? 0 : S::NP1;
    pf_begin(p); pf.flush();
    pf_begin(p + 1); pf.flush();
    phase_barrier();

    // ---- S1: h1 += W1c . ew   (K-outer over the 32-feature blocks of the edge state) ----------------------
    if (DO_S1) {
        for (int p1 = 0; p1 < S::NP1; ++p1, ++p) {
            bf8 xh, xm, xl;
            split3(xa, xb, xh, xm, xl);
            auto post = [&]() {
                pf_begin(p + 2);
                if (p1 + 1 < S::NP1) ld_pair(p1 + 1, xa, xb);
            };
            auto hook = [&]() { if (bar_left > 0 && --bar_left == 0) { phase_barrier(); post(); } pf.tick(); };
            bar_left = BAR_AT;
            chain_kouter_b3<HT>(SL(p), 0, xh, xm, xl, h1, hook);
            if (bar_left > 0) { bar_left = 0; phase_barrier(); post(); }
            pf.flush();
        }
    }
    if (TRAIN) {
#pragma unroll
        for (int t = 0; t < HT; ++t) st_blk(tape.z1, e, D::HP, t, lane, h1[t]);
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) h1[t] = silu4(h1[t]);
    bf8 bh[NBH], bm[NBH], bl[NBH];
#pragma unroll
    for (int b = 0; b < NBH; ++b) split3(h1[2 * b], (2 * b + 1 < HT) ? h1[2 * b + 1] : f4zero(), bh[b], bm[b], bl[b]);

    // ---- S2: m0 = SiLU(W2 h1 + b2); gate = SiLU(watt . m0 + batt) ------------------------------------------
    f4 m[HT];
    f4 on[GP];
    f4 pz2[TRAIN ? GP : 1];                             // TRAIN: z2 tiles of the previous phase, stored behind the next barrier
    float gate = 0.f;
#pragma unroll
    for (int p2 = 0; p2 < S::NP2; ++p2, ++p) {
        auto post = [&]() {
            if (TRAIN && p2 > 0) {
#pragma unroll
                for (int gg = 0; gg < GP; ++gg)
                    if ((p2 - 1) * GP + gg < HT) st_blk(tape.z2, e, D::HP, (p2 - 1) * GP + gg, lane, pz2[gg]);
            }
            pf_begin(p + 2);
            if (DO_S3 && p2 == S::NP2 - 1) {
#pragma unroll
                for (int gg = 0; gg < GP; ++gg)
                    on[gg] = gg < WB ? (DO_S1 ? ld_f4(erow + 16 * gg) : ld_f4(c0 + 16 * gg + 4 * g)) : f4zero();
            }
        };
        auto hook = [&]() { if (bar_left > 0 && --bar_left == 0) { phase_barrier(); post(); } pf.tick(); };
        bar_left = BAR_AT23;
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int tg = p2 * GP + gg;
            if (tg < S::NG2) {
                const f4 bias = A(p, gg * G2);
                if (tg == HT) {                                 // the gate group is fed with m0: split it now (used again by S3)
#pragma unroll
                    for (int b = 0; b < NBH; ++b) split3(m[2 * b], (2 * b + 1 < HT) ? m[2 * b + 1] : f4zero(), bh[b], bm[b], bl[b]);
                }
                const f4 acc = chain_tile_b3<NBH>(SL(p), gg * G2 + 1, bh, bm, bl, bias, hook);
                if (tg < HT) {
                    if (TRAIN) pz2[gg] = acc;
                    m[tg] = silu4(acc);
                } else {
                    const float av = __shfl(acc.x, lane & 15, 64);
                    if (TRAIN && g == 0) tape.att[e] = av;
                    gate = silu1(av);
                }
            }
        }
        if (bar_left > 0) { bar_left = 0; phase_barrier(); post(); }
        pf.flush();
    }
    if (TRAIN) {                                        // z2 tiles of the last S2 phase
#pragma unroll
        for (int gg = 0; gg < GP; ++gg)
            if ((S::NP2 - 1) * GP + gg < HT) st_blk(tape.z2, e, D::HP, (S::NP2 - 1) * GP + gg, lane, pz2[gg]);
    }
    // ---- S3: ew += SiLU(gate (W3 m0) + b3) -------------------------------------------------------------------
    if (!DO_S3) {
#pragma unroll
        for (int t = 0; t < HT; ++t) st_blk(mbuf, eid, D::HP, t, lane, m[t] * gate);
        return;
    }
    f4 pend[GP], om[GP], pendz[TRAIN ? GP : 1];
    auto s3_phase = [&](int p3, const f4 (&o)[GP], f4 (&onext)[GP]) {
        auto post = [&]() {
            if (p3 == 0) {
#pragma unroll
                for (int t = 0; t < HT; ++t) st_blk(mbuf, eid, D::HP, t, lane, m[t] * gate);
            } else {
#pragma unroll
                for (int gg = 0; gg < GP; ++gg) {
                    st_f4(orow + 16 * ((p3 - 1) * GP + gg), pend[gg]);
                    if (TRAIN) st_f4(tape.z3 + e * D::WP + 4 * g + 16 * ((p3 - 1) * GP + gg), pendz[gg]);
                }
            }
            pf_begin(p + 2);
            if (p3 + 1 < S::NP3) {
#pragma unroll
                for (int gg = 0; gg < GP; ++gg) {
                    const int t = (p3 + 1) * GP + gg;
                    if (t < WB) onext[gg] = DO_S1 ? ld_f4(erow + 16 * t) : ld_f4(c0 + 16 * t + 4 * g);
                }
            }
        };
        auto hook = [&]() { if (bar_left > 0 && --bar_left == 0) { phase_barrier(); post(); } pf.tick(); };
        bar_left = BAR_AT23;
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int t = p3 * GP + gg;
            if (t < WB) {
                const f4 bias = A(p, gg * G2);
                const f4 z = chain_tile_b3<NBH>(SL(p), gg * G2 + 1, bh, bm, bl, f4zero(), hook) * gate + bias;
                if (TRAIN) pendz[gg] = z;
                pend[gg] = o[gg] + silu4(z);
            }
        }
        if (bar_left > 0) { bar_left = 0; phase_barrier(); post(); }
        pf.flush();
        ++p;
    };
    {
        int p3 = 0;
        for (; p3 + 1 < S::NP3; p3 += 2) { s3_phase(p3, on, om); s3_phase(p3 + 1, om, on); }
        if (p3 < S::NP3) s3_phase(p3, on, om);
    }
#pragma unroll
    for (int gg = 0; gg < GP; ++gg) {
        const int t = (S::NP3 - 1) * GP + gg;
        if (t < WB) {
            st_f4(orow + 16 * t, pend[gg]);
            if (TRAIN) st_f4(tape.z3 + e * D::WP + 4 * g + 16 * t, pendz[gg]);
        }
    }
}

// =====================================================================================================================================
// EquiMessage edge part in split precision (inner edges): q[a][third][feature] = dir_proj(ew) * rbf_proj(rbf), as k_equi_edge_v1.
// Every stage runs K-OUTER, so that only ONE K block of the B operand is held as bf16 terms at a time:
//   T1   d1[37 tiles] = dp0b + dir_proj.0 . ew           22 K blocks of the edge state
//        d1 = SiLU(d1) goes to a scratch array in the wave's own lane order (fp32, 1 KiB per tile) - 37 tiles of d1 as three bf16 terms
//        (228 registers) next to 39 output accumulators do not fit, d1 in fp32 next to them neither
//   T2a  cr[39 tiles] = rbf_proj . rbf                     3 K blocks; stored to q (its final place), re-read at the end
//   T2b  cd[39 tiles] = dp2b + dir_proj.2 . d1            19 K blocks, each read back from the scratch array one phase ahead
//   q = cd * cr
// A K block's chunks (tiles x 3) exceed an LDS slab, so a K block is two phases (tile halves); two slabs, barrier at the phase start.
// Stream: T1 = NBW x (D1T x 3);  T2a = NBR x (NO x 3);  T2b = NBD x (NO x 3)   (NO = 3 HT output tiles, third-major).
// =====================================================================================================================================
template <class D>
struct EquiB3Stream {
    static constexpr int WB = D::WB, D1T = D::D1T, RB = D::RB, HT = D::HT, NO = 3 * HT;
    static constexpr int NBW = (WB + 1) / 2, NBR = (RB + 1) / 2, NBD = (D1T + 1) / 2;
    static constexpr int H1A = (D1T + 1) / 2, H1B = D1T - H1A;         // tile halves of T1 (19 + 18)
    static constexpr int H2A = (NO + 1) / 2, H2B = NO - H2A;           // of T2 (20 + 19)
    static constexpr int G1 = 3 * D1T, G2 = 3 * NO;
    static constexpr int SLAB = 3 * (H1A > H2A ? H1A : H2A);
    static constexpr int C1 = NBW * G1, C2A = NBR * G2, C2B = NBD * G2, CHUNKS = C1 + C2A + C2B;
    static constexpr int NPH = 2 * (NBW + NBR + NBD);
    static constexpr size_t LDS_BYTES = (size_t)2 * SLAB * 1024;
};

// TRAIN: zd1 (pre-activation of dir_proj.0, [A+1][D1P]) and cd (dir_proj's output before the product, [A+1][3][HP]) go to the tape.
template <class D, bool TRAIN = false>
__global__ __launch_bounds__(512, 2) void k_equi_edge_b3(TopoDev tp, const float* __restrict__ stream, const float* __restrict__ dp0b,
                                                         const float* __restrict__ dp2b, const float* __restrict__ ew,
                                                         const float* __restrict__ rbuf, float* __restrict__ qbuf,
                                                         float* __restrict__ d1s, float* __restrict__ zd1, float* __restrict__ cdbuf,
                                                         ActList al) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using S = EquiB3Stream<D>;
    constexpr int WB = D::WB, D1T = D::D1T, RB = D::RB, HT = D::HT, NO = S::NO, WAVES = 8;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SlabPrefetch<WAVES, S::SLAB, 0, 2> pf;
    pf.wave = wave;
    pf.lane_off = (unsigned)lane * 16u;
    // phase p: (stage, K block kb, half h) -> chunk range
    auto pf_begin = [&](int p) {
        int start = 0, n = 0;
        if (p < 2 * S::NBW) { const int kb = p >> 1, h = p & 1; start = kb * S::G1 + (h ? 3 * S::H1A : 0); n = 3 * (h ? S::H1B : S::H1A); }
        else if (p < 2 * (S::NBW + S::NBR)) { const int q = p - 2 * S::NBW, kb = q >> 1, h = q & 1;
                                              start = S::C1 + kb * S::G2 + (h ? 3 * S::H2A : 0); n = 3 * (h ? S::H2B : S::H2A); }
        else if (p < S::NPH) { const int q = p - 2 * (S::NBW + S::NBR), kb = q >> 1, h = q & 1;
                               start = S::C1 + S::C2A + kb * S::G2 + (h ? 3 * S::H2A : 0); n = 3 * (h ? S::H2B : S::H2A); }
        pf.begin(stream, smem, p, start, n);
    };
    auto hook = [&]() { pf.tick(); };
    auto SL = [&](int p) -> const float* { return smem + (size_t)(p & 1) * S::SLAB * 256 + lane * 4; };

    // al (round 6, as k_equi_edge_v1): the inner rows inside the cutoff; column c of the launch is list entry c, the message is exactly zero
    // on the other rows and the node stage walks the same list.  No list (training mode): every inner row.
    const long long n_cols = al.n != nullptr ? (long long)*al.n : tp.A;
    if ((long long)blockIdx.x * WAVES * 16 >= n_cols) return;            // (workgroup-uniform: before the first barrier)
    const long long wt = (long long)blockIdx.x * WAVES + wave;           // this wave's 16-edge tile
    const long long c = wt * 16 + (lane & 15);
    const bool live = c < n_cols;
    const size_t a = (size_t)(live ? (al.rows != nullptr ? (long long)al.rows[c] : c) : tp.A);      // padding columns use the spare entry A
    const float* erow = ew + (live ? a : (size_t)tp.E) * D::WP + 4 * g;
    const float* rrow = rbuf + a * D::RP + 4 * g;
    float* d1w = d1s + (size_t)wt * D1T * 256 + lane * 4;                 // [tile t][lane]: this lane's f4 of d1 tile t
    auto ld_pair = [&](const float* row, int kb, int nblk, f4& x, f4& y) {
        x = ld_f4(row + 32 * kb);
        y = (2 * kb + 1 < nblk) ? ld_f4(row + 32 * kb + 16) : f4zero();
    };
    int p = 0;
    f4 xa, xb;
    ld_pair(erow, 0, WB, xa, xb);
    pf_begin(0);
    pf.flush();
    {   // ---- T1 ----
        f4 d1[D1T];
#pragma unroll
        for (int t = 0; t < D1T; ++t) d1[t] = ld_vec(dp0b, t, lane);
        for (int kb = 0; kb < S::NBW; ++kb) {
            bf8 xh, xm, xl;
            split3(xa, xb, xh, xm, xl);
            phase_barrier();
            pf_begin(p + 1);
            if (kb + 1 < S::NBW) ld_pair(erow, kb + 1, WB, xa, xb);
            chain_kouter_b3<S::H1A>(SL(p), 0, xh, xm, xl, d1, hook);
            pf.flush();
            ++p;
            phase_barrier();
            pf_begin(p + 1);
            chain_kouter_b3<S::H1B>(SL(p), 0, xh, xm, xl, d1 + S::H1A, hook);
            pf.flush();
            ++p;
        }
        if (TRAIN) {
#pragma unroll
            for (int t = 0; t < D1T; ++t) st_blk(zd1, a, D::D1P, t, lane, d1[t]);
        }
#pragma unroll
        for (int t = 0; t < D1T; ++t) st_f4(d1w + t * 256, silu4(d1[t]));
    }
    f4 acc[NO];
    {   // ---- T2a: cr = rbf_proj . rbf ----
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = f4zero();
        ld_pair(rrow, 0, RB, xa, xb);
        for (int kb = 0; kb < S::NBR; ++kb) {
            bf8 xh, xm, xl;
            split3(xa, xb, xh, xm, xl);
            phase_barrier();
            pf_begin(p + 1);
            if (kb + 1 < S::NBR) ld_pair(rrow, kb + 1, RB, xa, xb);
            else { xa = ld_f4(d1w); xb = (D1T > 1) ? ld_f4(d1w + 256) : f4zero(); }      // first K block of T2b (d1 tiles 0, 1)
            chain_kouter_b3<S::H2A>(SL(p), 0, xh, xm, xl, acc, hook);
            pf.flush();
            ++p;
            phase_barrier();
            pf_begin(p + 1);
            chain_kouter_b3<S::H2B>(SL(p), 0, xh, xm, xl, acc + S::H2A, hook);
            pf.flush();
            ++p;
        }
    }
    float* qrow = qbuf + a * (size_t)(3 * D::HP) + 4 * g;
#pragma unroll
    for (int o = 0; o < NO; ++o) {                                        // cr -> q (re-read below), accumulators <- bias of dir_proj.2
        st_f4(qrow + (o / HT) * D::HP + 16 * (o % HT), acc[o]);
        acc[o] = ld_vec(dp2b, o, lane);
    }
    {   // ---- T2b: cd = dp2b + dir_proj.2 . d1 ----
        for (int kb = 0; kb < S::NBD; ++kb) {
            bf8 xh, xm, xl;
            split3(xa, xb, xh, xm, xl);
            phase_barrier();
            pf_begin(p + 1);
            if (kb + 1 < S::NBD) {
                xa = ld_f4(d1w + (2 * kb + 2) * 256);
                xb = (2 * kb + 3 < D1T) ? ld_f4(d1w + (2 * kb + 3) * 256) : f4zero();
            }
            chain_kouter_b3<S::H2A>(SL(p), 0, xh, xm, xl, acc, hook);
            pf.flush();
            ++p;
            phase_barrier();
            pf_begin(p + 1);
            chain_kouter_b3<S::H2B>(SL(p), 0, xh, xm, xl, acc + S::H2A, hook);
            pf.flush();
            ++p;
        }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        float* q = qrow + (o / HT) * D::HP + 16 * (o % HT);
        if (TRAIN) st_f4(cdbuf + a * (size_t)(3 * D::HP) + 4 * g + (o / HT) * D::HP + 16 * (o % HT), acc[o]);
        st_f4(q, acc[o] * ld_f4(q));
    }
}

// ---- the stream: built on the device from the natural fp32 packs of the same blob (after k_pack_all) --------------------------------
// One thread per (destination chunk triple, lane): reads the lane's two f4 blocks of the 16-feature chunks 2 kb and 2 kb + 1 of tile t,
// writes its 8 values of the h / m / l chunks.  mode 0: tile-major natural pack [t][KB16] (W2, W3, W1c);  mode 1: a single row vector
// (watt: row 0 of a 1-row tile).
struct B3Job {
    size_t src;                // natural pack (mode 0) / vector (mode 1), floats into the blob
    int KB16;                  // 16-feature chunks per tile of the source
    int MT, NB;                // tiles, K blocks (NB = ceil(KB16 / 2))
    int mode;
    size_t dst;                // floats into the blob: chunk triple (t, kb) at dst + t * tstride + kb * kstride (+ 0 / 256 / 512)
    size_t tstride, kstride;
};
struct B3Jobs { B3Job j[4]; };
OARD_KERNEL __global__ __launch_bounds__(256) void k_pack_b3(B3Jobs jobs, float* __restrict__ blob) {
    const B3Job j = jobs.j[blockIdx.y];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)j.MT * j.NB * 64) return;
    const int lane = (int)(i & 63);
    const size_t ch = i >> 6;
    const int kb = (int)(ch % j.NB), t = (int)(ch / j.NB);
    f4 a = f4zero(), b = f4zero();
    if (j.mode == 0) {
        const float* s = blob + j.src + ((size_t)t * j.KB16 + 2 * kb) * 256 + lane * 4;
        a = ld_f4(s);
        if (2 * kb + 1 < j.KB16) b = ld_f4(s + 256);
    } else if ((lane & 15) == 0) {                               // row 0 only
        const float* s = blob + j.src + 32 * kb + 4 * (lane >> 4);
        a = ld_f4(s);
        if (2 * kb + 1 < j.KB16) b = ld_f4(s + 16);
    }
    bf8 h, m, l;
    split3(a, b, h, m, l);
    float* d = blob + j.dst + (size_t)t * j.tstride + (size_t)kb * j.kstride + lane * 4;
    *reinterpret_cast<bf8*>(d) = h;
    *reinterpret_cast<bf8*>(d + 256) = m;
    *reinterpret_cast<bf8*>(d + 512) = l;
}
// fp32 bias chunks: copies of the fp32 stream's (same C-layout chunk)
OARD_KERNEL __global__ __launch_bounds__(256) void k_copy_chunks(float* __restrict__ blob, size_t src, size_t sstride, size_t dst, size_t dstride, int n) {
    const int c = blockIdx.x;
    if (c < n) blob[dst + (size_t)c * dstride + threadIdx.x] = blob[src + (size_t)c * sstride + threadIdx.x];
}
