// oard_edge_v1.h — the two hot per-layer edge kernels with the weights streamed through LDS.
//
// All waves of a workgroup walk the same sequence of weight chunks, so the sequence is packed once
// per layer in consumption order ("stream") and moved HBM/L2 -> LDS by global_load_lds (LDS-DMA, no
// VGPR round trip) one phase ahead of its use: two slabs, one __syncthreads() per phase.  Every wave
// reads its A operand from LDS (ds_read_b128, lane-linear, conflict-free) and keeps its B operands and
// accumulators in registers.  Per-feature biases travel in the same stream as "bias chunks", node-side
// terms initialise the accumulators before the first DMA is issued, and the only ordinary global loads
// inside the phase loop are the edge-state blocks, prefetched one phase ahead in the same rhythm —
// so no s_waitcnt vmcnt(0) ever has to wait for an in-flight DMA except the one at the phase barrier.
#pragma once
#include "oard_kernels.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// one 1-KiB chunk: per-lane global source, wave-uniform LDS destination (+ lane*16 by hardware)
OARD_DEV void glds16(const float* gsrc_lane, float* lds_chunk) {
#ifndef OARD_DBG_NODMA
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)gsrc_lane, (lds_ptr_t)lds_chunk, 16, 0, 0);
#endif
}

// Phase barrier: every LDS-DMA piece this wave issued must have landed before any wave reads the slab.
// hipcc's own "vmcnt(0) before the workgroup barrier" is NOT emitted once the DMA issues sit in
// data-dependent control flow inside a loop (observed: only lgkmcnt(0) before s_barrier), so the wait
// is stated explicitly.
OARD_DEV void phase_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// acc[i] += a x b[i] for NA independent accumulators, k-steps outermost so consecutive MFMAs never
// depend on each other (v_mfma_f32_16x16x4_f32: 32-cycle issue, 40-cycle dependent latency)
template <int NA>
OARD_DEV void mma_shared_a(f4 a, const f4 (&b)[NA], f4 (&acc)[NA]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[i].x, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[i].y, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[i].z, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[i].w, acc[i], 0, 0, 0);
}
// two (a, b, acc) triples interleaved
OARD_DEV void mma_pair(f4 a0, f4 b0, f4& c0, f4 a1, f4 b1, f4& c1) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, c1, 0, 0, 0);
}

// ---- software-pipelined LDS -> MFMA chains ---------------------------------------------------------
// `sl` = this lane's pointer into the current slab (slab base + lane*4); chunk j is at sl + j*256.
// The A fragments of the NEXT pair of chunks are read from LDS while the current pair's 8 MFMAs issue,
// so the ds_read latency (~100+ cycles) is covered; sched_group_barrier pins that order.
#ifdef OARD_DBG_NOLDS
OARD_DEV f4 lds_a(const float* sl, int j) { return (f4){1.0f + j, 0.5f, 0.25f, 2.0f}; }       // timing experiment only
#else
OARD_DEV f4 lds_a(const float* sl, int j) { return *reinterpret_cast<const f4*>(sl + j * 256); }
#endif
#ifdef OARD_DBG_NOEPI
#define EPI_SILU4(x) (x)
#else
#define EPI_SILU4(x) silu4(x)
#endif
#ifdef OARD_USE_SCHED
#define OARD_SCHED_PAIR() do { __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); \
                               __builtin_amdgcn_sched_group_barrier(0x008, 8, 0); } while (0)
#define OARD_SCHED_OPEN() __builtin_amdgcn_sched_group_barrier(0x100, 2, 0)
#else
#define OARD_SCHED_PAIR() do { } while (0)
#define OARD_SCHED_OPEN() do { } while (0)
#endif

// M-outer: one output tile = sum over KB chunks (slots j0..j0+KB-1) x in[b]; even/odd accumulators
struct NoHook { OARD_DEV void operator()() const {} };
template <int KB, class Hook = NoHook>
OARD_DEV f4 chain_tile(const float* sl, int j0, const f4 (&in)[KB], f4 init, Hook hook = Hook()) {
    f4 c0 = init, c1 = f4zero();
    f4 a0 = lds_a(sl, j0), a1 = KB > 1 ? lds_a(sl, j0 + 1) : f4zero();
    OARD_SCHED_OPEN();      // the first pair's reads open the pipeline
#pragma unroll
    for (int b = 0; b + 1 < KB; b += 2) {
        f4 n0 = a0, n1 = a1;
        if (b + 2 < KB) n0 = lds_a(sl, j0 + b + 2);
        if (b + 3 < KB) n1 = lds_a(sl, j0 + b + 3);
        mma_pair(a0, in[b], c0, a1, in[b + 1], c1);
        OARD_SCHED_PAIR();
        hook();
        a0 = n0; a1 = n1;
    }
    if (KB & 1) c0 = mma_chunk(a0, in[KB - 1], c0);
    return c0 + c1;
}
// K-outer: acc[t] += chunk(j0 + t) x x for t < MT, pairs of tiles interleaved
template <int MT, class Hook = NoHook>
OARD_DEV void chain_kouter(const float* sl, int j0, f4 x, f4 (&acc)[MT], Hook hook = Hook()) {
    f4 a0 = lds_a(sl, j0), a1 = MT > 1 ? lds_a(sl, j0 + 1) : f4zero();
    OARD_SCHED_OPEN();
#pragma unroll
    for (int t = 0; t + 1 < MT; t += 2) {
        f4 n0 = a0, n1 = a1;
        if (t + 2 < MT) n0 = lds_a(sl, j0 + t + 2);
        if (t + 3 < MT) n1 = lds_a(sl, j0 + t + 3);
        mma_pair(a0, x, acc[t], a1, x, acc[t + 1]);
        OARD_SCHED_PAIR();
        hook();
        a0 = n0; a1 = n1;
    }
    if (MT & 1) acc[MT - 1] = mma_chunk(a0, x, acc[MT - 1]);
}

// =====================================================================================================
// GCLMessage edge part, all edges.  Stream (chunks): S1 = WB groups x HT  [W1c, K-outer];
// S2 = (HT+1) groups x (1+HT)  [bias b2 | W2 tile t] + gate group [batt | watt as a 1-row tile];
// S3 = WB groups x (1+HT)  [bias b3 | W3 tile t].
// =====================================================================================================
template <class D, int GP>
struct GclStream {
    static constexpr int HT = D::HT, WB = D::WB, G1 = HT, G2 = HT + 1, NG2 = HT + 1;
    static constexpr int SLAB = GP * G2;                       // chunks per slab
    static constexpr int NP1 = (WB + GP - 1) / GP, NP2 = (NG2 + GP - 1) / GP, NP3 = NP1, NPH = NP1 + NP2 + NP3;
    static constexpr int C1 = WB * G1, C2 = NG2 * G2, C3 = WB * G2, CHUNKS = C1 + C2 + C3;
    static constexpr size_t LDS_BYTES = (size_t)2 * SLAB * 1024;
};

// NB == 1 variants are held to <= 256 registers (2 waves per SIMD).  Besides occupancy this keeps the
// accumulators out of the AGPR half of the file: with the default bound hipcc (ROCm 7.2) put some S1
// accumulators in AGPRs for the 4-wave variant and the result drifted to 6e-5 of the oracle (a missed
// MFMA->v_accvgpr_read hazard is the suspect); with the bound every variant is at 2.5e-7.
// DO_S1 = false: the columns are inter-object edges of layer 0, whose state is the constant row: stage S1
//   collapses to the precomputed vector u0 (added to the accumulator init) and its DMA is skipped.
// DO_S3 = false: the columns are inter-object edges of the last layer, whose updated state nobody reads
//   (EquiMessage only touches inner edges): the residual stage and its edge-state traffic are skipped.
// Columns are the physical rows [r0, r1).
template <class D, int NB, int WAVES, int GP, int PRIO, bool DO_S1, bool DO_S3>
__global__ __launch_bounds__(WAVES * 64, NB == 1 ? (PRIO == 3 ? 3 : 2) : 1) void k_gcl_edge_v1(TopoDev tp, const float* __restrict__ stream,
                                                            const float* __restrict__ P, const float* __restrict__ Q,
                                                            const float* __restrict__ u0, const float* __restrict__ c0,
                                                            long long r0, long long r1,
                                                            float* __restrict__ ew, float* __restrict__ mbuf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using S = GclStream<D, GP>;
    constexpr int HT = D::HT, WB = D::WB, G1 = S::G1, G2 = S::G2;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // provably wave-uniform
    // static priority for the younger half: the two waves sharing a SIMD then take the matrix pipe in
    // turns (one runs its MFMA chain while the other does its VALU epilogue) instead of in lockstep
    if (PRIO == 1 && wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);      // PRIO == 3: no priority, but 3 waves per SIMD (<= 168 registers)

    // DMA prefetch of the next phase, spread over the current phase: one global_load_lds costs the
    // issuing wave ~100-180 cycles, so the pieces are issued one at a time between MFMA pairs (every
    // third pair), staggered between the two waves that share a SIMD, instead of in a burst after the barrier
    constexpr int KMAX = (S::SLAB + WAVES - 1) / WAVES;        // pieces per wave per phase (upper bound)
    const float* pf_src = stream;
    float* pf_dst = smem;
    int pf_n = 0, pf_k = 0, pf_next = 0;
    auto pf_begin = [&](int p) {                               // p = phase to prefetch
        int start = 0, n = 0;
        if (p < S::NP1) { start = p * GP * G1; n = min(GP, WB - p * GP) * G1; }
        else if (p < S::NP1 + S::NP2) { const int q = p - S::NP1; start = S::C1 + q * GP * G2; n = min(GP, S::NG2 - q * GP) * G2; }
        else if (p < S::NPH && DO_S3) { const int q = p - S::NP1 - S::NP2; start = S::C1 + S::C2 + q * GP * G2; n = min(GP, WB - q * GP) * G2; }
        pf_src = stream + (size_t)start * 256 + lane * 4;
        pf_dst = smem + (size_t)(p & 1) * S::SLAB * 256;
        pf_n = n; pf_k = 0; pf_next = 1 + (wave >= WAVES / 2 ? 1 : 0);
    };
#ifdef OARD_STAGE_REG_GCL
    // register-staged alternative: plain 1-KiB loads right after the barrier, ds_write_b128 before the next one
    f4 stage[KMAX];
    auto hook = [&]() {};
    auto pf_flush = [&]() {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int j = wave + k * WAVES;
            if (j < pf_n) *reinterpret_cast<f4*>(pf_dst + j * 256 + lane * 4) = stage[k];
        }
    };
    auto pf_load = [&]() {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int j = wave + k * WAVES;
            if (j < pf_n) stage[k] = ld_f4(pf_src + (size_t)j * 256);
        }
    };
    auto issue = [&](int p) { pf_begin(p); pf_load(); pf_flush(); };
    auto pf_start = [&](int p) { pf_begin(p); pf_load(); };
#else
    auto pf_one = [&]() {
        const int j = wave + pf_k * WAVES;
        if (j < pf_n) glds16(pf_src + (size_t)j * 256, pf_dst + j * 256);
        ++pf_k;
    };
#ifndef OARD_NO_HOOK
    auto hook = [&]() { if (--pf_next == 0) { pf_one(); pf_next = 3; } };
#else
    auto hook = [&]() { while (pf_k < KMAX) pf_one(); };       // burst at the first hook
#endif
    auto pf_flush = [&]() { while (pf_k < KMAX) pf_one(); };
    auto issue = [&](int p) { pf_begin(p); pf_flush(); };      // burst form (prologue only)
    auto pf_start = [&](int p) { pf_begin(p); };
#endif
    auto A = [&](int p, int j) -> f4 {
        return *reinterpret_cast<const f4*>(smem + ((size_t)(p & 1) * S::SLAB + j) * 256 + lane * 4);
    };
    auto SL = [&](int p) -> const float* { return smem + (size_t)(p & 1) * S::SLAB * 256 + lane * 4; };

    // columns of this wave; padding columns work on the spare row E of ew / mbuf (allocated for that
    // purpose), so the kernel has no validity branches and every wave stays in the barrier protocol
    const long long colbase = r0 + ((long long)blockIdx.x * WAVES + wave) * (NB * 16) + (lane & 15);
    size_t e[NB];
    float* erow[NB];
    f4 h1[NB][HT];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const long long c = colbase + nb * 16;
        e[nb] = (size_t)(c < r1 ? c : tp.E);
        erow[nb] = ew + e[nb] * D::WP + 4 * g;
        const int src = tp.row_src[e[nb]], tgt = tp.row_tgt[e[nb]];
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            h1[nb][t] = ld_blk(P, src, D::HP, t, lane) + ld_blk(Q, tgt, D::HP, t, lane);
            if (!DO_S1) h1[nb][t] += ld_vec(u0, t, lane);
        }
    }
    f4 xn[GP][NB];
#pragma unroll
    for (int gg = 0; gg < GP; ++gg)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) xn[gg][nb] = (DO_S1 && gg < WB) ? ld_f4(erow[nb] + 16 * gg) : f4zero();
    int p = DO_S1 ? 0 : S::NP1;
    issue(p);

    // ---- S1: h1 += W1c . ew   (K-outer) -------------------------------------------------------------
    for (int p1 = 0; DO_S1 && p1 < S::NP1; ++p1, ++p) {
        phase_barrier();
        f4 x[GP][NB];
#pragma unroll
        for (int gg = 0; gg < GP; ++gg)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) x[gg][nb] = xn[gg][nb];
        pf_start(p + 1);
        if (p1 + 1 < S::NP1) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) {
                const int b = (p1 + 1) * GP + gg;
                if (b < WB)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) xn[gg][nb] = ld_f4(erow[nb] + 16 * b);
            }
        }
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            if (p1 * GP + gg < WB) {
                if (NB >= 2) {
#pragma unroll
                    for (int t = 0; t < HT; ++t) {
                        const f4 a = A(p, gg * G1 + t);
                        f4 acc[NB], xb[NB];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) { acc[nb] = h1[nb][t]; xb[nb] = x[gg][nb]; }
                        mma_shared_a<NB>(a, xb, acc);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) h1[nb][t] = acc[nb];
                    }
                } else {
                    chain_kouter<HT>(SL(p), gg * G1, x[gg][0], h1[0], hook);
                }
            }
        }
        pf_flush();
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int t = 0; t < HT; ++t) h1[nb][t] = EPI_SILU4(h1[nb][t]);

    // ---- S2: m = SiLU(W2 h1 + b2); gate = SiLU(watt . m + batt) ---------------------------------------
    f4 m[NB][HT];
    f4 on[GP][NB];
    // note: the gate tile is computed from h1?  no — from m; it is the last group of S2 and uses m as B operand
#pragma unroll
    for (int p2 = 0; p2 < S::NP2; ++p2, ++p) {
        phase_barrier();
        pf_start(p + 1);
        if (DO_S3 && p2 == S::NP2 - 1) {            // prefetch the old edge-state tiles of S3's first phase
#pragma unroll
            for (int gg = 0; gg < GP; ++gg)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    on[gg][nb] = gg < WB ? (DO_S1 ? ld_f4(erow[nb] + 16 * gg) : ld_f4(c0 + 16 * gg + 4 * g)) : f4zero();
        }
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int tg = p2 * GP + gg;            // compile-time after unrolling
            if (tg < S::NG2) {
                const f4 bias = A(p, gg * G2);
                if (NB >= 2) {
                    f4 acc[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = bias;
#pragma unroll
                    for (int b = 0; b < HT; ++b) {
                        f4 xb[NB];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) xb[nb] = tg < HT ? h1[nb][b] : m[nb][b];
                        mma_shared_a<NB>(A(p, gg * G2 + 1 + b), xb, acc);
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        if (tg < HT) m[nb][tg] = EPI_SILU4(acc[nb]);
                        else {
                            const float gate = silu1(__shfl(acc[nb].x, lane & 15, 64));
#pragma unroll
                            for (int t = 0; t < HT; ++t) m[nb][t] *= gate;
                        }
                    }
                } else {
                    const f4 acc = tg < HT ? chain_tile<HT>(SL(p), gg * G2 + 1, h1[0], bias, hook)
                                           : chain_tile<HT>(SL(p), gg * G2 + 1, m[0], bias, hook);
                    if (tg < HT) m[0][tg] = EPI_SILU4(acc);
                    else {
                        const float gate = silu1(__shfl(acc.x, lane & 15, 64));
#pragma unroll
                        for (int t = 0; t < HT; ++t) m[0][t] *= gate;
                    }
                }
            }
        }
        pf_flush();
    }
    // ---- S3: ew += SiLU(W3 m + b3), one output tile per group ----------------------------------------
    // stores are issued one phase late (right after the next barrier) so that the barrier's vmcnt(0)
    // never waits for a store that was issued a few cycles earlier
    if (!DO_S3) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int t = 0; t < HT; ++t) st_blk(mbuf, (size_t)tp.row_eid[e[nb]], D::HP, t, lane, m[nb][t]);
        return;
    }
    f4 pend[GP][NB];
    for (int p3 = 0; p3 < S::NP3; ++p3, ++p) {
        phase_barrier();
        if (p3 == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int t = 0; t < HT; ++t) st_blk(mbuf, (size_t)tp.row_eid[e[nb]], D::HP, t, lane, m[nb][t]);
        } else {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) st_f4(erow[nb] + 16 * ((p3 - 1) * GP + gg), pend[gg][nb]);   // (p3-1)*GP+gg < WB always
        }
        f4 o[GP][NB];
#pragma unroll
        for (int gg = 0; gg < GP; ++gg)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) o[gg][nb] = on[gg][nb];
        pf_start(p + 1);
        if (p3 + 1 < S::NP3) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) {
                const int t = (p3 + 1) * GP + gg;
                if (t < WB)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)      // !DO_S1: the old state of these rows IS the constant row (never materialised)
                        on[gg][nb] = DO_S1 ? ld_f4(erow[nb] + 16 * t) : ld_f4(c0 + 16 * t + 4 * g);
            }
        }
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int t = p3 * GP + gg;
            if (t < WB) {
                const f4 bias = A(p, gg * G2);
                if (NB >= 2) {
                    f4 acc[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = bias;
#pragma unroll
                    for (int b = 0; b < HT; ++b) {
                        f4 xb[NB];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) xb[nb] = m[nb][b];
                        mma_shared_a<NB>(A(p, gg * G2 + 1 + b), xb, acc);
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) pend[gg][nb] = o[gg][nb] + EPI_SILU4(acc[nb]);
                } else {
                    pend[gg][0] = o[gg][0] + EPI_SILU4(chain_tile<HT>(SL(p), gg * G2 + 1, m[0], bias, hook));
                }
            }
        }
        pf_flush();
    }
#pragma unroll
    for (int gg = 0; gg < GP; ++gg) {
        const int t = (S::NP3 - 1) * GP + gg;
        if (t < WB)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) st_f4(erow[nb] + 16 * t, pend[gg][nb]);
    }
}

// =====================================================================================================
// Stage S3 of the GCL edge pass as its own kernel (split schedule): ew += SiLU(W3 m + b3) with m read back
// from the message buffer.  Without h1 it needs ~half the registers of the fused kernel, so it can run two
// column blocks per wave (each streamed chunk feeds 8 MFMAs) at 2-3 waves per SIMD.
// OLD_C0: the rows are inter-object edges of layer 0 whose old state is the constant row (never materialised).
// =====================================================================================================
template <class D, int NB, int WAVES, int GP, bool OLD_C0>
__global__ __launch_bounds__(WAVES * 64, 2) void k_gcl_s3_v1(TopoDev tp, const float* __restrict__ stream,
                                                             const float* __restrict__ c0, long long r0, long long r1,
                                                             const float* __restrict__ mbuf, float* __restrict__ ew) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using S = GclStream<D, GP>;
    constexpr int HT = D::HT, WB = D::WB, G2 = S::G2, NP3 = S::NP3;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int KMAX = (S::SLAB + WAVES - 1) / WAVES;
    const float* pf_src = stream;
    float* pf_dst = smem;
    int pf_n = 0, pf_k = 0, pf_next = 0;
    auto pf_begin = [&](int q) {                               // q = S3 phase to prefetch
        int start = S::C1 + S::C2, n = 0;
        if (q < NP3) { start += q * GP * G2; n = min(GP, WB - q * GP) * G2; }
        pf_src = stream + (size_t)start * 256 + lane * 4;
        pf_dst = smem + (size_t)(q & 1) * S::SLAB * 256;
        pf_n = n; pf_k = 0; pf_next = 1 + (wave >= WAVES / 2 ? 1 : 0);
    };
    auto pf_one = [&]() {
        const int j = wave + pf_k * WAVES;
        if (j < pf_n) glds16(pf_src + (size_t)j * 256, pf_dst + j * 256);
        ++pf_k;
    };
    auto hook = [&]() { if (--pf_next == 0) { pf_one(); pf_next = 3; } };
    auto pf_flush = [&]() { while (pf_k < KMAX) pf_one(); };
    auto SL = [&](int q) -> const float* { return smem + (size_t)(q & 1) * S::SLAB * 256 + lane * 4; };

    const long long colbase = r0 + ((long long)blockIdx.x * WAVES + wave) * (NB * 16) + (lane & 15);
    float* erow[NB];
    f4 m[NB][HT];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const long long c = colbase + nb * 16;
        const size_t e = (size_t)(c < r1 ? c : tp.E);
        erow[nb] = ew + e * D::WP + 4 * g;
        const size_t mrow = (size_t)tp.row_eid[e];
#pragma unroll
        for (int t = 0; t < HT; ++t) m[nb][t] = ld_blk(mbuf, mrow, D::HP, t, lane);
    }
    f4 on[GP][NB], pend[GP][NB];
#pragma unroll
    for (int gg = 0; gg < GP; ++gg)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            on[gg][nb] = gg < WB ? (OLD_C0 ? ld_f4(c0 + 16 * gg + 4 * g) : ld_f4(erow[nb] + 16 * gg)) : f4zero();
    pf_begin(0);
    pf_flush();
    for (int q = 0; q < NP3; ++q) {
        phase_barrier();
        if (q > 0) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) st_f4(erow[nb] + 16 * ((q - 1) * GP + gg), pend[gg][nb]);
        }
        f4 o[GP][NB];
#pragma unroll
        for (int gg = 0; gg < GP; ++gg)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) o[gg][nb] = on[gg][nb];
        pf_begin(q + 1);
        if (q + 1 < NP3) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) {
                const int t = (q + 1) * GP + gg;
                if (t < WB)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        on[gg][nb] = OLD_C0 ? ld_f4(c0 + 16 * t + 4 * g) : ld_f4(erow[nb] + 16 * t);
            }
        }
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int t = q * GP + gg;
            if (t < WB) {
                const f4 bias = lds_a(SL(q), gg * G2);
                if (NB >= 2) {
                    f4 acc[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = bias;
#pragma unroll
                    for (int b = 0; b < HT; ++b) {
                        f4 xb[NB];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) xb[nb] = m[nb][b];
                        mma_shared_a<NB>(lds_a(SL(q), gg * G2 + 1 + b), xb, acc);
                        if (b & 1) hook();
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) pend[gg][nb] = o[gg][nb] + EPI_SILU4(acc[nb]);
                } else {
                    pend[gg][0] = o[gg][0] + EPI_SILU4(chain_tile<HT>(SL(q), gg * G2 + 1, m[0], bias, hook));
                }
            }
        }
        pf_flush();
    }
#pragma unroll
    for (int gg = 0; gg < GP; ++gg) {
        const int t = (NP3 - 1) * GP + gg;
        if (t < WB)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) st_f4(erow[nb] + 16 * t, pend[gg][nb]);
    }
}

// =====================================================================================================
// EquiMessage edge part, inner edges sorted by target.  Stream: T1 = WB groups x D1T [dir_proj.0, K-outer];
// T2 = 3*HT groups (order tt-major, third-minor) x (1 + D1T + RB) [bias dp2b | dir_proj.2 tile | rbf_proj tile].
// Output: q[a][third][feature] = (dir_proj(ew))[..] * (rbf_proj(rbf))[..]  — the node kernel forms the messages.
// =====================================================================================================
template <class D>
struct EquiStream {
    static constexpr int WB = D::WB, D1T = D::D1T, RB = D::RB, HT = D::HT;
    static constexpr int G1 = D1T, G2 = 1 + D1T + RB, NG2 = 3 * HT;
    static constexpr int SLAB = G2 > G1 ? G2 : G1;
    static constexpr int NPH = WB + NG2;
    static constexpr int C1 = WB * G1, CHUNKS = C1 + NG2 * G2;
    static constexpr size_t LDS_BYTES = (size_t)2 * SLAB * 1024;
};

template <class D, int WAVES, int PRIO>
__global__ __launch_bounds__(WAVES * 64) void k_equi_edge_v1(TopoDev tp, const float* __restrict__ stream,
                                                             const float* __restrict__ dp0b,
                                                             const float* __restrict__ ew, const float* __restrict__ rbuf,
                                                             float* __restrict__ qbuf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using S = EquiStream<D>;
    constexpr int WB = D::WB, D1T = D::D1T, RB = D::RB, HT = D::HT, G1 = S::G1, G2 = S::G2;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (PRIO && wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);

    constexpr int KMAX = (S::SLAB + WAVES - 1) / WAVES;
    const float* pf_src = stream;
    float* pf_dst = smem;
    int pf_n = 0, pf_k = 0, pf_next = 0;
    auto pf_begin = [&](int p) {
        const int start = p < WB ? p * G1 : S::C1 + (p - WB) * G2;
        pf_n = p >= S::NPH ? 0 : (p < WB ? G1 : G2);
        pf_src = stream + (size_t)start * 256 + lane * 4;
        pf_dst = smem + (size_t)(p & 1) * S::SLAB * 256;
        pf_k = 0; pf_next = 1 + (wave >= WAVES / 2 ? 1 : 0);
    };
#ifdef OARD_STAGE_REG_EQUI
    f4 stage[KMAX];
    auto hook = [&]() {};
    auto pf_flush = [&]() {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int j = wave + k * WAVES;
            if (j < pf_n) *reinterpret_cast<f4*>(pf_dst + j * 256 + lane * 4) = stage[k];
        }
    };
    auto pf_load = [&]() {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int j = wave + k * WAVES;
            if (j < pf_n) stage[k] = ld_f4(pf_src + (size_t)j * 256);
        }
    };
    auto issue = [&](int p) { pf_begin(p); pf_load(); pf_flush(); };
    auto pf_start = [&](int p) { pf_begin(p); pf_load(); };
#else
    auto pf_one = [&]() {
        const int j = wave + pf_k * WAVES;
        if (j < pf_n) glds16(pf_src + (size_t)j * 256, pf_dst + j * 256);
        ++pf_k;
    };
#ifndef OARD_NO_HOOK
    auto hook = [&]() { if (--pf_next == 0) { pf_one(); pf_next = 3; } };
#else
    auto hook = [&]() { while (pf_k < KMAX) pf_one(); };
#endif
    auto pf_flush = [&]() { while (pf_k < KMAX) pf_one(); };
    auto issue = [&](int p) { pf_begin(p); pf_flush(); };
    auto pf_start = [&](int p) { pf_begin(p); };
#endif
    auto A = [&](int p, int j) -> f4 {
        return *reinterpret_cast<const f4*>(smem + ((size_t)(p & 1) * S::SLAB + j) * 256 + lane * 4);
    };
    auto SL = [&](int p) -> const float* { return smem + (size_t)(p & 1) * S::SLAB * 256 + lane * 4; };

    const long long c = ((long long)blockIdx.x * WAVES + wave) * 16 + (lane & 15);
    const size_t a = (size_t)(c < tp.A ? c : tp.A);           // padding columns use the spare entry A
    const float* erow = ew + (c < tp.A ? a : (size_t)tp.E) * D::WP + 4 * g;     // inner entry a == physical row a
    f4 d1[D1T];
#pragma unroll
    for (int t = 0; t < D1T; ++t) d1[t] = ld_vec(dp0b, t, lane);
    f4 rb[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) rb[b] = ld_blk(rbuf, a, D::RP, b, lane);
    f4 xn = ld_f4(erow);
    issue(0);

    int p = 0;
    for (int b = 0; b < WB; ++b, ++p) {
        phase_barrier();
        const f4 x = xn;
        pf_start(p + 1);
        if (b + 1 < WB) xn = ld_f4(erow + 16 * (b + 1));
        chain_kouter<D1T>(SL(p), 0, x, d1, hook);
        pf_flush();
    }
#pragma unroll
    for (int t = 0; t < D1T; ++t) d1[t] = EPI_SILU4(d1[t]);

    float* qrow = qbuf + a * (size_t)(3 * D::HP) + 4 * g;
    f4 pend = f4zero();
    int pend_off = 0;
    for (int i = 0; i < S::NG2; ++i, ++p) {
        phase_barrier();
        if (i > 0) st_f4(qrow + pend_off, pend);               // store of the previous phase, issued one phase late
        pf_start(p + 1);
        const f4 cd = chain_tile<D1T>(SL(p), 1, d1, A(p, 0), hook);
        const f4 cr = chain_tile<RB>(SL(p), 1 + D1T, rb, f4zero(), hook);
        pf_flush();
        const int tt = i / 3, th = i - 3 * tt;
        pend = cd * cr;
        pend_off = th * D::HP + 16 * tt;
    }
    st_f4(qrow + pend_off, pend);
}

// =====================================================================================================
// message formation + aggregation (:264-283, 857-859) + first half of EquiUpdate, from q:
//   (x, a2, a3) = (xq[src] + xq[n]) * q;  dx = sum x;  dvec = sum (vec[src] * a2/sqrt3 + a3 * coord_diff)/sqrt(H)
// reads vec_in (all nodes), writes vec_out (own node) — the two must be different buffers.
// =====================================================================================================
template <class D>
__global__ __launch_bounds__(256) void k_equi_agg_v1(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                     const float* __restrict__ qbuf, const float* __restrict__ xq,
                                                     const float* __restrict__ geo, const float* __restrict__ x1,
                                                     float* __restrict__ s, const float* __restrict__ vec_in,
                                                     float* __restrict__ vec_out, float* __restrict__ v2buf,
                                                     float* __restrict__ scal, float* __restrict__ vdot) {
    bool live; const ColId id = col_id(tp.N, live);
    if (!live) return;
    const int n = id.col, a0 = tp.act_ptr[n], cnt = tp.act_ptr[n + 1] - a0;
    const int mx = wave_max(cnt);
    const float inv_sqrt2 = 0.70710678118654752f, inv_sqrt3 = 0.57735026918962576f,
                inv_sqrt_h = 1.0f / sqrtf((float)D::H);
    f4 vx[3][D::HT], dx[D::HT];
#pragma unroll
    for (int t = 0; t < D::HT; ++t) { dx[t] = f4zero(); vx[0][t] = f4zero(); vx[1][t] = f4zero(); vx[2][t] = f4zero(); }
    for (int k = 0; k < mx; ++k)
        if (k < cnt) {
            const size_t a = (size_t)a0 + k;
            const int m = tp.act_src[a];
            const float* g = geo + a * GEO_STRIDE;
            const float ux = g[2], uy = g[3], uz = g[4];
#pragma unroll
            for (int t = 0; t < D::HT; ++t) {
                const f4 q0 = ld_blk(qbuf, a, 3 * D::HP, t, id.lane);
                const f4 q1 = ld_blk(qbuf, a, 3 * D::HP, D::HT + t, id.lane);
                const f4 q2 = ld_blk(qbuf, a, 3 * D::HP, 2 * D::HT + t, id.lane);
                const f4 xs0 = ld_blk(xq, m, 3 * D::HP, t, id.lane) + ld_blk(xq, n, 3 * D::HP, t, id.lane);
                const f4 xs1 = ld_blk(xq, m, 3 * D::HP, D::HT + t, id.lane) + ld_blk(xq, n, 3 * D::HP, D::HT + t, id.lane);
                const f4 xs2 = ld_blk(xq, m, 3 * D::HP, 2 * D::HT + t, id.lane) + ld_blk(xq, n, 3 * D::HP, 2 * D::HT + t, id.lane);
                dx[t] += xs0 * q0;
                const f4 a2 = xs1 * q1 * inv_sqrt3, a3 = xs2 * q2;
                vx[0][t] += (ld_blk(vec_in, (size_t)m * 3 + 0, D::HP, t, id.lane) * a2 + a3 * ux) * inv_sqrt_h;
                vx[1][t] += (ld_blk(vec_in, (size_t)m * 3 + 1, D::HP, t, id.lane) * a2 + a3 * uy) * inv_sqrt_h;
                vx[2][t] += (ld_blk(vec_in, (size_t)m * 3 + 2, D::HP, t, id.lane) * a2 + a3 * uz) * inv_sqrt_h;
            }
        }
#pragma unroll
    for (int t = 0; t < D::HT; ++t) {
        const f4 sn = (ld_blk(s, n, D::HP, t, id.lane) + dx[t]) * inv_sqrt2;
        if (id.valid) st_blk(s, n, D::HP, t, id.lane, sn);
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            vx[x][t] += ld_blk(vec_in, (size_t)n * 3 + x, D::HP, t, id.lane);
            if (id.valid) st_blk(vec_out, (size_t)n * 3 + x, D::HP, t, id.lane, vx[x][t]);
        }
    }
    const float fx = x1[n * 3], fy = x1[n * 3 + 1], fz = x1[n * 3 + 2];
    const float* l3 = wb + lo.l3u;
    for (int t = 0; t < D::HT; ++t) {
        f4 v1[3] = {f4zero(), f4zero(), f4zero()}, v2[3] = {f4zero(), f4zero(), f4zero()};
        const float* w1 = wb + lo.vp + ((size_t)t * D::HT * 64 + id.lane) * 4;
        const float* w2 = wb + lo.vp + ((size_t)(D::HT + t) * D::HT * 64 + id.lane) * 4;
#pragma unroll
        for (int b = 0; b < D::HT; ++b) {
            const f4 c1 = ld_f4(w1 + (size_t)b * 256), c2 = ld_f4(w2 + (size_t)b * 256);
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                v1[x] = mma_chunk(c1, vx[x][b], v1[x]);
                v2[x] = mma_chunk(c2, vx[x][b], v2[x]);
            }
        }
        const f4 sc = v1[0] * fx + v1[1] * fy + v1[2] * fz;
        const f4 vd = (v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2]) * inv_sqrt_h;
        f4 sca;
        const int f0 = 16 * t + 4 * id.g;
        sca.x = f0 + 0 < D::H ? lin3u(l3, sc.x) : 0.f;
        sca.y = f0 + 1 < D::H ? lin3u(l3, sc.y) : 0.f;
        sca.z = f0 + 2 < D::H ? lin3u(l3, sc.z) : 0.f;
        sca.w = f0 + 3 < D::H ? lin3u(l3, sc.w) : 0.f;
        if (id.valid) {
            st_blk(scal, n, D::HP, t, id.lane, sca);
            st_blk(vdot, n, D::HP, t, id.lane, vd);
            st_blk(v2buf, (size_t)n * 3 + 0, D::HP, t, id.lane, v2[0]);
            st_blk(v2buf, (size_t)n * 3 + 1, D::HP, t, id.lane, v2[1]);
            st_blk(v2buf, (size_t)n * 3 + 2, D::HP, t, id.lane, v2[2]);
        }
    }
}
