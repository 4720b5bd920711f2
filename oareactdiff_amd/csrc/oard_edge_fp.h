// oard_edge_fp.h — the GCL edge kernel without workgroup barriers ("flag pipeline").
//
// k_gcl_edge_v1 synchronises its 8 waves with one s_barrier per phase.  The round-2 study (profiles/round2_gcl_phase_study.txt)
// shows what that costs: the two waves of a SIMD pair are released together, reach their chain starts, epilogues and the next
// barrier together, so their stalls coincide - 700-1 100 cycles per phase with no MFMA in flight - and the older wave of every
// pair waits ~23 % of each phase for the younger one.  Here the weight stream goes through a ring of RING = 4 slabs, fetched two
// phases ahead, and the waves synchronise through two counters per slab in LDS instead:
//   ready[b]  += 1 by every issuing wave (0..3) once its LDS-DMA pieces of the slab in buffer b have landed (s_waitcnt vmcnt(0));
//   done[b]   += 1 by every wave once it has finished reading the slab in buffer b.
// A wave starts phase r when ready[r % RING] has reached IW * (r / RING + 1); an issuing wave starts refilling buffer (r+2) % RING
// during phase r when done[(r+2) % RING] has reached WAVES * ((r+2) / RING), i.e. when every wave has finished phase r + 2 - RING.
// Nothing else couples the waves: waves 0..3 (served first by the SIMD arbiter) run up to two phases ahead of waves 4..7, so the
// phase transitions of the two halves no longer coincide and each half's bubbles are filled by the other half's MFMAs.
// No wait can deadlock: ready[] of a phase only depends on waves 0..3 having issued it, which only depends on done[] of a phase
// RING - 2 = 2 phases older than the one they are in, which the slower waves reach without waiting for anything newer.
// Every wait is also bounded (FP_SPIN_MAX polls): on a timeout the wave goes on (results are then wrong) and counts it in
// g_fp_timeouts, which the host checks (oard_debug_fp_timeouts) - a protocol bug must never hang the GPU.
#pragma once
#include "oard_edge_v1.h"

OARD_DEVVAR __device__ unsigned int g_fp_timeouts;
#define FP_SPIN_MAX (1 << 16)
#ifndef FP_GATE_EVERY
#define FP_GATE_EVERY 4
#endif

OARD_DEV unsigned fp_lds_addr(const void* p) { return (unsigned)(size_t)(lds_ptr_t)p; }
// poll a counter in LDS until it reaches `target` (wave-uniform); ~64-cycle naps between polls
OARD_DEV void fp_wait(const int* flag, int target) {
    const unsigned a = fp_lds_addr(flag);
    for (int spin = 0; spin < FP_SPIN_MAX; ++spin) {
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        if (__builtin_amdgcn_readfirstlane(v) >= target) return;
        __builtin_amdgcn_s_sleep(1);
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_fp_timeouts, 1u);
}
OARD_DEV bool fp_test(const int* flag, int target) {
    const unsigned a = fp_lds_addr(flag);
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return __builtin_amdgcn_readfirstlane(v) >= target;
}
// every LDS read of this wave has returned / every LDS-DMA piece of this wave has landed -> bump the counter (one lane)
OARD_DEV void fp_signal_reads_done(int* flag) {
    const unsigned a = fp_lds_addr(flag);
    if ((threadIdx.x & 63) == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\tds_add_u32 %0, %1" :: "v"(a), "v"(1) : "memory");
}
OARD_DEV void fp_signal_dma_landed(int* flag) {
    const unsigned a = fp_lds_addr(flag);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(a), "v"(1) : "memory");
}

// LDS-DMA prefetcher of the ring: like SlabPrefetch (waves 0..IW-1 issue, one piece per tick), plus the gate on done[]
template <int WAVES, int SLAB, int RING>
struct RingPrefetch {
    static constexpr int IW = WAVES >= 8 ? WAVES / 2 : WAVES;
    static constexpr int KMAX = (SLAB + IW - 1) / IW;
    const float* src;       // wave-uniform
    float* dst;
    unsigned lane_off;
    const int* gate;        // done[] counter of the buffer being refilled
    int gate_target;
    bool open;              // gate already seen open in this phase
    int n, k, wave, ticks;
    OARD_DEV void begin(const float* stream, float* smem, int slot, int first_chunk, int n_chunks, const int* gate_flag, int target) {
        src = stream + (size_t)first_chunk * 256;
        dst = smem + (size_t)slot * SLAB * 256;
        n = wave < IW ? n_chunks : 0; k = 0; ticks = 0;
        gate = gate_flag; gate_target = target; open = target <= 0 || n == 0;
    }
    OARD_DEV void one() {
        const int j = wave + k * IW;
        if (j < n) glds16u(src + (size_t)j * 256, lane_off, dst + j * 256);
        ++k;
    }
    OARD_DEV void tick() {                       // between MFMA pairs: never blocks (a closed gate is tried again every FP_GATE_EVERY ticks)
        if (k >= KMAX) return;
        if (!open && (ticks++ % FP_GATE_EVERY) == 0) open = fp_test(gate, gate_target);
        if (open) one();
    }
    OARD_DEV void flush() {                      // end of the phase: everything that is left, waiting for the gate if need be
        if (k >= KMAX) return;
        if (!open) { fp_wait(gate, gate_target); open = true; }
        while (k < KMAX) one();
    }
};

template <class D, int GP, int RING>
struct GclRing {
    using S = GclStream<D, GP>;
    static constexpr size_t LDS_BYTES = (size_t)RING * S::SLAB * 1024 + 64 * sizeof(int);
};

// Same arithmetic, same stream, same column / register layout as k_gcl_edge_v1 (see there); only the synchronisation differs.
template <class D, int WAVES, int GP, bool DO_S1, bool DO_S3, bool TRAIN, int RING = 4>
__global__ __launch_bounds__(WAVES * 64, 2) void k_gcl_edge_fp(TopoDev tp, const float* __restrict__ stream,
                                                               const float* __restrict__ P, const float* __restrict__ Q,
                                                               const float* __restrict__ u0, const float* __restrict__ c0,
                                                               long long r0, long long r1, const float* ew_in, float* ew_out,
                                                               float* __restrict__ mbuf, GclTape tape) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using S = GclStream<D, GP>;
    using PF = RingPrefetch<WAVES, S::SLAB, RING>;
    constexpr int HT = D::HT, WB = D::WB, G1 = S::G1, G2 = S::G2;
    constexpr bool TAIL1 = S::TAIL1, ROWS4 = S::ROWS4;
    constexpr int IW = PF::IW;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int* ready = reinterpret_cast<int*>(smem + (size_t)RING * S::SLAB * 256);      // [RING]
    int* done = ready + RING;                                                      // [RING]

    TL_DECL
    TL(0);
    PF pf;
    pf.wave = wave;
    pf.lane_off = (unsigned)lane * 16u;              // LDS-DMA source = uniform chunk address (SGPR pair) + this 32-bit lane offset
    const int p_first = DO_S1 ? 0 : S::NP1;
    // r = running phase index of this launch (0 = first phase executed), p = p_first + r = position in the stream
    auto chunks_of = [&](int p, int& start, int& n) {
        start = 0; n = 0;
        if (p < S::NP1) { start = p * GP * G1; n = min(GP, WB - p * GP) * G1; }
        else if (p < S::NP1 + S::NP2) { const int q = p - S::NP1; start = S::C1 + q * GP * G2; n = min(GP, S::NG2 - q * GP) * G2; }
        else if (p < S::NPH && DO_S3) { const int q = p - S::NP1 - S::NP2; start = S::C1 + S::C2 + q * GP * G2; n = min(GP, WB - q * GP) * G2; }
    };
    auto pf_begin = [&](int r) {                               // r = phase whose slab is to be fetched
        int start, n;
        chunks_of(p_first + r, start, n);
        const int b = r % RING;
        pf.begin(stream, smem, b, start, n, done + b, WAVES * (r / RING));
    };
    auto hook = [&]() { pf.tick(); };
    auto SL = [&](int r) -> const float* { return smem + (size_t)(r % RING) * S::SLAB * 256 + lane * 4; };
    auto A = [&](int r, int j) -> f4 { return *reinterpret_cast<const f4*>(SL(r) + (size_t)j * 256); };
    auto phase_enter = [&](int r) { fp_wait(ready + r % RING, IW * (r / RING + 1)); };
    auto phase_leave = [&](int r) {                            // after the last LDS read of phase r and the last DMA issue for r + 2
        pf.flush();
        fp_signal_reads_done(done + r % RING);
        if (wave < IW) fp_signal_dma_landed(ready + (r + 2) % RING);
    };

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
    f4 xn[GP];
#pragma unroll
    for (int gg = 0; gg < GP; ++gg) xn[gg] = (DO_S1 && gg < WB) ? ld_edge(erow + 16 * gg) : f4zero();

    // prologue: counters to zero, slabs of the first two phases, ONE workgroup barrier
    if (threadIdx.x < 2 * RING) ready[threadIdx.x] = 0;
    __syncthreads();
    pf_begin(0); pf.flush();
    pf_begin(1); pf.flush();
    if (wave < IW) { fp_signal_dma_landed(ready + 0); if ((threadIdx.x & 63) == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(fp_lds_addr(ready + 1)), "v"(1) : "memory"); }

    int r = 0;
    // ---- S1 ---------------------------------------------------------------------------------------------------------------
    f4 h1x = f4zero();
    for (int p1 = 0; DO_S1 && p1 < S::NP1; ++p1, ++r) {
        phase_enter(r);
        TL(1);
        f4 x[GP];
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) x[gg] = xn[gg];
        pf_begin(r + 2);
        if (p1 + 1 < S::NP1) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) {
                const int b = (p1 + 1) * GP + gg;
                if (b < WB) xn[gg] = ld_edge(erow + 16 * b);
            }
        }
#pragma unroll
        for (int gg = 0; gg < GP; ++gg)
            if (p1 * GP + gg < WB) { TL(2); chain_kouter<HT, ROWS4>(SL(r), gg * G1, x[gg], h1, h1x, hook); TL(3); }
        TL(4);
        phase_leave(r);
        TL(5);
    }
    if (ROWS4 && DO_S1) {
        const f4 v = reduce_g(h1[HT - 1] + h1x);
        h1[HT - 1] = g == 0 ? v : f4zero();
    }
    if (TRAIN) {
#pragma unroll
        for (int t = 0; t < HT; ++t) st_blk(tape.z1, e, D::HP, t, lane, h1[t]);
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) h1[t] = silu4(h1[t]);
    const float h1_tail = TAIL1 ? tail_compact(h1[HT - 1], lane) : 0.f;

    // ---- S2 ---------------------------------------------------------------------------------------------------------------
    f4 m[HT];
    f4 on[GP];
    float m_tail = 0.f;
#pragma unroll
    for (int p2 = 0; p2 < S::NP2; ++p2, ++r) {
        phase_enter(r);
        TL(1);
        pf_begin(r + 2);
        if (DO_S3 && p2 == S::NP2 - 1) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg)
                on[gg] = gg < WB ? (DO_S1 ? ld_edge(erow + 16 * gg) : ld_f4(c0 + 16 * gg + 4 * g)) : f4zero();
        }
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int tg = p2 * GP + gg;
            if (tg < S::NG2) {
                const f4 bias = A(r, gg * G2);
                f4 acc;
                TL(2);
                if (ROWS4 && tg >= HT - 1) {
                    acc = reduce_g(tg < HT ? chain_tile4<HT>(SL(r), gg * G2 + 1, h1, bias, hook)
                                           : chain_tile4<HT>(SL(r), gg * G2 + 1, m, bias, hook));
                    if (tg < HT && g != 0) acc = f4zero();
                } else {
                    acc = tg < HT ? chain_tile<HT, TAIL1>(SL(r), gg * G2 + 1, h1, bias, h1_tail, hook)
                                  : chain_tile<HT, TAIL1>(SL(r), gg * G2 + 1, m, bias, m_tail, hook);
                }
                TL(3);
                if (tg < HT) {
                    if (TRAIN) st_blk(tape.z2, e, D::HP, tg, lane, acc);
                    m[tg] = silu4(acc);
                    if (TAIL1 && tg == HT - 1) m_tail = tail_compact(m[HT - 1], lane);
                } else {
                    const float a = ROWS4 ? acc.x : __shfl(acc.x, lane & 15, 64);
                    if (TRAIN && g == 0) tape.att[e] = a;
                    const float gate = silu1(a);
#pragma unroll
                    for (int t = 0; t < HT; ++t) m[t] *= gate;
                    m_tail *= gate;
                }
            }
        }
        TL(4);
        phase_leave(r);
        TL(5);
    }
    // ---- S3 ---------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < HT; ++t) st_blk(mbuf, eid, D::HP, t, lane, m[t]);
    if (!DO_S3) { TL_END(); return; }
    f4 pend[GP], pendz[TRAIN ? GP : 1];
    for (int p3 = 0; p3 < S::NP3; ++p3, ++r) {
        phase_enter(r);
        TL(1);
        if (p3 > 0) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) {
                st_f4(orow + 16 * ((p3 - 1) * GP + gg), pend[gg]);
                if (TRAIN) st_f4(tape.z3 + e * D::WP + 4 * g + 16 * ((p3 - 1) * GP + gg), pendz[gg]);
            }
        }
        f4 o[GP];
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) o[gg] = on[gg];
        pf_begin(r + 2);
        if (p3 + 1 < S::NP3) {
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) {
                const int t = (p3 + 1) * GP + gg;
                if (t < WB) on[gg] = DO_S1 ? ld_edge(erow + 16 * t) : ld_f4(c0 + 16 * t + 4 * g);
            }
        }
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int t = p3 * GP + gg;
            if (t < WB) {
                TL(2);
                const f4 z = chain_tile<HT, TAIL1>(SL(r), gg * G2 + 1, m, A(r, gg * G2), m_tail, hook);
                TL(3);
                if (TRAIN) pendz[gg] = z;
                pend[gg] = o[gg] + silu4(z);
            }
        }
        TL(4);
        phase_leave(r);
        TL(5);
    }
#pragma unroll
    for (int gg = 0; gg < GP; ++gg) {
        const int t = (S::NP3 - 1) * GP + gg;
        if (t < WB) {
            st_f4(orow + 16 * t, pend[gg]);
            if (TRAIN) st_f4(tape.z3 + e * D::WP + 4 * g + 16 * t, pendz[gg]);
        }
    }
    TL_END();
}
