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
// timing-only ablations of an experiment build (results are garbage): OARD_ABL_NODMA_INSTR drops the LDS-DMA instruction and keeps
// the issue logic around it; OARD_ABL_BURST issues a phase's pieces right after the barrier instead of between MFMA pairs
// the same with the source given as a wave-uniform chunk address plus a 32-bit per-lane byte offset: hipcc selects the SGPR-base
// form of the instruction and the address costs no VALU work (a v_lshl_add_u64 per piece otherwise - and every VALU instruction
// takes ~6 cycles of the SIMD's MFMA issue time, tools/micro/mfma_valu.hip)
OARD_DEV void glds16(const float* gsrc_lane, float* lds_chunk);
OARD_DEV void glds16u(const float* chunk_uniform, unsigned lane_byte_off, float* lds_chunk) {
    glds16(reinterpret_cast<const float*>(reinterpret_cast<const char*>(chunk_uniform) + lane_byte_off), lds_chunk);
}
OARD_DEV void glds16(const float* gsrc_lane, float* lds_chunk) {
#ifndef OARD_ABL_NODMA_INSTR
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)gsrc_lane, (lds_ptr_t)lds_chunk, 16, 0, 0);
#else
    asm volatile("" :: "v"(gsrc_lane), "s"(lds_chunk));
#endif
}

// Phase barrier: every LDS-DMA piece this wave issued must have landed before any wave reads the slab.
// hipcc's own "vmcnt(0) before the workgroup barrier" is NOT emitted once the DMA issues sit in
// data-dependent control flow inside a loop (observed: only lgkmcnt(0) before s_barrier), so the wait
// is stated explicitly.
OARD_DEV void phase_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef OARD_ABL_NOBARRIER   // timing-only ablation (experiment build)
    __syncthreads();
#endif
}

// Experiment build only (-DOARD_PHASE_PROBE): where the waves of k_gcl_edge_v1 spend their cycles - the s_waitcnt in front of
// the phase barrier (own DMA pieces / loads / stores not landed), the barrier itself (waiting for the other waves), the issue of
// one LDS-DMA piece.  Sums over all waves in g_phase_probe (oard_debug_probe_read).
#ifdef OARD_PHASE_PROBE
OARD_DEVVAR __device__ unsigned long long g_phase_probe[8];
#define PROBE_DECL long long pr_t0_ = clock64(), pr_wait_ = 0, pr_bar_ = 0, pr_n_ = 0;
#define PHASE_BARRIER() do { const long long a_ = clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const long long b_ = clock64(); \
        __syncthreads(); const long long c_ = clock64(); pr_wait_ += b_ - a_; pr_bar_ += c_ - b_; ++pr_n_; } while (0)
#define PROBE_END(pf) do { if ((threadIdx.x & 63) == 0) { atomicAdd(&g_phase_probe[0], 1ull); atomicAdd(&g_phase_probe[1], (unsigned long long)(clock64() - pr_t0_)); \
        atomicAdd(&g_phase_probe[2], (unsigned long long)pr_wait_); atomicAdd(&g_phase_probe[3], (unsigned long long)pr_bar_); \
        atomicAdd(&g_phase_probe[4], (unsigned long long)(pf).dma_cyc); atomicAdd(&g_phase_probe[5], (unsigned long long)(pf).dma_n); \
        atomicAdd(&g_phase_probe[6], (unsigned long long)pr_n_); if ((threadIdx.x >> 6) < (blockDim.x >> 7)) atomicAdd(&g_phase_probe[7], (unsigned long long)pr_bar_); } } while (0)
#else
#define PROBE_DECL
#define PHASE_BARRIER() phase_barrier()
#define PROBE_END(pf) do {} while (0)
#endif

// experiment build (-DOARD_TIMELINE): timestamps of one workgroup's waves at the phase barriers and around every MFMA chain
// (tools/wave_timeline.py): code 1 = after the phase barrier, 2 = chain starts, 3 = chain done, 4 = phase work done
#ifdef OARD_TIMELINE
#define TL_MAX 1024
OARD_DEVVAR __device__ long long g_timeline[16][TL_MAX];
#define TL_DECL const bool tl_on_ = DO_S1 && DO_S3 && blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0; int tl_n_ = 0;
#define TL(code) do { if (tl_on_ && tl_n_ < TL_MAX) g_timeline[threadIdx.x >> 6][tl_n_++] = (clock64() << 3) | (code); } while (0)
#define TL_END() do { if (tl_on_ && tl_n_ < TL_MAX) g_timeline[threadIdx.x >> 6][tl_n_] = 0; } while (0)
#else
#define TL_DECL
#define TL(code) do {} while (0)
#define TL_END() do {} while (0)
#endif

// two (a, b, acc) triples interleaved: consecutive MFMAs never depend on each other
// (v_mfma_f32_16x16x4_f32: 32-cycle issue, 40-cycle dependent latency)
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

// ---- 4-row output tiles on v_mfma_f32_4x4x1_16b_f32 ------------------------------------------------------------------------
// A tile with at most 4 real output rows (the 13th tile of a 196-wide layer, the 1-row gate) wastes 3/4 of a 16x16x4 MFMA.  The
// 4x4x1 instruction computes 16 independent 4x4 blocks (lane l: block l/4, A row = B column = l%4, D register = row; 2 passes,
// measured 8.7-10.5 cycles, tools/micro/mfma4x4.hip).  With the column engine's layout, lane (g, e) = block (k-slice g, column
// group e/4): component s of a B block feeds feature 16b+4g+s of column e, so ONE chunk (4 instructions) covers the 16 features
// of a block for 4 output rows, each k-slice g accumulating its own partial sum; reduce_g() adds the four slices at the end.
// The A chunk ("rows4" packing) holds W[row0 + e%4][16b+4g+s] in component s of lane (g, e).
OARD_DEV void mma4_chunk(f4 a, f4 b, f4& c0, f4& c1) {
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, b.x, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, b.y, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, b.z, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, b.w, c1, 0, 0, 0);
}
OARD_DEV f4 reduce_g(f4 v) {            // sum over the four k-slices (lanes e, 16+e, 32+e, 48+e); every lane gets the total
    f4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float x = v[c];
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        r[c] = x;
    }
    return r;
}

// ---- software-pipelined LDS -> MFMA chains ---------------------------------------------------------
// `sl` = this lane's pointer into the current slab (slab base + lane*4); chunk j is at sl + j*256.
// The A fragments of the NEXT pair of chunks are read from LDS while the current pair's 8 MFMAs issue,
// so the ds_read latency (~100+ cycles) is covered.
OARD_DEV f4 lds_a(const float* sl, int j) { return *reinterpret_cast<const f4*>(sl + j * 256); }

// (An inline-asm variant that pins the ds_read_b128 of the next pair in front of the current pair's MFMAs - hipcc sinks them behind
// the 7th MFMA to reuse the fragment registers - was measured in round 2: no change, profiles/round2_gcl_phase_study.txt.)
// M-outer: one output tile = sum over KB chunks (slots j0..j0+KB-1) x in[b]; even/odd accumulators.
// TAIL (KB odd): the last chunk is a compact K tail - one k-step, A fragment in component x, B operand `tail` (see k_gcl_edge_v1).
struct NoHook { OARD_DEV void operator()() const {} };
template <int KB, bool TAIL, class Hook>
OARD_DEV f4 chain_tile(const float* sl, int j0, const f4 (&in)[KB], f4 init, float tail, Hook hook) {
    static_assert(!TAIL || (KB & 1), "compact tail needs an odd number of chunks");
    f4 c0 = init, c1 = f4zero();
    f4 a0 = lds_a(sl, j0), a1 = KB > 1 ? lds_a(sl, j0 + 1) : f4zero();
#pragma unroll
    for (int b = 0; b + 1 < KB; b += 2) {
        f4 n0 = a0, n1 = a1;
        if (b + 2 < KB) n0 = lds_a(sl, j0 + b + 2);
        if (b + 3 < KB) n1 = lds_a(sl, j0 + b + 3);
        mma_pair(a0, in[b], c0, a1, in[b + 1], c1);
        hook();
        a0 = n0; a1 = n1;
    }
    if (KB & 1) {
        if (TAIL) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, tail, c0, 0, 0, 0);
        else c0 = mma_chunk(a0, in[KB - 1], c0);
    }
    return c0 + c1;
}
template <int KB, class Hook = NoHook>
OARD_DEV f4 chain_tile(const float* sl, int j0, const f4 (&in)[KB], f4 init, Hook hook = Hook()) {
    return chain_tile<KB, false, Hook>(sl, j0, in, init, 0.f, hook);
}
// the same tile on 4x4x1 MFMAs (all KB chunks in rows4 packing): returns the UNREDUCED per-k-slice sums (apply reduce_g)
template <int KB, class Hook = NoHook>
OARD_DEV f4 chain_tile4(const float* sl, int j0, const f4 (&in)[KB], f4 init, Hook hook = Hook()) {
    f4 c0 = init, c1 = f4zero();
    f4 a0 = lds_a(sl, j0), a1 = KB > 1 ? lds_a(sl, j0 + 1) : f4zero();
#pragma unroll
    for (int b = 0; b + 1 < KB; b += 2) {
        f4 n0 = a0, n1 = a1;
        if (b + 2 < KB) n0 = lds_a(sl, j0 + b + 2);
        if (b + 3 < KB) n1 = lds_a(sl, j0 + b + 3);
        mma4_chunk(a0, in[b], c0, c1);
        mma4_chunk(a1, in[b + 1], c0, c1);
        hook();
        a0 = n0; a1 = n1;
        __builtin_amdgcn_sched_barrier(0);          // 4x4x1 MFMAs are short: left alone, hipcc hoists the LDS reads of many pairs (spills)
    }
    if (KB & 1) mma4_chunk(a0, in[KB - 1], c0, c1);
    return c0 + c1;
}
// K-outer: acc[t] += chunk(j0 + t) x x for t < MT, pairs of tiles interleaved.
// ROWS4 (MT odd): the last tile has at most 4 real rows and runs on 4x4x1 MFMAs into acc[MT-1] / extra (unreduced k-slice sums)
template <int MT, bool ROWS4, class Hook>
OARD_DEV void chain_kouter(const float* sl, int j0, f4 x, f4 (&acc)[MT], f4& extra, Hook hook) {
    static_assert(!ROWS4 || (MT & 1), "the 4-row tile must be the odd one");
    f4 a0 = lds_a(sl, j0), a1 = MT > 1 ? lds_a(sl, j0 + 1) : f4zero();
#pragma unroll
    for (int t = 0; t + 1 < MT; t += 2) {
        f4 n0 = a0, n1 = a1;
        if (t + 2 < MT) n0 = lds_a(sl, j0 + t + 2);
        if (t + 3 < MT) n1 = lds_a(sl, j0 + t + 3);
        mma_pair(a0, x, acc[t], a1, x, acc[t + 1]);
        hook();
        a0 = n0; a1 = n1;
    }
    if (MT & 1) {
        if (ROWS4) mma4_chunk(a0, x, acc[MT - 1], extra);
        else acc[MT - 1] = mma_chunk(a0, x, acc[MT - 1]);
    }
}
template <int MT, class Hook = NoHook>
OARD_DEV void chain_kouter(const float* sl, int j0, f4 x, f4 (&acc)[MT], Hook hook = Hook()) {
    f4 unused = f4zero();
    chain_kouter<MT, false, Hook>(sl, j0, x, acc, unused, hook);
}

// The LDS-DMA prefetcher shared by the streamed kernels: the pieces of the NEXT phase's slab are issued one at a time between
// MFMA pairs (a burst after the barrier stalls the chains: +7 % GCL, +37 % Equi time).  In an 8-wave workgroup only waves 0..3
// issue (OARD_PF_HALF), one piece after every pair (OARD_PF_PERIOD): the SIMD arbiter favours the older wave of a pair, so waves
// 4..7 are the critical path of every phase (probe build: 93 % of the barrier wait is spent by waves 0..3) - they should neither
// pay the issue cost nor reach the barrier with a piece still in flight.  Measured (profiles/round2_gcl_phase_study.txt): all
// waves / every third pair 9.82 ms, waves 0..3 / every pair 9.63 ms, waves 0..3 / every second pair 10.07 ms per step.
#ifndef OARD_PF_HALF
#define OARD_PF_HALF 1
#endif
#ifndef OARD_PF_PERIOD
#define OARD_PF_PERIOD 1
#endif
// HALF / PERIOD are per kernel: the EquiMessage kernel (37 live accumulators) spills 330 bytes per lane with the GCL kernel's policy
// and runs equally fast with all waves issuing every third pair, so it keeps that.
template <int WAVES, int SLAB, int HALF = OARD_PF_HALF, int PERIOD = OARD_PF_PERIOD, int RING = 2>
struct SlabPrefetch {
    static constexpr int IW = (HALF && WAVES >= 8) ? WAVES / 2 : WAVES;               // issuing waves
    static constexpr int KMAX = (SLAB + IW - 1) / IW;          // pieces per issuing wave per phase (upper bound)
    const float* src;       // wave-uniform
    float* dst;
    unsigned lane_off;
    int n, k, next, wave;
#ifdef OARD_PHASE_PROBE
    long long dma_cyc = 0, dma_n = 0;
#endif
    OARD_DEV void begin(const float* stream, float* smem, int phase, int first_chunk, int n_chunks) {
        src = stream + (size_t)first_chunk * 256;
        dst = smem + (size_t)(RING == 2 ? (phase & 1) : (phase % RING)) * SLAB * 256;
        n = wave < IW ? n_chunks : 0; k = 0; next = 1 + ((IW == WAVES && wave >= WAVES / 2) ? 1 : 0);
    }
    OARD_DEV void one() {
#ifdef OARD_ABL_NOHOOK       // timing-only ablation (experiment build): no LDS-DMA, no issue logic
        k = KMAX;
        return;
#endif
        const int j = wave + k * IW;
#ifdef OARD_PHASE_PROBE
        if (j < n) { const long long a_ = clock64(); glds16u(src + (size_t)j * 256, lane_off, dst + j * 256); dma_cyc += clock64() - a_; ++dma_n; }
#else
        if (j < n) glds16u(src + (size_t)j * 256, lane_off, dst + j * 256);
#endif
        ++k;
    }
#ifdef OARD_ABL_BURST
    OARD_DEV void tick() {}
#else
#ifdef OARD_ABL_NOHOOK
    OARD_DEV void tick() {}
#else
    OARD_DEV void tick() { if (--next == 0) { one(); next = PERIOD; } }
#endif
#endif
    OARD_DEV void flush() { while (k < KMAX) one(); }
};

// =====================================================================================================
// GCLMessage edge part, all edges.  Stream (chunks): S1 = WB groups x HT  [W1c, K-outer];
// S2 = (HT+1) groups x (1+HT)  [bias b2 | W2 tile t] + gate group [batt | watt as a 1-row tile];
// S3 = WB groups x (1+HT)  [bias b3 | W3 tile t].
// =====================================================================================================
// edge-state block load of the GCL kernel (its own name so that a timing experiment can stub it out: -DOARD_ABL_NOEDGELOAD)
#ifdef OARD_ABL_NOEDGELOAD
OARD_DEV f4 ld_edge(const float* p) { return (f4){0.25f, -0.5f, 0.125f, 1.0f}; }
#else
OARD_DEV f4 ld_edge(const float* p) { return ld_f4(p); }
#endif

template <class D, int GP>
struct GclStream {
    static constexpr int HT = D::HT, WB = D::WB, G1 = HT, G2 = HT + 1, NG2 = HT + 1;
    static constexpr bool TAIL1 = (D::H % 16) >= 1 && (D::H % 16) <= 4 && HT >= 3 && (HT & 1);      // compact K tail (see the kernel)
    static constexpr bool ROWS4 = TAIL1;                       // the 13th output tile of W1c / W2 and the gate on 4x4x1 MFMAs
    static constexpr int SLAB = GP * G2;                       // chunks per slab
    static constexpr int NP1 = (WB + GP - 1) / GP, NP2 = (NG2 + GP - 1) / GP, NP3 = NP1, NPH = NP1 + NP2 + NP3;
    static constexpr int C1 = WB * G1, C2 = NG2 * G2, C3 = WB * G2, CHUNKS = C1 + C2 + C3;
    static constexpr size_t LDS_BYTES = (size_t)2 * SLAB * 1024;
};

// What a training-mode forward keeps per layer for the backward pass (all NULL in inference): the
// pre-activations of the three Linear layers and of the gate, in physical row order like the edge state.
struct GclTape {
    float* z1;     // [E+1][HP]  W1c.ew + P[src] + Q[tgt]
    float* z2;     // [E+1][HP]  W2.h1 + b2
    float* att;    // [E+1]      watt.m0 + batt
    float* z3;     // [E+1][WP]  W3.m + b3
};

// Held to <= 256 registers (2 waves per SIMD).  Besides occupancy this keeps the accumulators out of the
// AGPR half of the file: with the default bound hipcc (ROCm 7.2) put some S1 accumulators in AGPRs for the
// 4-wave shape and the result drifted to 6e-5 of the oracle; with the bound every shape is at 2.5e-7.
// DO_S1 = false: the columns are inter-object edges of layer 0, whose state is the constant row: stage S1
//   collapses to the precomputed vector u0 (added to the accumulator init) and its DMA is skipped.
// DO_S3 = false: the columns are inter-object edges of the last layer, whose updated state nobody reads
//   (EquiMessage only touches inner edges): the residual stage and its edge-state traffic are skipped.
// TRAIN: the new state goes to `ew_out` (a different buffer: the backward pass needs every layer's input state)
//   and the pre-activations are stored (GclTape).  In inference ew_out == ew_in (in-place update).
// Columns are the physical rows [r0, r1).
// Compact K tail (GclStream::TAIL1, H % 16 in 1..4, e.g. H = 196): the last 16-feature block of h1 / m holds at most 4 real
//   features, which the block layout spreads over all four k-steps (one per step, lanes g = 0).  tail_compact() gathers them into
//   ONE k-step (lane (g, e) <- component g of lane (0, e)), the stream carries the matching A fragment in component x of the
//   last chunk of every W2 / watt / W3 tile (k_pack_matrix, tail_compact), and the chains end with 1 MFMA instead of 4:
//   49 instead of 52 MFMAs per tile of S2 and S3 (-3.3 % of the kernel's MFMAs).
OARD_DEV float tail_compact(f4 v, int lane) {
    const int e = lane & 15, g = lane >> 4;
    const float t0 = __shfl(v.x, e, 64), t1 = __shfl(v.y, e, 64), t2 = __shfl(v.z, e, 64), t3 = __shfl(v.w, e, 64);
    return g == 0 ? t0 : (g == 1 ? t1 : (g == 2 ? t2 : t3));
}
// RING = 3 (three slabs, the DMA runs TWO phases ahead): the slab of phase p + 1 has landed and been published by the barrier of
//   phase p, so a phase starts computing without a barrier; the barrier of phase p (which only has to separate the last reads of
//   phase p - 1 from the DMA writes of phase p + 2, and to publish phase p + 1) sits behind the OARD_BAR_AT-th MFMA pair of the phase,
//   where the next A fragments are already requested - no pipeline refill behind it.  Measured (B = 64, isolated launches, per
//   step): 9.15 -> 8.99 ms with the barrier behind pair 2; behind pair 1 / 4 / 6 / 9: 9.17 / 9.10 / 9.13 / 9.18 ms.
#ifndef OARD_BAR_AT
#define OARD_BAR_AT 2
#endif
template <class D, int WAVES, int GP, bool DO_S1, bool DO_S3, bool TRAIN, int MINW = 2, int RING = 2>
__global__ __launch_bounds__(WAVES * 64, MINW) void k_gcl_edge_v1(TopoDev tp, const float* __restrict__ stream,
                                                               const float* __restrict__ P, const float* __restrict__ Q,
                                                               const float* __restrict__ u0, const float* __restrict__ c0,
                                                               long long r0, long long r1, const float* ew_in, float* ew_out,
                                                               float* __restrict__ mbuf, GclTape tape) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using S = GclStream<D, GP>;
    constexpr int HT = D::HT, WB = D::WB, G1 = S::G1, G2 = S::G2;
    constexpr bool TAIL1 = S::TAIL1, ROWS4 = S::ROWS4;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // provably wave-uniform
    PROBE_DECL
    TL_DECL
    TL(0);
    static_assert(RING == 2 || RING == 3, "two slabs (barrier at the phase start) or three (barrier inside the phase)");
    static_assert(RING == 2 || HT >= 2, "the phase barrier must sit inside the FIRST chain of a phase (S3 stores the previous phase's results behind it)");
    constexpr int DIST = RING - 1;                             // phases the DMA runs ahead
    constexpr int BAR_AT = OARD_BAR_AT < HT / 2 ? OARD_BAR_AT : HT / 2;       // a chain over HT chunks has HT / 2 hook calls
    SlabPrefetch<WAVES, S::SLAB, OARD_PF_HALF, OARD_PF_PERIOD, RING> pf;
    pf.wave = wave;
    pf.lane_off = (unsigned)lane * 16u;              // LDS-DMA source = uniform chunk address (SGPR pair) + this 32-bit lane offset
    auto pf_begin = [&](int p) {                               // p = phase to prefetch
        int start = 0, n = 0;
        if (p < S::NP1) { start = p * GP * G1; n = min(GP, WB - p * GP) * G1; }
        else if (p < S::NP1 + S::NP2) { const int q = p - S::NP1; start = S::C1 + q * GP * G2; n = min(GP, S::NG2 - q * GP) * G2; }
        else if (p < S::NPH && DO_S3) { const int q = p - S::NP1 - S::NP2; start = S::C1 + S::C2 + q * GP * G2; n = min(GP, WB - q * GP) * G2; }
        pf.begin(stream, smem, p, start, n);
#ifdef OARD_ABL_BURST
        pf.flush();
#endif
    };
    // phase protocol: every phase has ONE barrier followed by its post-barrier work `post` (DMA issue for a later phase, edge-state
    // prefetch, late stores).  RING 2: at the phase start.  RING 3: armed at the phase start (bar_left), run by the OARD_BAR_AT-th hook
    // call (hooks sit between the MFMA pairs of the chains), or after the chains if the phase had fewer hook calls than that.
    int bar_left = 0;
    auto slab_of = [&](int p) -> int { return RING == 2 ? (p & 1) : (p % RING); };
    auto A = [&](int p, int j) -> f4 {
        return *reinterpret_cast<const f4*>(smem + ((size_t)slab_of(p) * S::SLAB + j) * 256 + lane * 4);
    };
    auto SL = [&](int p) -> const float* { return smem + (size_t)slab_of(p) * S::SLAB * 256 + lane * 4; };

    // column of this lane; padding columns work on the spare row E of every per-edge buffer (allocated for that
    // purpose), so the kernel has no validity branches and every wave stays in the barrier protocol
    const long long c = r0 + ((long long)blockIdx.x * WAVES + wave) * 16 + (lane & 15);
    const size_t e = (size_t)(c < r1 ? c : tp.E);
    const float* erow = ew_in + e * D::WP + 4 * g;
    float* orow = ew_out + e * D::WP + 4 * g;
    f4 h1[HT];
    const size_t eid = (size_t)tp.row_eid[e];                  // row of the message buffer (loaded here: its latency must not sit in S3)
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
    int p = DO_S1 ? 0 : S::NP1;
    pf_begin(p);
    pf.flush();                                                // burst form (prologue only)
    if (RING == 3) { pf_begin(p + 1); pf.flush(); PHASE_BARRIER(); }

    // ---- S1: h1 += W1c . ew   (K-outer) -------------------------------------------------------------
    f4 h1x = f4zero();                                         // ROWS4: second accumulator of the 4-row tile
    // The phase loops are unrolled by two with the prefetched edge-state blocks in ping-pong register sets: a single loop body
    // needs a copy "current = next" per block and phase, and every VALU instruction costs MFMA issue time (tools/micro/mfma_valu.hip).
    f4 xm[GP];
    auto s1_phase = [&](int p1, const f4 (&x)[GP], f4 (&xnext)[GP]) {
        auto post = [&]() {
            TL(1);
            pf_begin(p + DIST);
            if (p1 + 1 < S::NP1) {
#pragma unroll
                for (int gg = 0; gg < GP; ++gg) {
                    const int b = (p1 + 1) * GP + gg;
                    if (b < WB) xnext[gg] = ld_edge(erow + 16 * b);
                }
            }
        };
        auto hook = [&]() { if (RING == 3 && bar_left > 0 && --bar_left == 0) { PHASE_BARRIER(); post(); } pf.tick(); };
        if (RING == 2) { PHASE_BARRIER(); post(); } else bar_left = BAR_AT;
#pragma unroll
        for (int gg = 0; gg < GP; ++gg)
            if (p1 * GP + gg < WB) { TL(2); chain_kouter<HT, ROWS4>(SL(p), gg * G1, x[gg], h1, h1x, hook); TL(3); }
        TL(4);
        if (RING == 3 && bar_left > 0) { bar_left = 0; PHASE_BARRIER(); post(); }
        pf.flush();
        ++p;
    };
    if (DO_S1) {
        int p1 = 0;
        for (; p1 + 1 < S::NP1; p1 += 2) { s1_phase(p1, xn, xm); s1_phase(p1 + 1, xm, xn); }
        if (p1 < S::NP1) s1_phase(p1, xn, xm);
    }
    if (ROWS4 && DO_S1) {                                      // k-slice sums of the 4-row tile -> block layout (real rows in lanes g = 0)
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

    // ---- S2: m = SiLU(W2 h1 + b2); gate = SiLU(watt . m + batt) (the gate is the last group, fed with m) ----
    f4 m[HT];
    f4 on[GP];
    f4 pz2[TRAIN ? GP : 1];                         // TRAIN: z2 tiles of the previous phase, stored behind the next barrier
    float m_tail = 0.f;
#pragma unroll
    for (int p2 = 0; p2 < S::NP2; ++p2, ++p) {
        auto post = [&]() {
            TL(1);
            if (TRAIN && p2 > 0) {
#pragma unroll
                for (int gg = 0; gg < GP; ++gg)
                    if ((p2 - 1) * GP + gg < HT) st_blk(tape.z2, e, D::HP, (p2 - 1) * GP + gg, lane, pz2[gg]);
            }
            pf_begin(p + DIST);
            if (DO_S3 && p2 == S::NP2 - 1) {        // prefetch the old edge-state tiles of S3's first phase
#pragma unroll
                for (int gg = 0; gg < GP; ++gg)
                    on[gg] = gg < WB ? (DO_S1 ? ld_edge(erow + 16 * gg) : ld_f4(c0 + 16 * gg + 4 * g)) : f4zero();
            }
        };
        auto hook = [&]() { if (RING == 3 && bar_left > 0 && --bar_left == 0) { PHASE_BARRIER(); post(); } pf.tick(); };
        if (RING == 2) { PHASE_BARRIER(); post(); } else bar_left = BAR_AT;
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int tg = p2 * GP + gg;            // compile-time after unrolling
            if (tg < S::NG2) {
                const f4 bias = A(p, gg * G2);
                TL(2);
                f4 acc;
                if (ROWS4 && tg >= HT - 1) {                   // 13th tile of W2 / the gate: 4 real rows on 4x4x1 MFMAs
                    acc = reduce_g(tg < HT ? chain_tile4<HT>(SL(p), gg * G2 + 1, h1, bias, hook)
                                           : chain_tile4<HT>(SL(p), gg * G2 + 1, m, bias, hook));
                    if (tg < HT && g != 0) acc = f4zero();
                } else {
                    acc = tg < HT ? chain_tile<HT, TAIL1>(SL(p), gg * G2 + 1, h1, bias, h1_tail, hook)
                                  : chain_tile<HT, TAIL1>(SL(p), gg * G2 + 1, m, bias, m_tail, hook);
                }
                TL(3);
                if (tg < HT) {
                    if (TRAIN) pz2[gg] = acc;
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
        if (RING == 3 && bar_left > 0) { bar_left = 0; PHASE_BARRIER(); post(); }
        pf.flush();
    }
    if (TRAIN) {                                    // z2 tiles of the last S2 phase
#pragma unroll
        for (int gg = 0; gg < GP; ++gg)
            if ((S::NP2 - 1) * GP + gg < HT) st_blk(tape.z2, e, D::HP, (S::NP2 - 1) * GP + gg, lane, pz2[gg]);
    }
    // ---- S3: ew += SiLU(W3 m + b3), one output tile per group ----------------------------------------
    // stores are issued one phase late (right after the next barrier) so that the barrier's vmcnt(0)
    // never waits for a store that was issued a few cycles earlier
    if (!DO_S3) {
#pragma unroll
        for (int t = 0; t < HT; ++t) st_blk(mbuf, eid, D::HP, t, lane, m[t]);
        PROBE_END(pf);
        TL_END();
        return;
    }
    f4 pend[GP], pendz[TRAIN ? GP : 1];
    f4 om[GP];
    auto s3_phase = [&](int p3, const f4 (&o)[GP], f4 (&onext)[GP]) {
        auto post = [&]() {
            TL(1);
            if (p3 == 0) {
#pragma unroll
                for (int t = 0; t < HT; ++t) st_blk(mbuf, eid, D::HP, t, lane, m[t]);
            } else {
#pragma unroll
                for (int gg = 0; gg < GP; ++gg) {              // (p3-1)*GP+gg < WB always
                    st_f4(orow + 16 * ((p3 - 1) * GP + gg), pend[gg]);
                    if (TRAIN) st_f4(tape.z3 + e * D::WP + 4 * g + 16 * ((p3 - 1) * GP + gg), pendz[gg]);
                }
            }
            pf_begin(p + DIST);
            if (p3 + 1 < S::NP3) {
#pragma unroll
                for (int gg = 0; gg < GP; ++gg) {
                    const int t = (p3 + 1) * GP + gg;
                    if (t < WB)  // !DO_S1: the old state of these rows IS the constant row (never materialised)
                        onext[gg] = DO_S1 ? ld_edge(erow + 16 * t) : ld_f4(c0 + 16 * t + 4 * g);
                }
            }
        };
        auto hook = [&]() { if (RING == 3 && bar_left > 0 && --bar_left == 0) { PHASE_BARRIER(); post(); } pf.tick(); };
        if (RING == 2) { PHASE_BARRIER(); post(); } else bar_left = BAR_AT;
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) {
            const int t = p3 * GP + gg;
            if (t < WB) {
                TL(2);
                const f4 z = chain_tile<HT, TAIL1>(SL(p), gg * G2 + 1, m, A(p, gg * G2), m_tail, hook);
                TL(3);
                if (TRAIN) pendz[gg] = z;
                pend[gg] = o[gg] + silu4(z);
            }
        }
        TL(4);
        if (RING == 3 && bar_left > 0) { bar_left = 0; PHASE_BARRIER(); post(); }
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
    PROBE_END(pf);
    TL_END();
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
    static constexpr bool ROWS4 = (D::H % 16) >= 1 && (D::H % 16) <= 4;      // 13th tile of every third in rows4 packing (mma4_chunk)
    static constexpr int SLAB = G2 > G1 ? G2 : G1;
    static constexpr int NPH = WB + NG2;
    static constexpr int C1 = WB * G1, CHUNKS = C1 + NG2 * G2;
    static constexpr size_t LDS_BYTES = (size_t)2 * SLAB * 1024;
};

// TRAIN: also stores the pre-activation of dir_proj.0 (zd1 [A+1][D1P]) and dir_proj's output before the product
// with rbf_proj (cd [A+1][3][HP]) for the backward pass.
// (Three slabs with the barrier inside the phase, as in k_gcl_edge_v1, were measured here too: 7.09-7.14 ms per step against
// 7.10-7.13 ms - the phases of this kernel are twice as long and its barrier wait was small to begin with; not kept.)
// (Two K blocks of T1 per phase - half of T1's barriers, 74-chunk slabs - : 7.26-7.33 against 7.09-7.11 ms; not kept either.)
// al: the inner rows inside the cutoff (ActList; round 5).  Column c of the launch is list entry c; rows outside the cutoff are not
// touched at all - their q is exactly zero in the reference and nobody reads it (the node stage walks the same list).  The grid is
// sized for A (the host does not know n_act); workgroups behind the list end return at once.
template <class D, int WAVES, bool TRAIN>
__global__ __launch_bounds__(WAVES * 64) void k_equi_edge_v1(TopoDev tp, const float* __restrict__ stream,
                                                             const float* __restrict__ dp0b,
                                                             const float* __restrict__ ew, const float* __restrict__ rbuf,
                                                             float* __restrict__ qbuf, float* __restrict__ zd1,
                                                             float* __restrict__ cdbuf, ActList al) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using S = EquiStream<D>;
    constexpr int WB = D::WB, D1T = D::D1T, RB = D::RB, HT = D::HT, G1 = S::G1, G2 = S::G2;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    SlabPrefetch<WAVES, S::SLAB, 0, 3> pf;
    pf.wave = wave;
    pf.lane_off = (unsigned)lane * 16u;              // LDS-DMA source = uniform chunk address (SGPR pair) + this 32-bit lane offset
    auto pf_begin = [&](int p) {
        const int start = p < WB ? p * G1 : S::C1 + (p - WB) * G2;
        pf.begin(stream, smem, p, start, p >= S::NPH ? 0 : (p < WB ? G1 : G2));
    };
    auto hook = [&]() { pf.tick(); };
    auto A = [&](int p, int j) -> f4 {
        return *reinterpret_cast<const f4*>(smem + ((size_t)(p & 1) * S::SLAB + j) * 256 + lane * 4);
    };
    auto SL = [&](int p) -> const float* { return smem + (size_t)(p & 1) * S::SLAB * 256 + lane * 4; };

    const long long n_cols = al.n != nullptr ? (long long)*al.n : tp.A;
    if ((long long)blockIdx.x * WAVES * 16 >= n_cols) return;                  // (workgroup-uniform: before the first barrier)
    const long long c = ((long long)blockIdx.x * WAVES + wave) * 16 + (lane & 15);
    const bool live = c < n_cols;
    const size_t a = (size_t)(live ? (al.rows != nullptr ? (long long)al.rows[c] : c) : tp.A);    // padding columns use the spare entry A
    const float* erow = ew + (live ? a : (size_t)tp.E) * D::WP + 4 * g;         // inner entry a == physical row a
    f4 d1[D1T];
#pragma unroll
    for (int t = 0; t < D1T; ++t) d1[t] = ld_vec(dp0b, t, lane);
    f4 rb[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) rb[b] = ld_blk(rbuf, a, D::RP, b, lane);
    f4 xn = ld_f4(erow);
    pf_begin(0);
    pf.flush();

    int p = 0;
    for (int b = 0; b < WB; ++b, ++p) {
        phase_barrier();
        const f4 x = xn;
        pf_begin(p + 1);
        if (b + 1 < WB) xn = ld_f4(erow + 16 * (b + 1));
        chain_kouter<D1T>(SL(p), 0, x, d1, hook);
        pf.flush();
    }
    if (TRAIN) {
#pragma unroll
        for (int t = 0; t < D1T; ++t) st_blk(zd1, a, D::D1P, t, lane, d1[t]);
    }
#pragma unroll
    for (int t = 0; t < D1T; ++t) d1[t] = silu4(d1[t]);

    float* qrow = qbuf + a * (size_t)(3 * D::HP) + 4 * g;
    float* cdrow = TRAIN ? cdbuf + a * (size_t)(3 * D::HP) + 4 * g : nullptr;
    f4 pend = f4zero(), pendc = f4zero();
    int pend_off = 0;
    // slots are tile-major (slot i = 3 tt + th): the 4-row tiles (tt = HT - 1) are the last three slots and get their own loop,
    // so that each loop body holds ONE chain variant (both in one body cost 320 bytes of scratch per lane)
    auto t2_phase = [&](int i, auto rows4) {
        phase_barrier();
        if (i > 0) {                                           // stores of the previous phase, issued one phase late
            st_f4(qrow + pend_off, pend);
            if (TRAIN) st_f4(cdrow + pend_off, pendc);
        }
        pf_begin(p + 1);
        const int tt = i / 3, th = i - 3 * tt;
        f4 cd, cr;
        if (decltype(rows4)::value) {                          // the 13th tile of every third: 4 real rows, 4x4x1 MFMAs
            cd = reduce_g(chain_tile4<D1T>(SL(p), 1, d1, A(p, 0), hook));
            cr = reduce_g(chain_tile4<RB>(SL(p), 1 + D1T, rb, f4zero(), hook));
            if (g != 0) { cd = f4zero(); cr = f4zero(); }
        } else {
            cd = chain_tile<D1T>(SL(p), 1, d1, A(p, 0), hook);
            cr = chain_tile<RB>(SL(p), 1 + D1T, rb, f4zero(), hook);
        }
        pf.flush();
        pend = cd * cr;
        if (TRAIN) pendc = cd;
        pend_off = th * D::HP + 16 * tt;
        ++p;
    };
    constexpr int N_FULL = S::ROWS4 ? S::NG2 - 3 : S::NG2;
    for (int i = 0; i < N_FULL; ++i) t2_phase(i, std::false_type{});
    for (int i = N_FULL; i < S::NG2; ++i) t2_phase(i, std::true_type{});
    st_f4(qrow + pend_off, pend);
    if (TRAIN) st_f4(cdrow + pend_off, pendc);
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
            const float xc = lo.xcross ? 1.0f : 0.0f, cx = xc * g[5], cy = xc * g[6], cz = xc * g[7];  // reflect_equiv = False: + x (x) coord_cross
#pragma unroll
            for (int t = 0; t < D::HT; ++t) {
                const f4 q0 = ld_blk(qbuf, a, 3 * D::HP, t, id.lane);
                const f4 q1 = ld_blk(qbuf, a, 3 * D::HP, D::HT + t, id.lane);
                const f4 q2 = ld_blk(qbuf, a, 3 * D::HP, 2 * D::HT + t, id.lane);
                const f4 xs0 = ld_blk(xq, m, 3 * D::HP, t, id.lane) + ld_blk(xq, n, 3 * D::HP, t, id.lane);
                const f4 xs1 = ld_blk(xq, m, 3 * D::HP, D::HT + t, id.lane) + ld_blk(xq, n, 3 * D::HP, D::HT + t, id.lane);
                const f4 xs2 = ld_blk(xq, m, 3 * D::HP, 2 * D::HT + t, id.lane) + ld_blk(xq, n, 3 * D::HP, 2 * D::HT + t, id.lane);
                const f4 xm = xs0 * q0;
                dx[t] += xm;
                const f4 a2 = xs1 * q1 * inv_sqrt3, a3 = xs2 * q2;
                vx[0][t] += (ld_blk(vec_in, (size_t)m * 3 + 0, D::HP, t, id.lane) * a2 + a3 * ux + xm * cx) * inv_sqrt_h;
                vx[1][t] += (ld_blk(vec_in, (size_t)m * 3 + 1, D::HP, t, id.lane) * a2 + a3 * uy + xm * cy) * inv_sqrt_h;
                vx[2][t] += (ld_blk(vec_in, (size_t)m * 3 + 2, D::HP, t, id.lane) * a2 + a3 * uz + xm * cz) * inv_sqrt_h;
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
