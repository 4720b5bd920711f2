// oard_edge_p.h — the GCL edge kernel of oard_edge_v1.h as a PERSISTENT workgroup (round 5).
//
// k_gcl_edge_v1 is one 128-edge tile per workgroup: every tile pays the kernel entry (index loads -> dependent P / Q gathers ->
// a burst of LDS-DMA for the first two slabs -> first barrier: ~12 000 of a tile's ~400 000 cycles) and the workgroup hand-over on
// its CU, and an isolated launch of 2 346 tiles on 256 CUs ends with a round in which 42 CUs work and 214 wait (9.16 rounds).
// Here a workgroup walks a contiguous range of rows in ROUNDS of 8 wave-tiles (16 rows each); the weight-stream ring never stops
// (the slabs of the next round's first two phases are the prefetch targets of the current round's last two phases), the next
// round's row indices and first edge-state blocks are fetched during the residual stage S3, the node terms P[src] + Q[tgt] are added
// BEHIND S1 (gathers in flight during the last S1 phase, behind that phase's LDS-DMA pieces), and the rows are dealt in HALF-tiles (4
// wave-tiles): a workgroup whose share is an odd number of half-tiles ends with a half round in which waves 4..7 (one per SIMD)
// compute and waves 0..3 only keep the barrier / LDS-DMA protocol going - a wave that has its SIMD to itself runs its chains at
// ~1.8 x the speed, so the half round costs ~0.55 of a full one and the launch ends within half a tile time on every CU.
// (Tried and dropped: the node terms of the NEXT round fetched ahead during S3 - over five phases, over the last three, in the last
// one: 52 ... 104 more live registers in a stage that already holds m, hipcc spills 250 ... 700 bytes per lane; the gathers exposed
// at the round start: + 3 % - 3 328 cache lines per workgroup and nothing to hide them behind.)
// Same arithmetic as k_gcl_edge_v1<D, 8, 2, DO_S1, DO_S3, false, 2, 3> with ONE difference in summation order: h1 = W1c.ew + (P + Q)
// here, (P + Q) + W1c.ew there (the tile kernel loads the node terms at its entry, where they hide behind the LDS-DMA start-up; moved
// behind S1 it loses 1.8 %, and this kernel with the gathers at the round start loses 3 %) - a last-bit difference, both within 3e-7
// of the float64 reference per stage (tests/test_hip_parity.py: both against the reference, against each other, and this kernel
// bit-identical to itself over grid sizes).  Reference: model/leftnet.py:157-183 (GCLMessage), residual :164.
#pragma once
#include "oard_edge_v1.h"

// TRAIN: as k_gcl_edge_v1's training mode - the new state goes to a different buffer, the pre-activations z1 / z2 / att / z3 are taped.
template <class D, bool DO_S1, bool DO_S3, bool TRAIN = false>
__global__ __launch_bounds__(512, 2) void k_gcl_edge_p(TopoDev tp, const float* __restrict__ stream,
                                                       const float* __restrict__ P, const float* __restrict__ Q,
                                                       const float* __restrict__ u0, const float* __restrict__ c0,
                                                       long long r0, long long r1, const float* ew_in, float* ew_out,
                                                       float* __restrict__ mbuf, GclTape tape) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WAVES = 8, GP = 2, RING = 3, DIST = RING - 1;
    using S = GclStream<D, GP>;
    constexpr int HT = D::HT, WB = D::WB, G1 = S::G1, G2 = S::G2;
    constexpr bool TAIL1 = S::TAIL1, ROWS4 = S::ROWS4;
    constexpr int P0 = DO_S1 ? 0 : S::NP1, PEND = DO_S3 ? S::NPH : S::NP1 + S::NP2, NPR = PEND - P0;     // local phases of a round
    static_assert(NPR >= DIST, "a round must be at least as long as the prefetch distance");
    static_assert(HT >= 2, "the phase barrier sits inside the first chain of a phase");
    constexpr int BAR_AT = OARD_BAR_AT < HT / 2 ? OARD_BAR_AT : HT / 2;

    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- this workgroup's share: half-tiles [hb0, hb1) of the launch's cdiv(rows, 64) --------------------------------------
    const long long WT = (r1 - r0 + 15) >> 4, NH = (WT + 3) >> 2;
    const long long hb0 = NH * blockIdx.x / gridDim.x, hb1 = NH * (blockIdx.x + 1) / gridDim.x;
    const int nh = (int)(hb1 - hb0), nr = (nh + 1) >> 1;
    if (nr == 0) return;
    // round r is a half round iff it is the last one of an odd share; then waves 4..7 take its four wave-tiles
    auto is_half = [&](int r) -> bool { return 2 * r + 1 == nh; };
    auto active_in = [&](int r) -> bool { return !(is_half(r) && wave < 4); };
    auto row_of = [&](int r) -> size_t {          // this lane's physical row in round r; the spare row E for padding columns
        const int slot = is_half(r) ? wave - 4 : wave;
        const long long c = r0 + (((hb0 + 2 * r) << 2) + slot) * 16 + (lane & 15);
        return (size_t)((slot >= 0 && c < r1) ? c : tp.E);
    };

    SlabPrefetch<WAVES, S::SLAB, 1, 1, RING> pf;
    pf.wave = wave;
    pf.lane_off = (unsigned)lane * 16u;
    int gp = 0;                                                // phases executed so far: ring slot of a phase = its count % RING
    // Loop-invariant code motion is the enemy of a persistent kernel at its register limit: left alone, hipcc computes the per-lane
    // 64-bit source address of every LDS-DMA piece of the unrolled phases ONCE, in front of the round loop (dozens of VGPR pairs: 1.1 KB
    // of scratch per lane), and would do the same to the constant-row loads.  The pointers the round body uses are therefore re-derived
    // at the top of every round through an empty asm statement the optimiser has to treat as a new definition.
    const float* strm = stream;
    const float* c0r = c0;
    const float* u0r = u0;
    auto pf_local = [&](int q, int at, bool ok) {              // the slab of local phase q into the ring slot of global phase `at`
        int start = 0, n = 0;
        if (ok) {
            if (q < S::NP1) { start = q * GP * G1; n = min(GP, WB - q * GP) * G1; }
            else if (q < S::NP1 + S::NP2) { const int k = q - S::NP1; start = S::C1 + k * GP * G2; n = min(GP, S::NG2 - k * GP) * G2; }
            else { const int k = q - S::NP1 - S::NP2; start = S::C1 + S::C2 + k * GP * G2; n = min(GP, WB - k * GP) * G2; }
        }
        pf.begin(strm, smem, at, start, n);
    };
    bool more = nr > 1;                                        // another round follows the current one (workgroup-uniform)
    auto pf_ahead = [&](int lp) {                              // called in local phase lp: prefetch DIST phases ahead, across the round boundary
        int q = lp + DIST;
        bool ok = true;
        if (q >= PEND) { q -= NPR; ok = more; }
        pf_local(q, gp + DIST, ok);
    };
    auto slab_of = [&](int p) -> int { return p % RING; };
    auto A = [&](int j) -> f4 { return *reinterpret_cast<const f4*>(smem + ((size_t)slab_of(gp) * S::SLAB + j) * 256 + lane * 4); };
    auto SL = [&]() -> const float* { return smem + (size_t)slab_of(gp) * S::SLAB * 256 + lane * 4; };

    // ---- kernel entry: the LDS-DMA of the first two slabs goes out BEFORE the index loads (nothing depends on them) ----------
    pf_local(P0, 0, true);
    pf.flush();
    pf_local(P0 + 1 < PEND ? P0 + 1 : P0, 1, P0 + 1 < PEND || more);
    pf.flush();

    // per-round state of this lane's column: h1 = P[src] + Q[tgt] (+ u0 without S1), the first edge-state blocks, the message row
    size_t e = 0;
    const float* erow = nullptr;
    float* orow = nullptr;
    f4 h1[HT];
    f4 xn[GP];
    auto set_rows = [&](size_t row) {
        e = row;
        erow = ew_in + e * D::WP + 4 * g;
        orow = ew_out + e * D::WP + 4 * g;
    };
    auto load_xn = [&](f4 (&x)[GP], const float* er) {
#pragma unroll
        for (int gg = 0; gg < GP; ++gg) x[gg] = (DO_S1 && gg < WB) ? ld_edge(er + 16 * gg) : f4zero();
    };
    size_t eid = 0;
    int src = 0, tgt = 0;
    // h1 in front of the round: zero with S1 (P[src] + Q[tgt] are added behind S1, gathers in flight during its last phase),
    // P + Q + u0 without
    auto load_h1 = [&]() {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            if (DO_S1) h1[t] = f4zero();
            else h1[t] = ld_blk(P, src, D::HP, t, lane) + ld_blk(Q, tgt, D::HP, t, lane) + ld_vec(u0r, t, lane);
        }
    };
    if (active_in(0)) {
        set_rows(row_of(0));
        eid = (size_t)tp.row_eid[e]; src = tp.row_src[e]; tgt = tp.row_tgt[e];
        load_h1();
        load_xn(xn, erow);
    }
    PHASE_BARRIER();                                           // the first slab is published (the second by the barrier of the first phase)

    int bar_left = 0;
    for (int r = 0; r < nr; ++r) {
        more = r + 1 < nr;
        strm = stream; c0r = c0; u0r = u0;
        asm volatile("" : "+s"(strm), "+s"(c0r), "+s"(u0r));
        if (!active_in(r)) {
            // waves 0..3 of a half round: keep the protocol going - one barrier per phase, the LDS-DMA pieces right behind it
            for (int lp = P0; lp < PEND; ++lp) { PHASE_BARRIER(); pf_ahead(lp); pf.flush(); ++gp; }
            continue;                                          // (a half round is always the last one)
        }
        const bool nxt = more && active_in(r + 1);             // this wave has columns in the next round (wave-uniform)
        // the next round's columns: indices fetched in the first S3 phase, first edge-state blocks in the last one (declared per round:
        // as loop-carried variables they would be live through S1 and S2 as well)
        int src_n = 0, tgt_n = 0;
        size_t eid_n = 0;
        f4 xn_n[GP];
        f4 hp[HT], hq[HT];                                     // P[src], Q[tgt]: in flight during the last S1 phase
        int lp = P0;                                           // local phase

        // ---- S1: h1 += W1c . ew   (K-outer) -----------------------------------------------------------------------
        f4 h1x = f4zero();
        f4 xm[GP];
        auto s1_phase = [&](int p1, const f4 (&x)[GP], f4 (&xnext)[GP], auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;   // the last S1 phase (peeled): its post() issues the P / Q gathers
            auto post = [&]() {
                pf_ahead(lp);
                if (p1 + 1 < S::NP1) {
#pragma unroll
                    for (int gg = 0; gg < GP; ++gg) {
                        const int b = (p1 + 1) * GP + gg;
                        if (b < WB) xnext[gg] = ld_edge(erow + 16 * b);
                    }
                }
                if (LAST) {
                    pf.flush();                                // this phase's LDS-DMA pieces as a burst, in front of the gathers: the wait for the
                                                               // gathers behind S1 is a vmcnt(0) that would otherwise also wait for a piece issued a few cycles earlier
#pragma unroll
                    for (int t = 0; t < HT; ++t) { hp[t] = ld_blk(P, src, D::HP, t, lane); hq[t] = ld_blk(Q, tgt, D::HP, t, lane); }
                }
            };
            auto hook = [&]() { if (bar_left > 0 && --bar_left == 0) { PHASE_BARRIER(); post(); } pf.tick(); };
            bar_left = BAR_AT;
#pragma unroll
            for (int gg = 0; gg < GP; ++gg)
                if (p1 * GP + gg < WB) chain_kouter<HT, ROWS4>(SL(), gg * G1, x[gg], h1, h1x, hook);
            if (bar_left > 0) { bar_left = 0; PHASE_BARRIER(); post(); }
            pf.flush();
            ++gp; ++lp;
        };
        if (DO_S1) {
            int p1 = 0;
            for (; p1 + 2 < S::NP1; p1 += 2) { s1_phase(p1, xn, xm, std::false_type{}); s1_phase(p1 + 1, xm, xn, std::false_type{}); }
            if (p1 + 2 == S::NP1) { s1_phase(p1, xn, xm, std::false_type{}); s1_phase(p1 + 1, xm, xn, std::true_type{}); }
            else s1_phase(p1, xn, xm, std::true_type{});
        }
        if (ROWS4 && DO_S1) {
            const f4 v = reduce_g(h1[HT - 1] + h1x);
            h1[HT - 1] = g == 0 ? v : f4zero();
        }
        if (DO_S1) {
#pragma unroll
            for (int t = 0; t < HT; ++t) h1[t] += hp[t] + hq[t];
        }
        if (TRAIN) {
#pragma unroll
            for (int t = 0; t < HT; ++t) st_blk(tape.z1, e, D::HP, t, lane, h1[t]);
        }
#pragma unroll
        for (int t = 0; t < HT; ++t) h1[t] = silu4(h1[t]);
        const float h1_tail = TAIL1 ? tail_compact(h1[HT - 1], lane) : 0.f;

        // ---- S2: m = SiLU(W2 h1 + b2); gate = SiLU(watt . m + batt) --------------------------------------------------
        f4 m[HT];
        f4 on[GP];
        f4 pz2[TRAIN ? GP : 1];                     // TRAIN: z2 tiles of the previous phase, stored behind the next barrier
        float m_tail = 0.f;
#pragma unroll
        for (int p2 = 0; p2 < S::NP2; ++p2) {
            auto post = [&]() {
                if (TRAIN && p2 > 0) {
#pragma unroll
                    for (int gg = 0; gg < GP; ++gg)
                        if ((p2 - 1) * GP + gg < HT) st_blk(tape.z2, e, D::HP, (p2 - 1) * GP + gg, lane, pz2[gg]);
                }
                pf_ahead(lp);
                if (DO_S3 && p2 == S::NP2 - 1) {
#pragma unroll
                    for (int gg = 0; gg < GP; ++gg)
                        on[gg] = gg < WB ? (DO_S1 ? ld_edge(erow + 16 * gg) : ld_f4(c0r + 16 * gg + 4 * g)) : f4zero();
                }
            };
            auto hook = [&]() { if (bar_left > 0 && --bar_left == 0) { PHASE_BARRIER(); post(); } pf.tick(); };
            bar_left = BAR_AT;
#pragma unroll
            for (int gg = 0; gg < GP; ++gg) {
                const int tg = p2 * GP + gg;
                if (tg < S::NG2) {
                    const f4 bias = A(gg * G2);
                    f4 acc;
                    if (ROWS4 && tg >= HT - 1) {
                        acc = reduce_g(tg < HT ? chain_tile4<HT>(SL(), gg * G2 + 1, h1, bias, hook)
                                               : chain_tile4<HT>(SL(), gg * G2 + 1, m, bias, hook));
                        if (tg < HT && g != 0) acc = f4zero();
                    } else {
                        acc = tg < HT ? chain_tile<HT, TAIL1>(SL(), gg * G2 + 1, h1, bias, h1_tail, hook)
                                      : chain_tile<HT, TAIL1>(SL(), gg * G2 + 1, m, bias, m_tail, hook);
                    }
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
            if (bar_left > 0) { bar_left = 0; PHASE_BARRIER(); post(); }
            pf.flush();
            ++gp; ++lp;
        }

        if (TRAIN) {                                // z2 tiles of the last S2 phase
#pragma unroll
            for (int gg = 0; gg < GP; ++gg)
                if ((S::NP2 - 1) * GP + gg < HT) st_blk(tape.z2, e, D::HP, (S::NP2 - 1) * GP + gg, lane, pz2[gg]);
        }
        if (!DO_S3) {
#pragma unroll
            for (int t = 0; t < HT; ++t) st_blk(mbuf, eid, D::HP, t, lane, m[t]);
        } else {
            // ---- S3: ew += SiLU(W3 m + b3); the next round's columns are fetched behind its barriers -------------------
            f4 pend[GP], pendz[TRAIN ? GP : 1];
            f4 om[GP];
            auto s3_phase = [&](int p3, const f4 (&o)[GP], f4 (&onext)[GP], auto last_tag) {
                constexpr bool LAST = decltype(last_tag)::value;   // the last S3 phase (peeled): its post() fetches the next round's first edge-state blocks
                auto post = [&]() {
                    if (p3 == 0) {
#pragma unroll
                        for (int t = 0; t < HT; ++t) st_blk(mbuf, eid, D::HP, t, lane, m[t]);
                    } else {
#pragma unroll
                        for (int gg = 0; gg < GP; ++gg) {
                            st_f4(orow + 16 * ((p3 - 1) * GP + gg), pend[gg]);
                            if (TRAIN) st_f4(tape.z3 + e * D::WP + 4 * g + 16 * ((p3 - 1) * GP + gg), pendz[gg]);
                        }
                    }
                    pf_ahead(lp);
                    if (p3 + 1 < S::NP3) {
#pragma unroll
                        for (int gg = 0; gg < GP; ++gg) {
                            const int t = (p3 + 1) * GP + gg;
                            if (t < WB) onext[gg] = DO_S1 ? ld_edge(erow + 16 * t) : ld_f4(c0r + 16 * t + 4 * g);
                        }
                    }
                    if (nxt) {
                        if (p3 == 0) {                         // (with NP3 == 1 the blocks below wait for these: still one round trip saved)
                            const size_t en = row_of(r + 1);
                            src_n = tp.row_src[en]; tgt_n = tp.row_tgt[en]; eid_n = (size_t)tp.row_eid[en];
                        }
                        if (LAST) load_xn(xn_n, ew_in + row_of(r + 1) * D::WP + 4 * g);
                    }
                };
                auto hook = [&]() { if (bar_left > 0 && --bar_left == 0) { PHASE_BARRIER(); post(); } pf.tick(); };
                bar_left = BAR_AT;
#pragma unroll
                for (int gg = 0; gg < GP; ++gg) {
                    const int t = p3 * GP + gg;
                    if (t < WB) {
                        const f4 z = chain_tile<HT, TAIL1>(SL(), gg * G2 + 1, m, A(gg * G2), m_tail, hook);
                        if (TRAIN) pendz[gg] = z;
                        pend[gg] = o[gg] + silu4(z);
                    }
                }
                if (bar_left > 0) { bar_left = 0; PHASE_BARRIER(); post(); }
                pf.flush();
                ++gp; ++lp;
            };
            {
                int p3 = 0;
                for (; p3 + 2 < S::NP3; p3 += 2) { s3_phase(p3, on, om, std::false_type{}); s3_phase(p3 + 1, om, on, std::false_type{}); }
                if (p3 + 2 == S::NP3) { s3_phase(p3, on, om, std::false_type{}); s3_phase(p3 + 1, om, on, std::true_type{}); }
                else s3_phase(p3, on, om, std::true_type{});
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

        // ---- hand over to the next round -------------------------------------------------------------------------------
        if (nxt) {
            set_rows(row_of(r + 1));
            if (DO_S3) {                                       // indices and first edge-state blocks were fetched during S3
                eid = eid_n; src = src_n; tgt = tgt_n;
#pragma unroll
                for (int gg = 0; gg < GP; ++gg) xn[gg] = xn_n[gg];
            } else {                                           // no S3 to hide behind (last layer, inter-object rows)
                eid = (size_t)tp.row_eid[e]; src = tp.row_src[e]; tgt = tp.row_tgt[e];
                load_xn(xn, erow);
            }
            load_h1();
        }
    }
}
