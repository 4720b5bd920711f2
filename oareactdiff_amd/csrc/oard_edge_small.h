// oard_edge_small.h — latency-oriented edge kernels for small launches (a handful of reactions).
//
// The throughput kernels of oard_edge_v1.h give every wavefront 16 edges and push them through the whole
// per-layer chain: with few edges the launch cannot fill the chip and its duration is the time ONE wave needs
// for ~5,000 (GCL) / ~12,500 (EquiMessage) dependent MFMAs - 120 / 240 us per layer however small the batch.
// Here a workgroup of WAVES wavefronts shares the 16 edges instead: the activations of the 16 edges live in
// LDS as 1-KiB feature blocks (the node-stage layout of oard_node_v1.h), the output tiles of every dense layer
// are dealt round-robin to the waves, the layers are separated by __syncthreads(), and the weights (A operand)
// come straight from L2 with a 13-chunk prefetch (dense_tile_lds).  16x more wavefronts per edge tile, so a
// B = 1 launch spreads over the chip.  Same arithmetic per column as the v0 reference kernels (k_gcl_edge /
// k_equi_edge), same buffers and row conventions as the v1 kernels.
#pragma once
#include "oard_node_v1.h"

// Measured and dropped in round 2 (B = 1): 32 edges per workgroup sharing every weight chunk (1.70 -> 2.08 ms per call) and a
// deeper weight prefetch (ring of 3-4 steps: GCL 50 -> 56 us per layer, EquiMessage unchanged): these launches are bound by the
// MFMA time of the CUs that hold a tile (one 16-edge tile = ~17 us of a CU for GCL, ~43 us for EquiMessage, and B = 1 has 294 / 95
// tiles for 256 CUs), not by the weight traffic or its latency.
template <class D>
struct GclSmall {
    static constexpr size_t LDS_BYTES = (size_t)(D::WB + 2 * D::HT) * 1024;
};

// GCLMessage edge part on physical rows [r0, r1):  h1 = SiLU(W1c ew + P[src] + Q[tgt]);  m = SiLU(W2 h1 + b2);
// m *= SiLU(watt.m + batt);  mbuf[eid] = m;  ew += SiLU(W3 m + b3).     DO_S1 = false: rows whose state is the
// constant row c0 (W1c c0 = u0 precomputed);  DO_S3 = false: the updated state is never read (last layer).
template <class D, int WAVES, bool DO_S1, bool DO_S3>
__global__ __launch_bounds__(WAVES * 64) void k_gcl_edge_small(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                               const float* __restrict__ P, const float* __restrict__ Q,
                                                               const float* __restrict__ u0, const float* __restrict__ c0,
                                                               long long r0, long long r1, float* __restrict__ ew,
                                                               float* __restrict__ mbuf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WB = D::WB, HT = D::HT;
    float* ewv = smem;                      // [WB] edge state of the 16 rows
    float* h1v = smem + WB * 256;           // [HT] h1, later the gated message
    float* h2v = h1v + HT * 256;            // [HT]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const long long r = r0 + (long long)blockIdx.x * 16 + (lane & 15);
    const bool valid = r < r1;
    const size_t row = (size_t)(valid ? r : tp.E);          // padding columns use the spare row
    const int src = tp.row_src[row], tgt = tp.row_tgt[row];
    float* erow = ew + row * D::WP + 4 * g;

    if (DO_S1) {
        for (int b = wave; b < WB; b += WAVES) lds_st(ewv, b, lane, ld_f4(erow + 16 * b));
        __syncthreads();
    }
    for (int t = wave; t < HT; t += WAVES) {
        const f4 pq = ld_blk(P, src, D::HP, t, lane) + ld_blk(Q, tgt, D::HP, t, lane);
        const f4 acc = DO_S1 ? dense_tile_lds<WB>(wb + lo.W1c, t, ewv, lane, pq) : pq + ld_vec(u0, t, lane);
        lds_st(h1v, t, lane, silu4(acc));
    }
    __syncthreads();
    for (int t = wave; t < HT; t += WAVES)
        lds_st(h2v, t, lane, silu4(dense_tile_lds<HT>(wb + lo.W2, t, h1v, lane, ld_vec(wb + lo.b2, t, lane))));
    __syncthreads();
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const f4 m = lds_blk(h2v, t, lane), w = ld_vec(wb + lo.watt, t, lane);
        part += m.x * w.x + m.y * w.y + m.z * w.z + m.w * w.w;
    }
    const float gate = silu1(col_reduce(part) + wb[lo.batt]);
    const size_t eid = (size_t)tp.row_eid[row];
    for (int t = wave; t < HT; t += WAVES) {
        const f4 m = lds_blk(h2v, t, lane) * gate;
        lds_st(h1v, t, lane, m);                                   // h1 is dead since the barrier above
        if (valid) st_blk(mbuf, eid, D::HP, t, lane, m);
    }
    if (!DO_S3) return;
    __syncthreads();
    for (int t = wave; t < WB; t += WAVES) {
        const f4 acc = dense_tile_lds<HT>(wb + lo.W3, t, h1v, lane, ld_vec(wb + lo.b3, t, lane));
        const f4 old = DO_S1 ? lds_blk(ewv, t, lane) : ld_vec(c0, t, lane);
        if (valid) st_f4(erow + 16 * t, old + silu4(acc));
    }
}

template <class D>
struct EquiSmall {
    static constexpr size_t LDS_BYTES = (size_t)(D::WB + D::D1T + D::RB) * 1024;
};

// EquiMessage edge part on inner edges:  d1 = SiLU(dir_proj.0 ew);  q = (dir_proj.2 d1 + b) * (rbf_proj rbf)
// -> qbuf[a][3][HP]  (what k_equi_node_v1 consumes)
template <class D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_equi_edge_small(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                                const float* __restrict__ ew,
                                                                const float* __restrict__ rbuf, float* __restrict__ qbuf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WB = D::WB, D1T = D::D1T, RB = D::RB, HT = D::HT;
    float* ewv = smem;                      // [WB]
    float* d1v = smem + WB * 256;           // [D1T]
    float* rbv = d1v + D1T * 256;           // [RB]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const long long c = (long long)blockIdx.x * 16 + (lane & 15);
    const bool valid = c < tp.A;
    const size_t a = (size_t)(valid ? c : tp.A);                       // spare entry A for the padding columns
    const float* erow = ew + (valid ? a : (size_t)tp.E) * D::WP + 4 * g;  // inner entry a == physical row a
    for (int b = wave; b < WB + RB; b += WAVES) {
        if (b < WB) lds_st(ewv, b, lane, ld_f4(erow + 16 * b));
        else lds_st(rbv, b - WB, lane, ld_blk(rbuf, a, D::RP, b - WB, lane));
    }
    __syncthreads();
    for (int t = wave; t < D1T; t += WAVES)
        lds_st(d1v, t, lane, silu4(dense_tile_lds<WB>(wb + lo.dp0, t, ewv, lane, ld_vec(wb + lo.dp0b, t, lane))));
    __syncthreads();
    float* qrow = qbuf + a * (size_t)(3 * D::HP) + 4 * g;
    for (int t = wave; t < 3 * HT; t += WAVES) {
        const f4 cd = dense_tile_lds<D1T>(wb + lo.dp2, t, d1v, lane, ld_vec(wb + lo.dp2b, t, lane));
        const f4 cr = dense_tile_lds<RB>(wb + lo.rbfp, t, rbv, lane, f4zero());
        if (valid) st_f4(qrow + 16 * t, cd * cr);                       // t = th * HT + tt  ->  offset th * HP + 16 tt
    }
}

// =====================================================================================================================================
// Stage-split latency path of the EquiMessage edge part.  One CU needs >= 43 us for the ~13 000 MFMAs of a 16-edge tile however the
// work is dealt to its waves, and a B = 1 launch has only 95 tiles for 256 CUs (k_equi_edge_small: 63 us per layer, unchanged by a
// deeper weight prefetch, slower with 32 edges per workgroup).  Here each of the two dense stages is its own launch and a
// workgroup owns 16 edges x WAVES OUTPUT TILES of one stage, so a tile's work spreads over five CUs; d1 (16 x 592 floats per tile)
// goes through a scratch buffer.  Same arithmetic per column.  Measured per layer at B = 1: 63 -> 42 us; from ~2 reactions on
// (190 tiles) the single kernel wins again, and the same split of the GCL kernel (294 tiles at B = 1) loses at every size
// (three launches, redundant tile loads: 67 -> 79 us), so only this one is kept.
// blockIdx.x = tile * NOG + og (og = group of WAVES output tiles).
// =====================================================================================================================================
template <class D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_equi_small_s1(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                              const float* __restrict__ ew, float* __restrict__ d1buf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];        // [WB] edge state
    constexpr int WB = D::WB, D1T = D::D1T, NOG = (D1T + WAVES - 1) / WAVES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int tile = blockIdx.x / NOG, og = blockIdx.x % NOG;
    const long long c = (long long)tile * 16 + (lane & 15);
    const bool valid = c < tp.A;
    const size_t a = (size_t)(valid ? c : tp.A);                         // spare entry A for the padding columns
    const float* erow = ew + (valid ? a : (size_t)tp.E) * D::WP + 4 * g; // inner entry a == physical row a
    for (int b = wave; b < WB; b += WAVES) lds_st(smem, b, lane, ld_f4(erow + 16 * b));
    __syncthreads();
    const int t = og * WAVES + wave;
    if (t >= D1T) return;
    TileJob job[1] = {tile_job<WB>(wb + lo.dp0, t, smem)};
    f4 acc[1] = {ld_vec(wb + lo.dp0b, t, lane)};
    dense_seq<WB, 1, 7, 3>(job, lane, acc);
    st_blk(d1buf, a, D::D1P, t, lane, silu4(acc[0]));
}
template <class D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_equi_small_s2(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                              const float* __restrict__ d1buf, const float* __restrict__ rbuf,
                                                              float* __restrict__ qbuf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];        // [D1T] d1 | [RB] rbf
    constexpr int D1T = D::D1T, RB = D::RB, HT = D::HT, NOG = (3 * HT + WAVES - 1) / WAVES;
    float* rbv = smem + D1T * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int tile = blockIdx.x / NOG, og = blockIdx.x % NOG;
    const long long c = (long long)tile * 16 + (lane & 15);
    const bool valid = c < tp.A;
    const size_t a = (size_t)(valid ? c : tp.A);
    for (int b = wave; b < D1T + RB; b += WAVES) {
        if (b < D1T) lds_st(smem, b, lane, ld_blk(d1buf, a, D::D1P, b, lane));
        else lds_st(rbv, b - D1T, lane, ld_blk(rbuf, a, D::RP, b - D1T, lane));
    }
    __syncthreads();
    const int t = og * WAVES + wave;
    if (t >= 3 * HT) return;
    TileJob jd[1] = {tile_job<D1T>(wb + lo.dp2, t, smem)}, jr[1] = {tile_job<RB>(wb + lo.rbfp, t, rbv)};
    f4 cd[1] = {ld_vec(wb + lo.dp2b, t, lane)}, cr[1] = {f4zero()};
    dense_seq<D1T, 1, 7, 3>(jd, lane, cd);
    dense_seq<RB, 1, 7, 2>(jr, lane, cr);
    if (valid) st_f4(qbuf + a * (size_t)(3 * D::HP) + 4 * g + 16 * t, cd[0] * cr[0]);      // t = th * HT + tt
}
