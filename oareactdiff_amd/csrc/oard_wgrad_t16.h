// oard_wgrad_t16.h — weight-gradient GEMM of the long (edge-level) contractions, round 4.
//
// Reference: the parameter gradients torch autograd forms for the nn.Linear layers of GCLMessage / EquiMessage
// (oa_reactdiff/model/leftnet.py:157-183, 244-284) in DDPMModule.training_step (trainer/pl_trainer.py:327-347):
// dW[o][i] = sum_rows dY[row][o] * act(X[row][i]), db[o] = sum_rows dY[row][o].  The contraction index is the edge, the SLOW index
// of both row-major operands, 97 152 ... 300 288 long; the outputs are small (196 ... 684 wide).
//
// Why a third kernel (k_wgrad, k_wgrad_lds: oard_edge_bwd.h).  Measured in round 3, the LDS-panel kernel already issues MFMAs at
// ~91 % of the pipe rate on the 684 x 196 products - but its tiles are 64 features wide on one operand and 7 x 16 on the other, in
// 4 x 2 groups: 684 x 196 executes as 768 x 224 (+28 %), 196 x 196 as 256 x 224 (+49 %), and the workgroups of one row chunk
// re-read the shared operand from HBM (58.6 GB fetched per training step for 22 GB of unique operands).  Here
//   * BOTH operands are cut into 16-feature tiles (the MFMA's own granularity: 684 -> 43 tiles, 196 -> 13, 588 -> 37), so the
//     only padding left is 196 -> 208 / 684 -> 688 / 588 -> 592;
//   * an 8-wave workgroup owns up to 24 x 14 tiles (4 x 2 waves of up to 6 x 7 tiles = 168 accumulator registers): the whole
//     196-wide operand and half of the 684-wide one, so 684 x 196 needs 2 workgroups per row chunk, 196 x 196 one, 588 x 684
//     six - and the workgroups of a chunk are placed on the same XCD back to back, so that the operand they share is served by
//     that XCD's L2 after its first read;
//   * tiles are dealt to the waves so that the two waves of a SIMD (w and w + 4) get one large and one small tile set: for
//     22 x 13 tiles every SIMD executes 71 or 72 tile-MFMAs per 4 rows (the even deal would give 78 / 65);
//   * rows are streamed HBM -> LDS by LDS-DMA (global_load_lds, 16 bytes per lane, no VGPR round trip) in groups of 16 rows
//     into a ring of three 40-KB buffers, two groups ahead; one barrier per group (~10 000 MFMA cycles per SIMD);
//   * SiLU-on-load (X = SiLU(z), two of the five products per layer) is applied IN LDS, once per element, to the group that
//     has landed but is not yet being read (the same barrier publishes it).
// Both operands of an MFMA step are plain ds_read_b32 (lane (g, i): row 4 s + g, feature 16 t + i of the panel; the panel's
// row strides are = 16 mod 32 floats, so the two rows a 32-lane half reads sit in different banks): 13 reads per 42 MFMAs.
// Deterministic: fixed chunking, per-chunk partials, the fixed-order reduce passes of oard_edge_bwd.h (k_wgrad_reduce /
// k_bgrad_reduce).  Layout of the partials: [chunk][PP][QP], P-major (P = the operand with more tiles), psum [chunk][PP],
// qsum [chunk][QP] - exactly k_wgrad's, so the reduce passes are shared.
#pragma once
#include "oard_edge_bwd.h"

#define WGT_R 16                           // rows per group
#define WGT_SP 400                         // P panel row stride (floats): <= 24 tiles (384) + 16
#define WGT_SQ 240                         // Q panel row stride: <= 14 tiles (224) + 16
#define WGT_BUF (WGT_R * (WGT_SP + WGT_SQ))        // floats per ring buffer (40 KB)
#define WGT_RING 3
#define WGT_LDS_BYTES ((size_t)WGT_RING * WGT_BUF * sizeof(float))

struct WgtArgs {
    const float* P; const float* Q;        // row-major operands; columns [0, 16 MT) / [0, 16 NT) are readable (pads are zero)
    int ldP, ldQ;
    int MT, NT;                            // 16-feature tiles of P / Q
    int nPT, nQT;                          // workgroup tiles along P / Q (a workgroup: <= 4 TM x 2 TN tiles)
    long long r0, r1, rpc;                 // rows [r0, r1), rows per chunk (multiple of WGT_R)
    int n_chunks;
    float* partial;                        // [n_chunks][16 MT][16 NT]
    float* psum;                           // [n_chunks][16 MT] column sums of P (bias gradient when P = dY), or nullptr
    float* qsum;                           // [n_chunks][16 NT] column sums of Q, or nullptr
};

// part k of `n` tiles cut into `parts` nearly equal pieces (the larger pieces first): [start, start + size)
OARD_DEV int wgt_start(int n, int parts, int k) { return k * (n / parts) + (k < n % parts ? k : n % parts); }
__host__ __device__ inline int wgt_size(int n, int parts, int k) { return n / parts + (k < n % parts ? 1 : 0); }

template <int TMV, int TNV, bool QSILU, int BIAS>
OARD_DEV void wgt_run(const WgtArgs& a, float* smem, int wave, int lane, int chunk, int p_tile0, int p_tiles, int q_tile0, int q_tiles,
                      int pl0, int tm, int ql0, int tn, bool want_ps, bool want_qs) {
    // p_tile0 / p_tiles: this WORKGROUP's P tiles (absolute first tile, count); pl0 / tm: this WAVE's first tile inside them and its
    // tile count (tm <= TMV; with tm < TMV the surplus accumulators are never stored)
    const int g = lane >> 4, i = lane & 15, tid = wave * 64 + lane;
    const long long rb = a.r0 + (long long)chunk * a.rpc;
    const long long re = rb + a.rpc < a.r1 ? rb + a.rpc : a.r1;
    const int ngroups = (int)((re - rb + WGT_R - 1) / WGT_R);
    const int wP4 = p_tiles * 4, wQ4 = q_tiles * 4;                 // float4 per panel row
    const float* gP = a.P + (size_t)p_tile0 * 16 + 4 * lane;         // + row * ldP
    const float* gQ = a.Q + (size_t)q_tile0 * 16 + 4 * lane;

    auto issue = [&](int grp) {                                      // rows `wave` and `wave + 8` of group grp -> ring buffer grp % 3
        float* buf = smem + (size_t)(grp % WGT_RING) * WGT_BUF;
#pragma unroll
        for (int h = 0; h < WGT_R / 8; ++h) {
            const int r = wave + 8 * h;
            const long long row = rb + (long long)grp * WGT_R + r;
            float* dP = buf + r * WGT_SP;
            float* dQ = buf + WGT_R * WGT_SP + r * WGT_SQ;
            if (row < re) {                                          // wave-uniform
                const float* sp = gP + (size_t)row * a.ldP;
                const float* sq = gQ + (size_t)row * a.ldQ;
                if (lane < wP4) glds16(sp, dP);
                if (lane + 64 < wP4) glds16(sp + 256, dP + 256);
                if (lane < wQ4) glds16(sq, dQ);
            } else {                                                 // beyond the chunk: the row counts as zero in both operands
                if (lane < wP4) st_f4(dP + 4 * lane, f4zero());
                if (lane + 64 < wP4) st_f4(dP + 256 + 4 * lane, f4zero());
                if (lane < wQ4) st_f4(dQ + 4 * lane, f4zero());
            }
        }
    };
    // SiLU in place on the Q panel of a landed group: float4 number tid, tid + 512 of its WGT_R x wQ4 grid
    int so[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + 512 * k;
        so[k] = idx < WGT_R * wQ4 ? (idx / wQ4) * WGT_SQ + 4 * (idx % wQ4) : -1;
    }
    auto transform = [&](int grp) {
        float* q = smem + (size_t)(grp % WGT_RING) * WGT_BUF + WGT_R * WGT_SP;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (so[k] >= 0) st_f4(q + so[k], silu4(ld_f4(q + so[k])));
    };

    f4 acc[TMV][TNV];
#pragma unroll
    for (int x = 0; x < TMV; ++x)
#pragma unroll
        for (int y = 0; y < TNV; ++y) acc[x][y] = f4zero();
    // column sums of the dY operand (the bias gradient; never the SiLU operand): thread tid owns column tid of the workgroup's P
    // (BIAS 1) or Q (BIAS 2) panel and adds its 16 rows of every group in row order - one register, work spread over all waves
    float bsum = 0.f;
    const bool bias_on = BIAS == 1 ? (want_ps && tid < 4 * wP4) : (BIAS == 2 ? (want_qs && tid < 4 * wQ4) : false);
    auto colsum = [&](int grp) {
        const float* col = smem + (size_t)(grp % WGT_RING) * WGT_BUF + (BIAS == 1 ? tid : WGT_R * WGT_SP + tid);
#pragma unroll
        for (int r = 0; r < WGT_R; ++r) bsum += col[r * (BIAS == 1 ? WGT_SP : WGT_SQ)];
    };
    // per-lane read offset (floats) inside a buffer; tile x / y of this wave is a compile-time displacement from it.  A wave with
    // fewer tiles than its variant has accumulators (tm < TMV) reads whatever follows its last tile - inside the LDS allocation,
    // values never stored
    const int oa = g * WGT_SP + 16 * pl0 + i, ob = WGT_R * WGT_SP + g * WGT_SQ + 16 * ql0 + i;

    issue(0);
    phase_barrier();
    if (QSILU) transform(0);
    if (ngroups > 1) issue(1);
    for (int grp = 0; grp < ngroups; ++grp) {
        phase_barrier();                                             // group grp + 1 landed, transform(grp) visible, buffer (grp + 2) % 3 free
        if (grp + 2 < ngroups) issue(grp + 2);
        if (QSILU && grp + 1 < ngroups) transform(grp + 1);
        if (BIAS != 0 && bias_on) colsum(grp);
        const float* buf = smem + (size_t)(grp % WGT_RING) * WGT_BUF;
        float av[2][TMV], bv[2][TNV];
        auto load = [&](int slot, int s) {
#pragma unroll
            for (int x = 0; x < TMV; ++x) av[slot][x] = buf[oa + 4 * s * WGT_SP + 16 * x];
#pragma unroll
            for (int y = 0; y < TNV; ++y) bv[slot][y] = buf[ob + 4 * s * WGT_SQ + 16 * y];
        };
        load(0, 0);
#pragma unroll
        for (int s = 0; s < WGT_R / 4; ++s) {
            const int cur = s & 1;
            if (s + 1 < WGT_R / 4) load(cur ^ 1, s + 1);
#pragma unroll
            for (int y = 0; y < TNV; ++y)
#pragma unroll
                for (int x = 0; x < TMV; ++x)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][x], bv[cur][y], acc[x][y], 0, 0, 0);
        }
    }
    // accumulator (x, y), component r of lane (g, i):  out[16 (P tile) + 4 g + r][16 (Q tile) + i]
    const int PP = a.MT * 16, QP = a.NT * 16;
    float* out = a.partial + (size_t)chunk * PP * QP;
#pragma unroll
    for (int x = 0; x < TMV; ++x) {
        if (x >= tm) break;
#pragma unroll
        for (int y = 0; y < TNV; ++y) {
            if (y >= tn) break;
            float* o = out + (size_t)(16 * (p_tile0 + pl0 + x) + 4 * g) * QP + 16 * (q_tile0 + ql0 + y) + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(size_t)r * QP] = acc[x][y][r];
        }
    }
    if (BIAS == 1 && bias_on) a.psum[(size_t)chunk * PP + 16 * p_tile0 + tid] = bsum;
    if (BIAS == 2 && bias_on) a.qsum[(size_t)chunk * QP + 16 * q_tile0 + tid] = bsum;
}

// BIAS: 0 = no column sums, 1 = of P (psum), 2 = of Q (qsum)
template <int TM, int TN, bool QSILU, int BIAS>
__global__ __launch_bounds__(512, 2) void k_wgrad_t16(WgtArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wgt_sm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // blocks b, b + 8, ... run on the same XCD (observed placement; speed only): the workgroup tiles of one row chunk are
    // consecutive there, so the operand they share comes out of that XCD's L2 after its first read
    const int ntile = a.nPT * a.nQT;
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int tile = k % ntile, chunk = (k / ntile) * 8 + xcd;
    if (chunk >= a.n_chunks) return;
    const int pt = tile / a.nQT, qt = tile % a.nQT;
    const int p_tile0 = wgt_start(a.MT, a.nPT, pt), p_tiles = wgt_size(a.MT, a.nPT, pt);
    const int q_tile0 = wgt_start(a.NT, a.nQT, qt), q_tiles = wgt_size(a.NT, a.nQT, qt);
    // waves w and w + 4 share a SIMD: wave w < 4 takes P part w x Q part 0 (the larger Q part), wave w + 4 takes P part (w + 2) % 4
    // x Q part 1, so that a SIMD gets one of the larger and one of the smaller P parts
    const int qw = wave >> 2, pw = wave < 4 ? wave : ((wave - 4 + 2) & 3);
    const int pl0 = wgt_start(p_tiles, 4, pw), tm = wgt_size(p_tiles, 4, pw);
    const int ql0 = wgt_start(q_tiles, 2, qw), tn = wgt_size(q_tiles, 2, qw);
    const bool want_ps = BIAS == 1 && a.psum != nullptr && qt == 0, want_qs = BIAS == 2 && a.qsum != nullptr && pt == 0;     // workgroup-uniform
    if (tm <= 0 || tn <= 0) {                       // nothing to own (tiny operands): stay in the barrier protocol with one dummy tile
        wgt_run<1, 1, QSILU, 0>(a, wgt_sm, wave, lane, chunk, p_tile0, p_tiles, q_tile0, q_tiles, 0, 0, 0, 0, false, false);
        return;
    }
    const bool bigm = tm == TM, bign = tn == TN;
    if (bigm && bign) wgt_run<TM, TN, QSILU, BIAS>(a, wgt_sm, wave, lane, chunk, p_tile0, p_tiles, q_tile0, q_tiles, pl0, tm, ql0, tn, want_ps, want_qs);
    else if (bigm) wgt_run<TM, (TN > 1 ? TN - 1 : 1), QSILU, BIAS>(a, wgt_sm, wave, lane, chunk, p_tile0, p_tiles, q_tile0, q_tiles, pl0, tm, ql0, tn, want_ps, want_qs);
    else if (bign) wgt_run<(TM > 1 ? TM - 1 : 1), TN, QSILU, BIAS>(a, wgt_sm, wave, lane, chunk, p_tile0, p_tiles, q_tile0, q_tiles, pl0, tm, ql0, tn, want_ps, want_qs);
    else wgt_run<(TM > 1 ? TM - 1 : 1), (TN > 1 ? TN - 1 : 1), QSILU, BIAS>(a, wgt_sm, wave, lane, chunk, p_tile0, p_tiles, q_tile0, q_tiles, pl0, tm, ql0, tn, want_ps, want_qs);
}
