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
//     into a ring of FOUR 37-KB buffers, three groups ahead.  One barrier per group (~10 000 MFMA cycles per SIMD), and it sits
//     in the MIDDLE of a group's MFMA steps: it publishes data that is needed two steps later at the earliest, so the operands of
//     a group's first step are read from LDS during the last step of the group before (no barrier, no LDS latency at a group's
//     start), and a wave that waits at the barrier leaves its SIMD partner a full queue of MFMAs;
//   * SiLU-on-load (X = SiLU(z), two of the five products per layer) is applied IN LDS, once per element, two groups ahead
//     (after the barrier that saw the group land, published by the next one);
//   * the bias gradient (column sums of dY) costs nothing: X always has a spare pad column (196 -> 208, 588 -> 592, 684 -> 688);
//     it is overwritten by 1.0 in LDS, and the product's padded output column IS db.
// Both operands of an MFMA step are plain ds_read_b32 (lane (g, i): row 4 s + g, feature 16 t + i of the panel; the panel's
// row strides are = 16 mod 32 floats, so the two rows a 32-lane half reads sit in different banks): 13 reads per 42 MFMAs,
// requested one step ahead.
// Deterministic: fixed chunking, per-chunk partials, ONE fixed-order reduce pass.  Layout of the partials: TILE-MAJOR -
// partial[(chunk * MT + p_tile) * NT + q_tile] is a 1-KiB image of the accumulator registers of that 16 x 16 tile (lane (g, i),
// register r: P feature 16 p_tile + 4 g + r, Q feature 16 q_tile + i), one coalesced store per tile.  This is NOT k_wgrad's
// [chunk][PP][QP] layout: the only valid reducer is k_wgt_reduce (below), which sums the chunks in order and scatters dW and -
// from the padded output column of the ones trick - db; handing these partials to k_wgrad_reduce / k_bgrad_reduce gives wrong
// gradients without any error.
// Precondition on the operands: their PAD columns (196 -> 208, 588 -> 592, 684 -> 688) are read and multiplied like any other
// column (their products land in output rows / columns the reduce never reads, or - the ones column - are overwritten in LDS),
// so they must hold FINITE values: a NaN / Inf bit pattern there would not reach dW, but 0 x Inf in the same MFMA would.  Every
// producer in oard_train_stages.h writes zero pads; wgrad_impl's callers own that invariant.
#pragma once
#include "oard_edge_bwd.h"

// timing-only ablations of an experiment build (results are garbage): -DOARD_WGT_ABL=1 no vmcnt wait at the group barrier, 2 no
// barrier, 3 neither, 4 no LDS-DMA issue in the loop
#ifndef OARD_WGT_ABL
#define OARD_WGT_ABL 0
#endif
OARD_DEV void wgt_barrier() {
    if (OARD_WGT_ABL != 1 && OARD_WGT_ABL != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (OARD_WGT_ABL != 2 && OARD_WGT_ABL != 3) __syncthreads();
}
#define WGT_R 16                           // rows per group
#define WGT_PT 22                          // P tiles per workgroup at most (4 waves x <= 6)
#define WGT_QT 13                          // Q tiles per workgroup at most (2 waves x <= 7)
#define WGT_SP (16 * WGT_PT + (WGT_PT % 2 ? 0 : 16))       // P panel row stride (floats), = 16 mod 32: the two rows a 32-lane half of a
#define WGT_SQ (16 * WGT_QT + (WGT_QT % 2 ? 0 : 16))       // ds_read_b32 touches sit 16 banks apart (Q: 208, P: 368)
#define WGT_BUF (WGT_R * (WGT_SP + WGT_SQ))        // floats per ring buffer (37 KB)
#define WGT_RING 4
#define WGT_LDS_BYTES ((size_t)WGT_RING * WGT_BUF * sizeof(float))

struct WgtArgs {
    const float* P; const float* Q;        // row-major operands; columns [0, 16 MT) / [0, 16 NT) are readable
    int ldP, ldQ;
    int MT, NT;                            // 16-feature tiles of P / Q
    int nPT, nQT;                          // workgroup tiles along P / Q (a workgroup: <= WGT_PT x WGT_QT tiles)
    long long r0, r1, rpc;                 // rows [r0, r1), rows per chunk (multiple of WGT_R)
    int n_chunks;
    int ones_side, ones_col;               // 1 / 2: column ones_col of P / Q reads as 1.0 (bias gradient through X's pad column); 0: none
    float* partial;                        // [n_chunks][MT][NT] tile images of 256 floats (tile-major; reducer: k_wgt_reduce ONLY)
};

// part k of `n` tiles cut into `parts` nearly equal pieces (the larger pieces first): [start, start + size)
__host__ __device__ inline int wgt_start(int n, int parts, int k) { return k * (n / parts) + (k < n % parts ? k : n % parts); }
__host__ __device__ inline int wgt_size(int n, int parts, int k) { return n / parts + (k < n % parts ? 1 : 0); }

template <int TMV, int TNV, bool QSILU>
OARD_DEV void wgt_run(const WgtArgs& a, float* smem, int wave, int lane, int chunk, int p_tile0, int p_tiles, int q_tile0, int q_tiles,
                      int pl0, int tm, int ql0, int tn) {
    // p_tile0 / p_tiles: this WORKGROUP's P tiles (absolute first tile, count); pl0 / tm: this WAVE's first tile inside them and its
    // tile count (tm <= TMV; with tm < TMV the surplus accumulators read whatever follows the wave's last tile - inside the LDS
    // allocation - and are never stored)
    const int g = lane >> 4, i = lane & 15, tid = wave * 64 + lane;
    const long long rb = a.r0 + (long long)chunk * a.rpc;
    const long long re = rb + a.rpc < a.r1 ? rb + a.rpc : a.r1;
    const int ng = (int)((re - rb + WGT_R - 1) / WGT_R);
    const int wQ4 = q_tiles * 4;                                     // float4 per Q panel row
    auto ring = [&](int grp) -> float* { return smem + (size_t)(grp & (WGT_RING - 1)) * WGT_BUF; };

    // LDS-DMA: a ring buffer is NPP + NPQ = 36 pieces of 1 KiB (64 lanes x 16 bytes, contiguous in LDS; the panels' row strides make
    // both panels whole numbers of pieces), piece j issued by wave j % 8.  A lane's 16 bytes lie in panel row o / rowbytes at byte
    // o % rowbytes of the row; lanes that fall into the stride padding or beyond this workgroup's tiles fetch the row's first bytes
    // again (never read).  Source = uniform group address (SGPR pair) + a per-lane 32-bit offset that is the same for every full
    // group, so a piece costs no VALU work and no exec masking (measured: the 48 masked per-row pieces of the first version cost
    // 6.5 % of the matrix pipe's time, ~50 idle cycles per piece and SIMD).  The last group of a chunk clamps the row to the last
    // valid one; its surplus rows are zeroed by the fix-up pass.
    constexpr int NPP = WGT_R * WGT_SP * 4 / 1024, NPQ = WGT_R * WGT_SQ * 4 / 1024, NPW = (NPP + NPQ + 7) / 8;
    static_assert(NPP * 1024 == WGT_R * WGT_SP * 4 && NPQ * 1024 == WGT_R * WGT_SQ * 4, "panels must be whole LDS-DMA pieces");
    auto piece_off = [&](int k, int maxrow) -> unsigned {             // per-lane source byte offset of this wave's k-th piece
        const int j = wave + 8 * k;
        const bool isP = j < NPP;
        const int o = (isP ? j : j - NPP) * 1024 + lane * 16, rowbytes = isP ? WGT_SP * 4 : WGT_SQ * 4;
        int row = o / rowbytes, colb = o - row * rowbytes;
        if (colb >= 64 * (isP ? p_tiles : q_tiles)) colb = 0;
        if (row > maxrow) row = maxrow;
        return (unsigned)(row * (isP ? a.ldP : a.ldQ) * 4 + colb);
    };
    unsigned voff[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) voff[k] = piece_off(k, WGT_R - 1);
    auto issue = [&](int grp) {
        float* buf = ring(grp);
        const long long row0 = rb + (long long)grp * WGT_R;
        const int nvalid = (int)(re - row0 < WGT_R ? re - row0 : WGT_R);          // wave-uniform
        const float* bP = a.P + (size_t)row0 * a.ldP + (size_t)p_tile0 * 16;
        const float* bQ = a.Q + (size_t)row0 * a.ldQ + (size_t)q_tile0 * 16;
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int j = wave + 8 * k;
            if (j >= NPP + NPQ) break;
            const unsigned vo = nvalid == WGT_R ? voff[k] : piece_off(k, nvalid - 1);
            glds16u(j < NPP ? bP : bQ, vo, buf + j * 256);            // Q panel starts right behind the P panel: piece j at byte 1024 j
        }
    };
    // In-LDS fix-ups of a landed group, once per element: SiLU on the Q panel (float4 number tid, tid + 512 of its WGT_R x wQ4
    // grid), the ones column (X's pad column := 1.0, so that the padded output column of the product is the bias gradient) and, in
    // the last group of a chunk, zeros in the rows beyond the chunk.  Every element has ONE writer: the ones column of a SiLU
    // operand is set by the thread that rewrites that float4, and surplus rows are left to the zero fill.
    const int op_local = a.ones_side == 1 ? a.ones_col - 16 * p_tile0 : -1, oq_local = a.ones_side == 2 ? a.ones_col - 16 * q_tile0 : -1;
    const bool ones_p = op_local >= 0 && op_local < 16 * p_tiles, ones_q = oq_local >= 0 && oq_local < 16 * q_tiles;
    int so[2], sone[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + 512 * k, c4 = idx % wQ4;
        so[k] = (QSILU && idx < WGT_R * wQ4) ? (idx / wQ4) * WGT_SQ + 4 * c4 : -1;
        sone[k] = (ones_q && c4 == (oq_local >> 2)) ? (oq_local & 3) : -1;
    }
    auto fixup = [&](int grp) {
        float* pp = ring(grp);
        float* q = pp + WGT_R * WGT_SP;
        const long long left = re - (rb + (long long)grp * WGT_R);
        const int nvalid = (int)(left < WGT_R ? left : WGT_R);       // workgroup-uniform; < WGT_R in the last group only
        if (QSILU) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (so[k] >= 0 && so[k] < nvalid * WGT_SQ) {
                    f4 v = silu4(ld_f4(q + so[k]));
                    if (sone[k] == 0) v.x = 1.0f; else if (sone[k] == 1) v.y = 1.0f; else if (sone[k] == 2) v.z = 1.0f; else if (sone[k] == 3) v.w = 1.0f;
                    st_f4(q + so[k], v);
                }
        } else if (ones_q && tid < nvalid) q[tid * WGT_SQ + oq_local] = 1.0f;
        if (ones_p && tid < nvalid) pp[tid * WGT_SP + op_local] = 1.0f;
        if (nvalid < WGT_R) {
            for (int c = nvalid * WGT_SP + tid; c < WGT_R * WGT_SP; c += 512) pp[c] = 0.f;
            for (int c = nvalid * WGT_SQ + tid; c < WGT_R * WGT_SQ; c += 512) q[c] = 0.f;
        }
    };

    f4 acc[TMV][TNV];
#pragma unroll
    for (int x = 0; x < TMV; ++x)
#pragma unroll
        for (int y = 0; y < TNV; ++y) acc[x][y] = f4zero();
    // per-lane read offset (floats) inside a buffer; tile x / y of this wave is a compile-time displacement from it
    const int oa = g * WGT_SP + 16 * pl0 + i, ob = WGT_R * WGT_SP + g * WGT_SQ + 16 * ql0 + i;
    // One MFMA step with the NEXT step's operand reads dealt out between its MFMAs.  The MFMAs are volatile asm statements: (1) the
    // accumulator is updated in place ("+v" - with the builtin hipcc writes many results to fresh registers and copies, which cost
    // 39 spilled registers here), (2) volatile asm and memory operations keep their program order, so the LDS reads below stay exactly
    // where they are written - a wave never waits for LDS with an empty MFMA queue.  Hazards the compiler cannot see inside the asm:
    // none in this loop (A / B come from LDS - s_waitcnt, which the compiler does insert for asm operands; an accumulator is touched
    // once per 30+ MFMAs); the epilogue puts two s_nop 15 in front of its stores.
    float av[2][TMV], bv[2][TNV];
    auto rd = [&](int slot, const float* buf, int s, int k) {        // operand number k of step s: the TMV A tiles first, then the TNV B tiles
        if (k < TMV) av[slot][k] = buf[oa + 4 * s * WGT_SP + 16 * k];
        else bv[slot][k - TMV] = buf[ob + 4 * s * WGT_SQ + 16 * (k - TMV)];
    };
    auto load = [&](int slot, int grp, int s) {
        const float* buf = ring(grp);
#pragma unroll
        for (int k = 0; k < TMV + TNV; ++k) rd(slot, buf, s, k);
    };
    // step on operand set `slot`; if `next`: set slot ^ 1 := operands of (ngrp, ns), one LDS read after every PER MFMAs
    auto step = [&](int slot, bool next, int ngrp, int ns) {
        constexpr int NR = TMV + TNV, NM = TMV * TNV, PER = NM / NR > 0 ? NM / NR : 1;
        const float* nbuf = ring(ngrp);
        int m = 0, k = 0;
#pragma unroll
        for (int y = 0; y < TNV; ++y)
#pragma unroll
            for (int x = 0; x < TMV; ++x) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[x][y]) : "v"(av[slot][x]), "v"(bv[slot][y]));
                ++m;
                if (m % PER == 0 && k < NR) { if (next) rd(slot ^ 1, nbuf, ns, k); ++k; }
            }
#pragma unroll
        for (; k < NR; ++k) if (next) rd(slot ^ 1, nbuf, ns, k);
    };

    // ---- prologue: three groups in flight, the first two fixed up ------------------------------------------------------------------
    issue(0);
    if (ng > 1) issue(1);
    if (ng > 2) issue(2);
    phase_barrier();
    fixup(0);
    if (ng > 1) fixup(1);
    __syncthreads();
    load(0, 0, 0);
    // ---- one group per iteration: steps 0, 1 | barrier | steps 2, 3 -----------------------------------------------------------------
    // barrier of group grp (every wave has waited for its own LDS-DMA pieces before it): group grp + 3's buffer - read last in group
    // grp - 1 - is free, group grp + 2 has landed (fixed up behind the barrier, published by the next one), the fix-ups of group
    // grp + 1 are visible (its first operands are read during step 3 below).  Waves 0..3 issue their share of group grp + 3 right behind
    // the barrier, waves 4..7 one step later: the two waves of a SIMD never sit in their DMA issue at the same time.
    for (int grp = 0; grp < ng; ++grp) {
        step(0, true, grp, 1);
        step(1, true, grp, 2);
        wgt_barrier();
        if (OARD_WGT_ABL != 4 && wave < 4 && grp + 3 < ng) issue(grp + 3);
        if (grp + 2 < ng) fixup(grp + 2);
        step(0, true, grp, 3);
        if (OARD_WGT_ABL != 4 && wave >= 4 && grp + 3 < ng) issue(grp + 3);
        step(1, grp + 1 < ng, grp + 1, 0);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                // MFMA results -> stores (see above)
    // partials are kept TILE-MAJOR: tile (P tile, Q tile) of a chunk = the accumulator's register image, one float4 per lane (lane
    // (g, i): rows 4 g .. 4 g + 3 of the tile, column i), 1 KiB, one coalesced store; k_wgt_reduce undoes the layout
    float* out = a.partial + (size_t)chunk * a.MT * a.NT * 256 + lane * 4;
#pragma unroll
    for (int x = 0; x < TMV; ++x) {
        if (x >= tm) break;
#pragma unroll
        for (int y = 0; y < TNV; ++y) {
            if (y >= tn) break;
            st_f4(out + ((size_t)(p_tile0 + pl0 + x) * a.NT + (q_tile0 + ql0 + y)) * 256, acc[x][y]);
        }
    }
}

// TM x TN = the largest wave tile of the launch; a wave owns TM or TM - 1 (or fewer) P tiles and TN or TN - 1 (or fewer) Q tiles
template <int TM, int TN, bool QSILU>
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
    if (tm <= 0 || tn <= 0) {                       // nothing to own (tiny operands): stay in the barrier protocol with one dummy tile
        wgt_run<1, 1, QSILU>(a, wgt_sm, wave, lane, chunk, p_tile0, p_tiles, q_tile0, q_tiles, 0, 0, 0, 0);
        return;
    }
    const bool bigm = tm == TM, bign = tn == TN;
    if (bigm && bign) wgt_run<TM, TN, QSILU>(a, wgt_sm, wave, lane, chunk, p_tile0, p_tiles, q_tile0, q_tiles, pl0, tm, ql0, tn);
    else if (bigm) wgt_run<TM, (TN > 1 ? TN - 1 : 1), QSILU>(a, wgt_sm, wave, lane, chunk, p_tile0, p_tiles, q_tile0, q_tiles, pl0, tm, ql0, tn);
    else if (bign) wgt_run<(TM > 1 ? TM - 1 : 1), TN, QSILU>(a, wgt_sm, wave, lane, chunk, p_tile0, p_tiles, q_tile0, q_tiles, pl0, tm, ql0, tn);
    else wgt_run<(TM > 1 ? TM - 1 : 1), (TN > 1 ? TN - 1 : 1), QSILU>(a, wgt_sm, wave, lane, chunk, p_tile0, p_tiles, q_tile0, q_tiles, pl0, tm, ql0, tn);
}

// Second pass: dW (logical nn.Linear shape, row stride ldW) and db from the tile-major partials, chunks in ascending order.  One
// workgroup per 16 x 16 tile: 4 chunk slices x 64 lanes, a lane sums its float4 (rows 4 g .. 4 g + 3 of the tile, column i) over
// the chunks sl, sl + 4, ... (coalesced 1-KiB reads), the four slice sums are added in slice order - a fixed order.  Padded index
// -> logical index by sections (undoes 196 -> 208 between the thirds of split projections); pads are dropped.  `transposed`: the
// product was formed with P = X, Q = dY.  db = the padded column (row, when transposed) `ones_col` of the product.
OARD_DEV int wgt_logical(int padded, int len, int pad, int n) {
    const int sec = padded / pad, w = padded - sec * pad, l = sec * len + w;
    return (w < len && l < n) ? l : -1;
}
OARD_KERNEL __global__ __launch_bounds__(256) void k_wgt_reduce(const float* __restrict__ partial, int n_chunks, int MT, int NT, int transposed, int o_len, int o_pad,
                                                    int MO, int i_len, int i_pad, int MI, float* __restrict__ dW, int ldW, float* __restrict__ db,
                                                    int ones_col, int acc) {
    __shared__ f4 red[4][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6, tile = blockIdx.x;
    const size_t cs = (size_t)MT * NT * 256;
    const float* p = partial + (size_t)tile * 256 + lane * 4;
    f4 v = f4zero();
    for (int ch = sl; ch < n_chunks; ch += 4) v += ld_f4(p + (size_t)ch * cs);
    red[sl][lane] = v;
    __syncthreads();
    if (sl != 0) return;
    v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    const int tp = tile / NT, tq = tile - tp * NT, g = lane >> 4, j = lane & 15;
    const int qq = 16 * tq + j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pp = 16 * tp + 4 * g + r;
        const int o = transposed ? wgt_logical(qq, o_len, o_pad, MO) : wgt_logical(pp, o_len, o_pad, MO);
        if (o < 0) continue;
        const int xp = transposed ? pp : qq;                        // padded X column of this element
        if (db != nullptr && xp == ones_col) db[o] = acc ? db[o] + v[r] : v[r];
        const int i = wgt_logical(xp, i_len, i_pad, MI);
        if (dW != nullptr && i >= 0) {
            float* d = dW + (size_t)o * ldW + i;
            *d = acc ? *d + v[r] : v[r];
        }
    }
}
