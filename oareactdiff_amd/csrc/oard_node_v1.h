// oard_node_v1.h — per-layer node stages, second generation.
//
// A workgroup of WAVES wavefronts owns 16 nodes (one MFMA column block).  Activations of those 16 nodes
// live in LDS as "vectors": block b (16 features x 16 nodes) = 1 KiB stored lane-linear, exactly the
// register block layout of oard_engine.h, so a B operand is one conflict-free ds_read_b128 and an output
// tile is one ds_write_b128.  The output tiles of every dense layer are dealt round-robin to the waves
// (tile t -> wave t % WAVES), the layers are separated by __syncthreads(), and the weights (A operand)
// come straight from L2 — every workgroup reads the same chunks, so they stay L2-resident.
// 276 workgroups x 8 waves at N = 4416 instead of 276 single waves.
#pragma once
#include "oard_kernels.h"

OARD_DEV f4 lds_blk(const float* v, int b, int lane) { return *reinterpret_cast<const f4*>(v + b * 256 + lane * 4); }
OARD_DEV void lds_st(float* v, int t, int lane, f4 x) { *reinterpret_cast<f4*>(v + t * 256 + lane * 4) = x; }

// one output tile from an LDS-resident input vector; even / odd K blocks on two accumulators.
// The weight chunks come straight from L2 (latency ~1 us): they are requested a whole group (<= 13 chunks,
// 52 VGPRs) at a time BEFORE the first MFMA of the group - left to itself the compiler keeps only two
// loads in flight and the tile costs one L2 round trip per chunk pair (measured: 5-6 us per tile instead of ~2).
#ifndef OARD_NODE_PF
#define OARD_NODE_PF 13
#endif
template <int KB>
OARD_DEV f4 dense_tile_lds(const float* __restrict__ wp, int t, const float* in, int lane, f4 init) {
    const float* base = wp + ((size_t)t * KB * 64 + lane) * 4;
    f4 c0 = init, c1 = f4zero();
    constexpr int G = OARD_NODE_PF;
#pragma unroll
    for (int b0 = 0; b0 < KB; b0 += G) {
        f4 a[G];
#pragma unroll
        for (int i = 0; i < G; ++i)
            if (b0 + i < KB) a[i] = ld_f4(base + (size_t)(b0 + i) * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i)
            if (b0 + i < KB) {
                const f4 x = lds_blk(in, b0 + i, lane);
                if (i & 1) c1 = mma_chunk(a[i], x, c1);
                else c0 = mma_chunk(a[i], x, c0);
            }
    }
    return c0 + c1;
}

// LayerNorm statistics of an LDS vector (every wave computes them redundantly)
template <int HT, int H>
OARD_DEV void ln_stats_lds(const float* v, int lane, float& mean, float& rstd) {
    const int g = lane >> 4;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const f4 x = lds_blk(v, t, lane);
        const int f0 = 16 * t + 4 * g;
        s += (f0 + 0 < H ? x.x : 0.f) + (f0 + 1 < H ? x.y : 0.f) + (f0 + 2 < H ? x.z : 0.f) + (f0 + 3 < H ? x.w : 0.f);
    }
    mean = col_reduce(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const f4 x = lds_blk(v, t, lane);
        const int f0 = 16 * t + 4 * g;
        const float dx = x.x - mean, dy = x.y - mean, dz = x.z - mean, dw = x.w - mean;
        q += (f0 + 0 < H ? dx * dx : 0.f) + (f0 + 1 < H ? dy * dy : 0.f) + (f0 + 2 < H ? dz * dz : 0.f) +
             (f0 + 3 < H ? dw * dw : 0.f);
    }
    rstd = 1.0f / sqrtf(col_reduce(q) * (1.0f / H) + 1e-5f);
}
template <int H>
OARD_DEV f4 ln_apply(f4 x, float mean, float rstd, const float* gamma, const float* beta, int t, int lane) {
    const int f0 = 16 * t + 4 * (lane >> 4);
    f4 y = (x - mean) * rstd * ld_vec(gamma, t, lane) + ld_vec(beta, t, lane);
    y.x = f0 + 0 < H ? y.x : 0.f; y.y = f0 + 1 < H ? y.y : 0.f; y.z = f0 + 2 < H ? y.z : 0.f; y.w = f0 + 3 < H ? y.w : 0.f;
    return y;
}

struct NodeBlk {
    int lane, wave, g, n;
    bool valid;
};
// NPB = real nodes per workgroup (<= 16): fewer than 16 leaves MFMA columns empty but doubles the
// number of workgroups, which is what these latency-bound stages need at N = 4416
#ifndef OARD_NPB
#define OARD_NPB 16
#endif
OARD_DEV NodeBlk node_blk(int N) {
    NodeBlk b;
    b.lane = threadIdx.x & 63; b.wave = threadIdx.x >> 6; b.g = b.lane >> 4;
    const int c = blockIdx.x * OARD_NPB + (b.lane & 15);
    b.valid = (b.lane & 15) < OARD_NPB && c < N;
    b.n = b.valid ? c : N - 1;
    return b;
}

// =====================================================================================================
// s += pos_expansion(pos_prjt);  xh = LN_gcl(s);  P = W1a xh + b1;  Q = W1b xh      (see k_node_pre)
// =====================================================================================================
template <class D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_node_pre_v1(TopoDev tp, const float* __restrict__ wb, PackOff po,
                                                            LayerOff lo, const float* __restrict__ s,
                                                            const float* __restrict__ pp0, float* __restrict__ xh,
                                                            float* __restrict__ P, float* __restrict__ Q) {
    __shared__ __attribute__((aligned(16))) float sm[(D::PB + 2 * D::HT) * 256];
    float* hid = sm;                      // [PB]
    float* sv = sm + D::PB * 256;         // [HT]
    float* xv = sv + D::HT * 256;         // [HT]
    const NodeBlk nb = node_blk(tp.N);
    const float pp = pp0[nb.n];
    for (int b = nb.wave; b < D::PB; b += WAVES) {
        const int k0 = 16 * b + 4 * nb.g;
        const float* w = wb + po.pe0;
        f4 h;
        h.x = k0 + 0 < D::H2 ? silu1(w[(k0 + 0) * 3] * pp) : 0.f;
        h.y = k0 + 1 < D::H2 ? silu1(w[(k0 + 1) * 3] * pp) : 0.f;
        h.z = k0 + 2 < D::H2 ? silu1(w[(k0 + 2) * 3] * pp) : 0.f;
        h.w = k0 + 3 < D::H2 ? silu1(w[(k0 + 3) * 3] * pp) : 0.f;
        lds_st(hid, b, nb.lane, h);
    }
    __syncthreads();
    for (int t = nb.wave; t < D::HT; t += WAVES)
        lds_st(sv, t, nb.lane, dense_tile_lds<D::PB>(wb + po.pe1, t, hid, nb.lane, ld_blk(s, nb.n, D::HP, t, nb.lane)));
    __syncthreads();
    float mean, rstd;
    ln_stats_lds<D::HT, D::H>(sv, nb.lane, mean, rstd);
    for (int t = nb.wave; t < D::HT; t += WAVES) {
        const f4 y = ln_apply<D::H>(lds_blk(sv, t, nb.lane), mean, rstd, wb + lo.ln_g_w, wb + lo.ln_g_b, t, nb.lane);
        lds_st(xv, t, nb.lane, y);
        if (nb.valid) st_blk(xh, nb.n, D::HP, t, nb.lane, y);
    }
    __syncthreads();
    for (int t = nb.wave; t < 2 * D::HT; t += WAVES) {
        if (t < D::HT) {
            const f4 p = dense_tile_lds<D::HT>(wb + lo.W1a, t, xv, nb.lane, ld_vec(wb + lo.b1, t, nb.lane));
            if (nb.valid) st_blk(P, nb.n, D::HP, t, nb.lane, p);
        } else {
            const f4 q = dense_tile_lds<D::HT>(wb + lo.W1b, t - D::HT, xv, nb.lane, f4zero());
            if (nb.valid) st_blk(Q, nb.n, D::HP, t - D::HT, nb.lane, q);
        }
    }
}

// =====================================================================================================
// agg = mean_e m_e;  s = xh + node_mlp([xh, agg]);  xq = x_proj(LN_msg(s))           (see k_gcl_node)
// =====================================================================================================
template <class D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_gcl_node_v1(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                            const float* __restrict__ xh, const float* __restrict__ mbuf,
                                                            float* __restrict__ s, float* __restrict__ xq) {
    __shared__ __attribute__((aligned(16))) float sm[4 * D::HT * 256];
    float* in = sm;                        // [2 HT]: xh | agg      (later: xln | hq)
    float* hm = sm + 2 * D::HT * 256;      // [HT]
    float* sv = sm + 3 * D::HT * 256;      // [HT]
    const NodeBlk nb = node_blk(tp.N);
    const int smp = tp.node_sample[nb.n];
    const int deg = tp.sample_ptr[smp + 1] - tp.sample_ptr[smp] - 1;
    const size_t e0 = (size_t)tp.edge_ptr[nb.n];
    const int mx = wave_max(deg);
    const float inv = 1.0f / (float)max(deg, 1);
    for (int t = nb.wave; t < D::HT; t += WAVES) {
        lds_st(in, t, nb.lane, ld_blk(xh, nb.n, D::HP, t, nb.lane));
        f4 a0 = f4zero(), a1 = f4zero();
        // 8 message rows in flight per step (branch-free: out-of-range slots re-read the last row with weight 0)
        const int last = max(deg - 1, 0);
#ifdef OARD_ABL_NOGATHER
        for (int k = 0; k < 0; k += 8) {
#else
        for (int k = 0; k < mx; k += 8) {
#endif
            f4 r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = ld_blk(mbuf, e0 + min(k + i, last), D::HP, t, nb.lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f4 v = k + i < deg ? r[i] : f4zero();
                if (i & 1) a1 += v; else a0 += v;
            }
        }
        lds_st(in, D::HT + t, nb.lane, (a0 + a1) * inv);
    }
    __syncthreads();
    for (int t = nb.wave; t < D::HT; t += WAVES)
        lds_st(hm, t, nb.lane, silu4(dense_tile_lds<2 * D::HT>(wb + lo.nm0, t, in, nb.lane, ld_vec(wb + lo.nm0b, t, nb.lane))));
    __syncthreads();
    for (int t = nb.wave; t < D::HT; t += WAVES) {
        const f4 v = lds_blk(in, t, nb.lane) +
                     dense_tile_lds<D::HT>(wb + lo.nm1, t, hm, nb.lane, ld_vec(wb + lo.nm1b, t, nb.lane));
        lds_st(sv, t, nb.lane, v);
        if (nb.valid) st_blk(s, nb.n, D::HP, t, nb.lane, v);
    }
    __syncthreads();
    float mean, rstd;
    ln_stats_lds<D::HT, D::H>(sv, nb.lane, mean, rstd);
    float* xln = in;                       // safe: `in` is no longer read after the barrier above
    float* hq = in + D::HT * 256;
    for (int t = nb.wave; t < D::HT; t += WAVES)
        lds_st(xln, t, nb.lane, ln_apply<D::H>(lds_blk(sv, t, nb.lane), mean, rstd, wb + lo.ln_q_w, wb + lo.ln_q_b, t, nb.lane));
    __syncthreads();
    for (int t = nb.wave; t < D::HT; t += WAVES)
        lds_st(hq, t, nb.lane, silu4(dense_tile_lds<D::HT>(wb + lo.xp0, t, xln, nb.lane, f4zero())));
    __syncthreads();
    for (int t = nb.wave; t < 3 * D::HT; t += WAVES) {
        const f4 o = dense_tile_lds<D::HT>(wb + lo.xp2, t, hq, nb.lane, f4zero());
        if (nb.valid) st_blk(xq, nb.n, 3 * D::HP, t, nb.lane, o);
    }
}

// =====================================================================================================
// EquiMessage node side + EquiUpdate, fused (see k_equi_agg_v1 / k_equi_upd):
//   messages from q, aggregation, s = (s + dx)/sqrt2, vec += dvec, vec_proj, frame scalar MLP,
//   xvec_proj, s += (a + b + vdot)/sqrt2, vec += c * vec2.        vec_in != vec_out.
// =====================================================================================================
template <class D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_equi_node_v1(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                             const float* __restrict__ qbuf, const float* __restrict__ xq,
                                                             const float* __restrict__ geo, const float* __restrict__ x1,
                                                             float* __restrict__ s, const float* __restrict__ vec_in,
                                                             float* __restrict__ vec_out) {
    constexpr int HT = D::HT;
    constexpr int TPW = (HT + WAVES - 1) / WAVES;            // tiles owned per wave (upper bound)
    __shared__ __attribute__((aligned(16))) float sm[6 * HT * 256];
    float* vx = sm;                        // [3][HT] updated vec
    float* in = sm + 3 * HT * 256;         // [2 HT]: s_mid | scal
    float* hx = sm + 5 * HT * 256;         // [HT]
    const NodeBlk nb = node_blk(tp.N);
    const int n = nb.n, a0 = tp.act_ptr[n], cnt = tp.act_ptr[n + 1] - a0;
    const int mx = wave_max(cnt);
    const float inv_sqrt2 = 0.70710678118654752f, inv_sqrt3 = 0.57735026918962576f,
                inv_sqrt_h = 1.0f / sqrtf((float)D::H);

    // 1. messages + aggregation for the tiles this wave owns
    for (int t = nb.wave; t < HT; t += WAVES) {
        f4 dx = f4zero(), v0 = f4zero(), v1 = f4zero(), v2 = f4zero();
        const f4 xn0 = ld_blk(xq, n, 3 * D::HP, t, nb.lane), xn1 = ld_blk(xq, n, 3 * D::HP, HT + t, nb.lane),
                 xn2 = ld_blk(xq, n, 3 * D::HP, 2 * HT + t, nb.lane);
        // two edges in flight per step, branch-free (out-of-range slots re-read a valid edge and are discarded)
        const long long a_hi = max(tp.A - 1, 0LL);
#ifdef OARD_ABL_NOGATHER
        for (int k = 0; k < 0; k += 2) {
#else
        for (int k = 0; k < mx; k += 2) {
#endif
            f4 q0[2], q1[2], q2[2], y0[2], y1[2], y2[2], w0[2], w1[2], w2[2];
            float gx[2], gy[2], gz[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const size_t a = (size_t)min((long long)a0 + min(k + i, max(cnt - 1, 0)), a_hi);
                const int m = tp.act_src[a];
                const float* g = geo + a * GEO_STRIDE;
                gx[i] = g[2]; gy[i] = g[3]; gz[i] = g[4];
                q0[i] = ld_blk(qbuf, a, 3 * D::HP, t, nb.lane); q1[i] = ld_blk(qbuf, a, 3 * D::HP, HT + t, nb.lane);
                q2[i] = ld_blk(qbuf, a, 3 * D::HP, 2 * HT + t, nb.lane);
                y0[i] = ld_blk(xq, m, 3 * D::HP, t, nb.lane); y1[i] = ld_blk(xq, m, 3 * D::HP, HT + t, nb.lane);
                y2[i] = ld_blk(xq, m, 3 * D::HP, 2 * HT + t, nb.lane);
                w0[i] = ld_blk(vec_in, (size_t)m * 3 + 0, D::HP, t, nb.lane);
                w1[i] = ld_blk(vec_in, (size_t)m * 3 + 1, D::HP, t, nb.lane);
                w2[i] = ld_blk(vec_in, (size_t)m * 3 + 2, D::HP, t, nb.lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (k + i < cnt) {
                    dx += (y0[i] + xn0) * q0[i];
                    const f4 a2 = (y1[i] + xn1) * q1[i] * inv_sqrt3;
                    const f4 a3 = (y2[i] + xn2) * q2[i];
                    v0 += (w0[i] * a2 + a3 * gx[i]) * inv_sqrt_h;
                    v1 += (w1[i] * a2 + a3 * gy[i]) * inv_sqrt_h;
                    v2 += (w2[i] * a2 + a3 * gz[i]) * inv_sqrt_h;
                }
        }
        lds_st(in, t, nb.lane, (ld_blk(s, n, D::HP, t, nb.lane) + dx) * inv_sqrt2);
        lds_st(vx + 0 * HT * 256, t, nb.lane, v0 + ld_blk(vec_in, (size_t)n * 3 + 0, D::HP, t, nb.lane));
        lds_st(vx + 1 * HT * 256, t, nb.lane, v1 + ld_blk(vec_in, (size_t)n * 3 + 1, D::HP, t, nb.lane));
        lds_st(vx + 2 * HT * 256, t, nb.lane, v2 + ld_blk(vec_in, (size_t)n * 3 + 2, D::HP, t, nb.lane));
    }
    __syncthreads();

    // 2. vec_proj for owned tile pairs (t, HT + t), frame scalar, vdot; keep vec2 / vdot in registers
    const float fx = x1[n * 3], fy = x1[n * 3 + 1], fz = x1[n * 3 + 2];
    const float* l3 = wb + lo.l3u;
    f4 v2k[TPW][3], vdk[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = nb.wave + i * WAVES;
        if (t < HT) {
            f4 v1[3], v2[3];
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                v1[x] = dense_tile_lds<HT>(wb + lo.vp, t, vx + x * HT * 256, nb.lane, f4zero());
                v2[x] = dense_tile_lds<HT>(wb + lo.vp, HT + t, vx + x * HT * 256, nb.lane, f4zero());
                v2k[i][x] = v2[x];
            }
            const f4 sc = v1[0] * fx + v1[1] * fy + v1[2] * fz;
            vdk[i] = (v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2]) * inv_sqrt_h;
            f4 sca;
            const int f0 = 16 * t + 4 * nb.g;
#ifdef OARD_ABL_NOLIN3U
            sca = sc;
#else
            sca.x = f0 + 0 < D::H ? lin3u(l3, sc.x) : 0.f;
            sca.y = f0 + 1 < D::H ? lin3u(l3, sc.y) : 0.f;
            sca.z = f0 + 2 < D::H ? lin3u(l3, sc.z) : 0.f;
            sca.w = f0 + 3 < D::H ? lin3u(l3, sc.w) : 0.f;
#endif
            lds_st(in, HT + t, nb.lane, sca);
        }
    }
    __syncthreads();

    // 3. xvec_proj hidden
    for (int t = nb.wave; t < HT; t += WAVES)
        lds_st(hx, t, nb.lane, silu4(dense_tile_lds<2 * HT>(wb + lo.xv0, t, in, nb.lane, f4zero())));
    __syncthreads();

    // 4. outputs for owned tiles
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = nb.wave + i * WAVES;
        if (t < HT) {
            const f4 a = dense_tile_lds<HT>(wb + lo.xv2, t, hx, nb.lane, f4zero());
            const f4 b = dense_tile_lds<HT>(wb + lo.xv2, HT + t, hx, nb.lane, f4zero());
            const f4 c = dense_tile_lds<HT>(wb + lo.xv2, 2 * HT + t, hx, nb.lane, f4zero());
            if (nb.valid) {
                st_blk(s, n, D::HP, t, nb.lane, lds_blk(in, t, nb.lane) + (a + b + vdk[i]) * inv_sqrt2);
#pragma unroll
                for (int x = 0; x < 3; ++x)
                    st_blk(vec_out, (size_t)n * 3 + x, D::HP, t, nb.lane,
                           lds_blk(vx + x * HT * 256, t, nb.lane) + c * v2k[i][x]);
            }
        }
    }
}
