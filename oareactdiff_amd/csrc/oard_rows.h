// oard_rows.h — row-generic building blocks of the node-side backward pass (training path, row N2).
//
// The node-side stages of LEFTNet (EquiUpdate, the GCL node MLP, x_proj, pos_expansion + the node halves of edge_mlp.0, the
// output block, the init head: < 2 % of the FLOPs) act on O(N) rows of H-wide features.  Their adjoints are compositions of
//   * a dense layer on rows:   Y = epi(W X + b)                      k_rows_dense      (fp32 MFMA column engine, 16 rows / workgroup)
//   * its transpose:           dX = W^T dY (x SiLU'(z) | + residual)  the same kernel on a transposed weight pack
//   * LayerNorm forward / backward on rows                            k_rows_ln_fwd / k_rows_ln_bwd
//   * fixed-order column sums over rows (bias / LayerNorm / gate gradients)   k_colsum_part + k_colsum_fin
// plus a few bespoke element-wise kernels per stage (oard_train_stages.h).  Everything is row-major [rows][ld] fp32 with the
// feature pads (196 -> 208) kept at exactly 0: a dense layer's padded output rows have zero weights and zero bias, SiLU(0) = 0,
// and SiLU'(0) x 0 = 0, so pads never need masking after the first producer (the tape's buffers come with zero pads).
// Weight gradients are NOT formed here: every stage writes its (dY, X) operand pairs row-major and the weight-gradient GEMM
// k_wgrad (oard_edge_bwd.h) contracts them over the rows, as for the edge stages.
#pragma once
#include "oard_node_v1.h"
#include "oard_edge_bwd.h"
#include "oard_node_bwd.h"

enum { EPI_NONE = 0, EPI_SILU = 1, EPI_MUL_DSILU = 2, EPI_ADD = 3 };

struct RowsDense {
    const float* X; int ldx;        // input rows: K blocks 0 .. KB1-1 are X[r][16 b ..]
    const float* X2; int ldx2;      // K blocks KB1 .. KB-1 are X2[r][16 (b - KB1) ..]   (concatenated inputs); unused if KB1 == KB
    int KB1;
    const float* W;                 // packed chunks [MT][KB] (oard_pack_weights / oard_pack_weights_bwd)
    const float* bias;              // [16 MT] or nullptr
    float* Y; int ldy;              // Y[r][16 t ..]
    const float* Z; int ldz;        // EPI_MUL_DSILU: pre-activation z (Y = acc * SiLU'(z));  EPI_ADD: addend (Y = Z + scale * acc)
    float* Zo; int ldzo;            // EPI_SILU: optional copy of the pre-activation (Zo = acc + bias), Y = SiLU(Zo)
    long long rows;
    int MT;
    float scale;
};

template <int KB, int EPI, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_rows_dense(RowsDense a) {
    __shared__ __attribute__((aligned(16))) float xin[KB * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r = (long long)blockIdx.x * 16 + (lane & 15);
    const bool valid = r < a.rows;
    const size_t row = (size_t)(valid ? r : a.rows - 1);
    for (int b = wave; b < KB; b += WAVES) {
        const f4 v = b < a.KB1 ? ld_blk(a.X, row, a.ldx, b, lane) : ld_blk(a.X2, row, a.ldx2, b - a.KB1, lane);
        lds_st(xin, b, lane, v);
    }
    __syncthreads();
    for (int t = wave; t < a.MT; t += WAVES) {
        f4 acc = a.bias != nullptr ? ld_vec(a.bias, t, lane) : f4zero();
        acc = dense_tile_lds<KB>(a.W, t, xin, lane, acc);
        if (!valid) continue;
        if (EPI == EPI_SILU) {
            if (a.Zo != nullptr) st_blk(a.Zo, row, a.ldzo, t, lane, acc);
            acc = silu4(acc);
        } else if (EPI == EPI_MUL_DSILU) {
            acc = acc * dsilu4(ld_blk(a.Z, row, a.ldz, t, lane));
        } else if (EPI == EPI_ADD) {
            acc = ld_blk(a.Z, row, a.ldz, t, lane) + acc * a.scale;
        }
        st_blk(a.Y, row, a.ldy, t, lane, acc);
    }
}

// Two layers on the same 16 rows in ONE launch (round 4): Y1 = epi1(W1 X) as k_rows_dense does (stored - the weight-gradient GEMMs
// read it), kept in LDS as the input of up to two second layers  Y2_j = epi2_j(W2_j Y1),  epi2 in {none, add}.  The node adjoints are chains
// of such pairs (dz = (W^T dy) SiLU'(z); dx = V^T dz [+ residual]); every fused pair saves a ~15-us launch and the 8-us dispatch gap behind it.
struct RowsOut2 { const float* W; float* Y; int ldy; int MT; const float* Zadd; int ldz; };     // Zadd != nullptr: Y = Zadd + acc
struct RowsDense2 { RowsDense a; RowsOut2 o[2]; int n2; };
template <int KB, int EPI, int WAVES, int KB2>
__global__ __launch_bounds__(WAVES * 64) void k_rows_dense2(RowsDense2 q) {
    const RowsDense& a = q.a;
    __shared__ __attribute__((aligned(16))) float xin[KB * 256];
    __shared__ __attribute__((aligned(16))) float y1[KB2 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r = (long long)blockIdx.x * 16 + (lane & 15);
    const bool valid = r < a.rows;
    const size_t row = (size_t)(valid ? r : a.rows - 1);
    for (int b = wave; b < KB; b += WAVES) {
        const f4 v = b < a.KB1 ? ld_blk(a.X, row, a.ldx, b, lane) : ld_blk(a.X2, row, a.ldx2, b - a.KB1, lane);
        lds_st(xin, b, lane, v);
    }
    __syncthreads();
    for (int t = wave; t < KB2; t += WAVES) {                     // a.MT == KB2
        f4 acc = a.bias != nullptr ? ld_vec(a.bias, t, lane) : f4zero();
        acc = dense_tile_lds<KB>(a.W, t, xin, lane, acc);
        if (EPI == EPI_SILU) {
            if (a.Zo != nullptr && valid) st_blk(a.Zo, row, a.ldzo, t, lane, acc);
            acc = silu4(acc);
        } else if (EPI == EPI_MUL_DSILU) {
            acc = acc * dsilu4(ld_blk(a.Z, row, a.ldz, t, lane));
        } else if (EPI == EPI_ADD) {
            acc = ld_blk(a.Z, row, a.ldz, t, lane) + acc * a.scale;
        }
        lds_st(y1, t, lane, acc);
        if (valid) st_blk(a.Y, row, a.ldy, t, lane, acc);
    }
    __syncthreads();
    for (int j = 0; j < q.n2; ++j) {
        const RowsOut2& o = q.o[j];
        for (int t = wave; t < o.MT; t += WAVES) {
            f4 acc = dense_tile_lds<KB2>(o.W, t, y1, lane, f4zero());
            if (!valid) continue;
            if (o.Zadd != nullptr) acc = ld_blk(o.Zadd, row, o.ldz, t, lane) + acc;
            st_blk(o.Y, row, o.ldy, t, lane, acc);
        }
    }
}

// The same layer on LONG inputs (the inner-edge rows: rbf_proj, radial_lin): RG 16-row groups per workgroup, the weight chunks of an
// output tile loaded ONCE and used for all RG groups.  k_rows_dense re-reads the whole weight matrix from L2 for every 16 rows: at
// A = 97 152 rows x 39 output tiles x 6 chunks that is 1.4 GB of L2 traffic per launch (146 us, L2-bound); here a quarter of it.
template <int KB, int EPI, int WAVES, int RG>
__global__ __launch_bounds__(WAVES * 64) void k_rows_dense_long(RowsDense a) {
    static_assert(KB <= 13, "weight chunks of one tile are held in registers");
    __shared__ __attribute__((aligned(16))) float xin[RG * KB * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * (16 * RG);
    for (int i = wave; i < RG * KB; i += WAVES) {
        const int rg = i / KB, b = i % KB;
        const long long r = r0 + 16 * rg + (lane & 15);
        const size_t row = (size_t)(r < a.rows ? r : a.rows - 1);
        const f4 v = b < a.KB1 ? ld_blk(a.X, row, a.ldx, b, lane) : ld_blk(a.X2, row, a.ldx2, b - a.KB1, lane);
        lds_st(xin, i, lane, v);
    }
    __syncthreads();
    for (int t = wave; t < a.MT; t += WAVES) {
        const float* base = a.W + ((size_t)t * KB * 64 + lane) * 4;
        f4 w[KB];
#pragma unroll
        for (int b = 0; b < KB; ++b) w[b] = ld_f4(base + (size_t)b * 256);
        const f4 bias = a.bias != nullptr ? ld_vec(a.bias, t, lane) : f4zero();
#pragma unroll 1                                      // unrolled, hipcc hoists the LDS reads of all groups (208 registers at KB = 13: spills)
        for (int rg = 0; rg < RG; ++rg) {
            f4 c0 = bias, c1 = f4zero();
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const f4 x = lds_blk(xin, rg * KB + b, lane);
                if (b & 1) c1 = mma_chunk(w[b], x, c1);
                else c0 = mma_chunk(w[b], x, c0);
            }
            f4 acc = c0 + c1;
            const long long r = r0 + 16 * rg + (lane & 15);
            if (r >= a.rows) continue;
            const size_t row = (size_t)r;
            if (EPI == EPI_SILU) {
                if (a.Zo != nullptr) st_blk(a.Zo, row, a.ldzo, t, lane, acc);
                acc = silu4(acc);
            } else if (EPI == EPI_MUL_DSILU) {
                acc = acc * dsilu4(ld_blk(a.Z, row, a.ldz, t, lane));
            } else if (EPI == EPI_ADD) {
                acc = ld_blk(a.Z, row, a.ldz, t, lane) + acc * a.scale;
            }
            st_blk(a.Y, row, a.ldy, t, lane, acc);
        }
    }
}

// ---- LayerNorm on rows (torch.nn.LayerNorm: biased variance, eps = 1e-5 inside the sqrt) ---------------------------------------
// one wave per row; HP <= 256 (4 features per lane).  gamma / beta may be nullptr (no affine part).
// Y = LN(X + (Xadd ? Xadd : 0)); Sum (optional) receives X + Xadd (the tensor the statistics are taken of)
OARD_KERNEL __global__ __launch_bounds__(256) void k_rows_ln_fwd(const float* __restrict__ X, const float* __restrict__ Xadd, int ld, int H, int HP,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ Y, float* __restrict__ Sum, long long rows) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63, c0 = 4 * lane;
    f4 x = f4zero();
    if (c0 < HP) {
        x = ld_f4(X + (size_t)r * ld + c0);
        if (Xadd != nullptr) x += ld_f4(Xadd + (size_t)r * ld + c0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c0 + i < H ? x[i] : 0.f;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float mean = s * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float dx = x[i] - mean; q += c0 + i < H ? dx * dx : 0.f; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) q += __shfl_xor(q, d, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / H) + 1e-5f);
    if (c0 < HP) {
        f4 y;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = (x[i] - mean) * rstd;
            if (gamma != nullptr) v = v * gamma[c0 + i] + beta[c0 + i];
            y[i] = c0 + i < H ? v : 0.f;
        }
        *reinterpret_cast<f4*>(Y + (size_t)r * ld + c0) = y;
        if (Sum != nullptr) *reinterpret_cast<f4*>(Sum + (size_t)r * ld + c0) = x;
    }
}
// adjoint: given X (the normalised tensor's input) and dY:   dxhat = dY * gamma;
//   dX = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat));   G = dY * xhat (its column sums are d gamma; those of dY are d beta)
// dX = (Dadd ? Dadd : 0) + that.
OARD_KERNEL __global__ __launch_bounds__(256) void k_rows_ln_bwd(const float* __restrict__ X, int ld, int H, int HP, const float* __restrict__ gamma,
                                                     const float* __restrict__ dY, const float* __restrict__ Dadd,
                                                     float* __restrict__ dX, float* __restrict__ G, long long rows) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63, c0 = 4 * lane;
    f4 x = f4zero(), dy = f4zero();
    if (c0 < HP) { x = ld_f4(X + (size_t)r * ld + c0); dy = ld_f4(dY + (size_t)r * ld + c0); }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c0 + i < H ? x[i] : 0.f;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float mean = s * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float dx = x[i] - mean; q += c0 + i < H ? dx * dx : 0.f; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) q += __shfl_xor(q, d, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / H) + 1e-5f);
    f4 xhat, dxh;
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool ok = c0 + i < H;
        xhat[i] = ok ? (x[i] - mean) * rstd : 0.f;
        dxh[i] = ok ? dy[i] * (gamma != nullptr ? gamma[c0 + i] : 1.0f) : 0.f;
        m1 += dxh[i];
        m2 += dxh[i] * xhat[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { m1 += __shfl_xor(m1, d, 64); m2 += __shfl_xor(m2, d, 64); }
    m1 *= 1.0f / H; m2 *= 1.0f / H;
    if (c0 < HP) {
        f4 o, g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = c0 + i < H;
            o[i] = ok ? rstd * (dxh[i] - m1 - xhat[i] * m2) : 0.f;
            g[i] = ok ? dy[i] * xhat[i] : 0.f;
        }
        if (Dadd != nullptr) o += ld_f4(Dadd + (size_t)r * ld + c0);
        *reinterpret_cast<f4*>(dX + (size_t)r * ld + c0) = o;
        if (G != nullptr) *reinterpret_cast<f4*>(G + (size_t)r * ld + c0) = g;
    }
}

// ---- fixed-order column sums over rows [r0, r1) of a row-major buffer: out[c] (+)= sum_r w(r) X[r][c] ------------------------------
// pass 1: row chunk q (CS_ROWS rows) -> part[q][c] (ascending rows, one thread per column);  pass 2: one wave per column adds the
// chunks in a fixed order (chunk_sum_wave).  Deterministic, no atomics.  Optional per-row weight (a [rows] vector, e.g. the gate
// adjoint) and SiLU applied to X on load (the att_mlp gradient: sum_e da_e SiLU(z2_e)).
#define CS_ROWS 64
OARD_KERNEL __global__ __launch_bounds__(256) void k_colsum_part(const float* __restrict__ X, int ld, long long r0, long long r1, int ncols,
                                                     const float* __restrict__ wrow, int x_silu, float* __restrict__ part) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= ncols) return;
    const long long rb = r0 + (long long)blockIdx.x * CS_ROWS, re = rb + CS_ROWS < r1 ? rb + CS_ROWS : r1;
    float s = 0.f;
    for (long long r = rb; r < re; ++r) {
        float v = X[(size_t)r * ld + c];
        if (x_silu) v = silu1(v);
        s += wrow != nullptr ? v * wrow[r] : v;
    }
    part[(size_t)blockIdx.x * ncols + c] = s;
}
// long matrices (thousands of rows, ncols a multiple of 4, 16-byte aligned rows): float4 loads, the block's 256 threads as RL row
// lanes x nc4 column quads (RL = 256 / nc4), a thread adds rows rb + rl, rb + rl + RL, ... of its quad in ascending order, the RL lane
// sums are then added in lane order - a fixed order.  CSL_ROWS rows per block -> part[q][c] as k_colsum_part.  (The one-thread-per-
// column kernel above spent 1.66 ms per training step, most of it on the six [E][196] sums of the att_mlp weight gradient.)
// rows per block: 512 on the long edge-level inputs; node-level inputs (a few thousand rows) get 32, i.e. > 100 blocks instead of 9 (the
// sums are a function of (rows, rows per block) only - still a fixed order)
#define CSL_ROWS 512
#define CSL_ROWS_SHORT 32
OARD_KERNEL __global__ __launch_bounds__(256) void k_colsum_long(const float* __restrict__ X, int ld, long long r0, long long r1, int ncols,
                                                     const float* __restrict__ wrow, int x_silu, float* __restrict__ part, int rows_per_block) {
    // float64 accumulators (round 5): the att_mlp gradients are sums of a signed value over every edge whose terms largely cancel (condition
    // ~ sqrt(E)); with float32 running sums the bias gradient - one scalar - carried ~1e-5 of relative error.  The kernel is bound by its loads.
    typedef double d4 __attribute__((ext_vector_type(4)));
    __shared__ d4 red[256];
    const int nc4 = ncols >> 2, RL = 256 / nc4, rl = threadIdx.x / nc4, c4 = threadIdx.x - rl * nc4;
    const long long rb = r0 + (long long)blockIdx.x * rows_per_block, re = rb + rows_per_block < r1 ? rb + rows_per_block : r1;
    d4 s = {0.0, 0.0, 0.0, 0.0};
    if (rl < RL) {
        const float* p = X + 4 * c4;
#pragma unroll 4
        for (long long r = rb + rl; r < re; r += RL) {
            f4 v = ld_f4(p + (size_t)r * ld);
            if (x_silu) v = silu4(v);
            if (wrow != nullptr) v = v * wrow[r];
            s += (d4){(double)v.x, (double)v.y, (double)v.z, (double)v.w};
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < RL; ++k) s += red[k * nc4 + c4];
        st_f4(part + (size_t)blockIdx.x * ncols + 4 * c4, (f4){(float)s.x, (float)s.y, (float)s.z, (float)s.w});
    }
}
// narrow matrices (ncols <= 16, e.g. the [items][12] operand of the frame-scalar MLP's last layer): one thread per ROW slice instead
// of one per column - thread t of chunk q adds rows q CSN_ROWS + t, + 256, ... for all columns in registers, then the 256 thread
// sums are combined by a fixed tree in LDS.  part[q][c] as above.
#define CSN_ROWS 8192
OARD_KERNEL __global__ __launch_bounds__(256) void k_colsum_narrow(const float* __restrict__ X, int ld, long long r0, long long r1, int ncols,
                                                       const float* __restrict__ wrow, float* __restrict__ part) {
    __shared__ float red[256][17];
    const long long rb = r0 + (long long)blockIdx.x * CSN_ROWS, re = rb + CSN_ROWS < r1 ? rb + CSN_ROWS : r1;
    float s[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) s[c] = 0.f;
    for (long long r = rb + threadIdx.x; r < re; r += 256) {
        const float w = wrow != nullptr ? wrow[r] : 1.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < ncols) s[c] += X[(size_t)r * ld + c] * w;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) red[threadIdx.x][c] = s[c];
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d)
#pragma unroll
            for (int c = 0; c < 16; ++c) red[threadIdx.x][c] += red[threadIdx.x + d][c];
        __syncthreads();
    }
    if ((int)threadIdx.x < ncols) part[(size_t)blockIdx.x * ncols + threadIdx.x] = red[0][threadIdx.x];
}
OARD_KERNEL __global__ __launch_bounds__(256) void k_colsum_fin(const float* __restrict__ part, int n_chunks, int ncols, float* __restrict__ out,
                                                    int accumulate, float scale) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= ncols) return;
    const float s = chunk_sum_wave(part + c, (size_t)ncols, n_chunks, threadIdx.x & 63) * scale;
    if ((threadIdx.x & 63) == 0) out[c] = accumulate ? out[c] + s : s;
}

// out[i][j] (+)= a[i] * b[j]   (the constant-row term of layer 0's edge_mlp.0 gradient: outer(sum_e dz1_e, c0row))
OARD_KERNEL __global__ void k_outer_acc(float* __restrict__ out, int ld, const float* __restrict__ a, int na, const float* __restrict__ b, int nb_,
                            int accumulate) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)na * nb_) return;
    const int i = (int)(idx / nb_), j = (int)(idx % nb_);
    const float v = a[i] * b[j];
    float* o = out + (size_t)i * ld + j;
    *o = accumulate ? *o + v : v;
}

// out[i * stride] += a[i]   (a gradient that lands in one column of a wider parameter)
OARD_KERNEL __global__ void k_strided_acc(const float* __restrict__ a, int n, float* __restrict__ out, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[(size_t)i * stride] += a[i];
}
OARD_KERNEL __global__ void k_silu_rows(const float* __restrict__ z, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = silu1(z[i]);
}
// out = a * SiLU'(z)
OARD_KERNEL __global__ void k_mul_dsilu_rows(const float* __restrict__ a, const float* __restrict__ z, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float zz = z[i], s = __builtin_amdgcn_rcpf(1.0f + __expf(-zz));
    out[i] = a[i] * (s * (1.0f + zz * (1.0f - s)));
}
