// oard_engine.h — the "column engine": device-side building blocks shared by every kernel.
//
// Data model.  A wavefront (64 lanes) owns 16 *columns* (edges or nodes).  A feature vector of
// a column lives in registers as "blocks": block b = features 16b..16b+15, held as one float4
// per lane where lane = 16*g + e (g = lane>>4 in 0..3, e = lane&15 = column) holds features
// 16b + 4g + {0,1,2,3} of column e.  This is simultaneously
//   * what a float4 load from a row-major [column][feature] array delivers,
//   * the B operand of four consecutive v_mfma_f32_16x16x4_f32 steps (step r uses component r:
//     B[k = g][j = e] = feature 16b + 4g + r), and
//   * the C/D layout of that MFMA (row = 4g + r, col = e) when the *output features* are the
//     MFMA rows,
// so a chain of Linear layers runs register-to-register with features as MFMA rows, columns as
// MFMA columns, and the weights as the A operand.  Weights are pre-packed (oard_pack_weights)
// into 1-KiB chunks, chunk (t, b) = W[16t..16t+15][16b..16b+15] stored lane-linear: lane
// (g, o) holds W[16t + o][16b + 4g + 0..3], i.e. A[i = o][k = g] for step r = component r.
//
// fp32-input MFMA is an exact fmaf chain (MI355X guide, section 3), so the only deviation from
// a scalar fp32 evaluation is summation order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

#define OARD_DEV __device__ __forceinline__
// The library is several translation units (oareactdiff_amd/build.py): oard_hip.hip holds the host code and launches every kernel; the
// heavy kernel families are explicitly instantiated in oard_inst_*.hip (declared `extern template` in oard_hip.hip, oard_inst.h), so that
// an edit to one family recompiles one unit.  A kernel that is NOT a template, and every __device__ variable, must exist in one unit only:
// in an instantiation unit such a kernel becomes a template that nothing instantiates (a `static` kernel would still be emitted), a
// __device__ variable a static of that unit.  -DOARD_SINGLE_TU (probe / timeline builds that read __device__
// variables back): everything in oard_hip.hip, as before round 6.
#ifdef OARD_INST_TU
#define OARD_KERNEL template <int OARD_NEVER_INSTANTIATED_>
#define OARD_DEVVAR static __attribute__((unused))
#else
#define OARD_KERNEL
#define OARD_DEVVAR
#endif

OARD_DEV f4 f4zero() { return (f4){0.f, 0.f, 0.f, 0.f}; }

// acc(16 out-features x 16 columns) += A-chunk x B-block : four k=4 MFMA steps
OARD_DEV f4 mma_chunk(f4 a, f4 b, f4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

// packed A chunk (t, b) of a matrix with KB input blocks
OARD_DEV f4 ld_chunk(const float* __restrict__ wp, int t, int b, int KB, int lane) {
    return *reinterpret_cast<const f4*>(wp + ((size_t)(t * KB + b) * 64 + lane) * 4);
}

// per-feature vector (bias / gamma / ...) for output tile t, in C layout
OARD_DEV f4 ld_vec(const float* __restrict__ v, int t, int lane) {
    return *reinterpret_cast<const f4*>(v + 16 * t + 4 * (lane >> 4));
}

// block b of a row-major [row][ld] array for this lane's column row
OARD_DEV f4 ld_blk(const float* __restrict__ base, size_t row, int ld, int b, int lane) {
    return *reinterpret_cast<const f4*>(base + row * (size_t)ld + 16 * b + 4 * (lane >> 4));
}
OARD_DEV void st_blk(float* __restrict__ base, size_t row, int ld, int b, int lane, f4 v) {
    *reinterpret_cast<f4*>(base + row * (size_t)ld + 16 * b + 4 * (lane >> 4)) = v;
}

#ifdef OARD_ABL_NOEPI        // timing-only ablation (experiment build)
OARD_DEV float silu1(float x) { return x; }
OARD_DEV float silu1_real(float x) {
#else
OARD_DEV float silu1(float x) {
#endif
    // x * sigmoid(x);  v_exp_f32 + v_rcp_f32 (both <= 1 ulp)
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
// four values: the three non-transcendental steps as packed two-float operations (v_pk_mul_f32 / v_pk_add_f32) - every VALU
// instruction of an MFMA kernel costs ~6 cycles of the SIMD's MFMA issue time (tools/micro/mfma_valu.hip), transcendental ones ~10
typedef float f2 __attribute__((ext_vector_type(2)));
#ifdef OARD_ABL_NOEPI
OARD_DEV f4 silu4(f4 v) { return v; }
#else
OARD_DEV f4 silu4(f4 v) {
    const f2 a = {v.x, v.y}, b = {v.z, v.w};
    const f2 ta = a * -1.44269504088896340736f, tb = b * -1.44269504088896340736f;      // exp(-x) = 2^(-x log2 e)
    const f2 da = (f2){__builtin_amdgcn_exp2f(ta.x), __builtin_amdgcn_exp2f(ta.y)} + 1.0f;
    const f2 db = (f2){__builtin_amdgcn_exp2f(tb.x), __builtin_amdgcn_exp2f(tb.y)} + 1.0f;
    const f2 ya = a * (f2){__builtin_amdgcn_rcpf(da.x), __builtin_amdgcn_rcpf(da.y)};
    const f2 yb = b * (f2){__builtin_amdgcn_rcpf(db.x), __builtin_amdgcn_rcpf(db.y)};
    return (f4){ya.x, ya.y, yb.x, yb.y};
}
#endif

// one output tile of a dense layer: sum_b chunk(t,b) x in[b]   (t may be a runtime value)
template <int KB>
OARD_DEV f4 dense_tile(const float* __restrict__ wp, int t, const f4 (&in)[KB], int lane, f4 acc) {
    const float* base = wp + ((size_t)t * KB * 64 + lane) * 4;
#pragma unroll
    for (int b = 0; b < KB; ++b) {
        f4 a = *reinterpret_cast<const f4*>(base + (size_t)b * 256);
        acc = mma_chunk(a, in[b], acc);
    }
    return acc;
}

// two output tiles at once (independent accumulators hide the 40-cycle MFMA dependency)
template <int KB>
OARD_DEV void dense_tile2(const float* __restrict__ wp, int t0, int t1, const f4 (&in)[KB], int lane,
                          f4& acc0, f4& acc1) {
    const float* b0 = wp + ((size_t)t0 * KB * 64 + lane) * 4;
    const float* b1 = wp + ((size_t)t1 * KB * 64 + lane) * 4;
#pragma unroll
    for (int b = 0; b < KB; ++b) {
        f4 a0 = *reinterpret_cast<const f4*>(b0 + (size_t)b * 256);
        f4 a1 = *reinterpret_cast<const f4*>(b1 + (size_t)b * 256);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, in[b].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, in[b].x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, in[b].y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, in[b].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, in[b].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, in[b].z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, in[b].w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, in[b].w, acc1, 0, 0, 0);
    }
}

// full dense layer into registers: out[t] = act(sum_b chunk(t,b) x in[b] + bias[t])
template <int KB, int MT, bool ACT, bool BIAS>
OARD_DEV void dense_regs(const float* __restrict__ wp, const float* __restrict__ bias,
                         const f4 (&in)[KB], f4 (&out)[MT], int lane) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        f4 acc = BIAS ? ld_vec(bias, t, lane) : f4zero();
        acc = dense_tile<KB>(wp, t, in, lane, acc);
        out[t] = ACT ? silu4(acc) : acc;
    }
}

// sum over the features of each column: per-lane partial -> across the 4 lane groups
OARD_DEV float col_reduce(float partial) {
    partial += __shfl_xor(partial, 16, 64);
    partial += __shfl_xor(partial, 32, 64);
    return partial;
}

// LayerNorm over the H valid features of each column (biased variance, eps inside the sqrt,
// torch.nn.LayerNorm semantics).  Padded features (>= H) are forced to 0 on output.
template <int HT, int H, bool AFFINE>
OARD_DEV void layer_norm(f4 (&v)[HT], const float* __restrict__ gamma, const float* __restrict__ beta,
                         int lane) {
    const int g = lane >> 4;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const int f0 = 16 * t + 4 * g;
        s += (f0 + 0 < H ? v[t].x : 0.f) + (f0 + 1 < H ? v[t].y : 0.f) +
             (f0 + 2 < H ? v[t].z : 0.f) + (f0 + 3 < H ? v[t].w : 0.f);
    }
    const float mean = col_reduce(s) * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const int f0 = 16 * t + 4 * g;
        float dx = v[t].x - mean, dy = v[t].y - mean, dz = v[t].z - mean, dw = v[t].w - mean;
        q += (f0 + 0 < H ? dx * dx : 0.f) + (f0 + 1 < H ? dy * dy : 0.f) +
             (f0 + 2 < H ? dz * dz : 0.f) + (f0 + 3 < H ? dw * dw : 0.f);
    }
    const float var = col_reduce(q) * (1.0f / H);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const int f0 = 16 * t + 4 * g;
        f4 y = (v[t] - mean) * rstd;
        if (AFFINE) y = y * ld_vec(gamma, t, lane) + ld_vec(beta, t, lane);
        y.x = f0 + 0 < H ? y.x : 0.f;
        y.y = f0 + 1 < H ? y.y : 0.f;
        y.z = f0 + 2 < H ? y.z : 0.f;
        y.w = f0 + 3 < H ? y.w : 0.f;
        v[t] = y;
    }
}
