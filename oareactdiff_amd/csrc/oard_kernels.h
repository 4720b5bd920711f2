// oard_kernels.h — all device kernels of the denoising call.  See DESIGN.md for the stage map.
// Reference line numbers are into oa_reactdiff/model/leftnet.py unless a file is named.
#pragma once
#include "oard_engine.h"
#include "oard_layout.h"

#define GEO_STRIDE 12   // per inner edge: d, env, u[3], c[3], v[3], mask
#define EPSF 1e-6

// column bookkeeping shared by the engine kernels: 4 independent waves per block
struct ColId {
    int lane, g, col;
    bool valid;
    long long c;   // clamped column index (safe to load from)
};
OARD_DEV ColId col_id(long long ncols, bool& wave_live) {
    ColId id;
    id.lane = threadIdx.x & 63;
    id.g = id.lane >> 4;
    const long long cb = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    wave_live = cb * 16 < ncols;
    const long long c = cb * 16 + (id.lane & 15);
    id.valid = c < ncols;
    id.c = id.valid ? c : (ncols > 0 ? ncols - 1 : 0);
    id.col = (int)id.c;
    return id;
}

OARD_DEV int wave_max(int v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

OARD_DEV f4 ld_f4(const float* p) { return *reinterpret_cast<const f4*>(p); }
OARD_DEV void st_f4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }

// =====================================================================================================
// weight packing
// =====================================================================================================
struct PackJob {
    const float* src;
    int src_ld, col_off;          // source row stride (in features) and first source column
    int msect_len, msect_pad, msects;   // dest row r -> section r / msect_pad, within r % msect_pad (< msect_len valid)
    int ksect_len, ksect_pad, ksects;
    int MT, KB;
    size_t dst;
    // destination of chunk (t, b): dst + slot(t) * tstride + b * bstride  (floats); slot(t) = t, or with
    // perm_ht > 0 (thirds interleave): slot(t) = (t % perm_ht) * 3 + t / perm_ht
    size_t tstride, bstride;
    int perm_ht;
    int transpose;                // 1: the chunk rows index the SOURCE columns and the chunk K index the source rows
                                  //    (packs W^T for the backward pass: dX = W^T dY); col_off applies to the row index
    int rows4;                    // P > 0: every tile t with t % P == P-1 has <= 4 real rows and is packed for the 4x4x1 MFMA: lane (g, i)
                                  //    holds row 16t + i%4 (oard_edge_v1.h, mma4_chunk); its last K block is never compacted
    int tail_compact;             // 1: the last K block (b == KB-1, <= 4 real features) is packed as ONE k-step: component 0 of
                                  //    lane (g, i) = W[row 16t+i][col 16b+g], components 1..3 = 0 (oard_edge_v1.h, compact K tail)
};

// matrix -> MFMA chunks: dst[((t*KB + b)*64 + lane)*4 + c] = W[row(16t + (lane&15))][col(16b + 4(lane>>4) + c)]
OARD_KERNEL __global__ void k_pack_matrix(PackJob j, float* __restrict__ blob) {
    const size_t total = (size_t)j.MT * j.KB * 256;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const size_t ch = i >> 8;
        const int b = (int)(ch % j.KB), t = (int)(ch / j.KB);
        const bool r4 = j.rows4 > 0 && t % j.rows4 == j.rows4 - 1;
        const bool tail = j.tail_compact && b == j.KB - 1 && !r4;
        const int r = 16 * t + (r4 ? (lane & 3) : (lane & 15)), k = tail ? 16 * b + (lane >> 4) : 16 * b + 4 * (lane >> 4) + c;
        const int rs = r / j.msect_pad, rw = r % j.msect_pad;
        const int ks = k / j.ksect_pad, kw = k % j.ksect_pad;
        float v = 0.f;
        if (rs < j.msects && rw < j.msect_len && ks < j.ksects && kw < j.ksect_len && !(tail && c != 0))
            v = j.transpose ? j.src[(size_t)(ks * j.ksect_len + kw) * j.src_ld + j.col_off + rs * j.msect_len + rw]
                            : j.src[(size_t)(rs * j.msect_len + rw) * j.src_ld + j.col_off + ks * j.ksect_len + kw];
        const int slot = j.perm_ht > 0 ? (t % j.perm_ht) * 3 + t / j.perm_ht : t;
        blob[j.dst + (size_t)slot * j.tstride + (size_t)b * j.bstride + (i & 255)] = v;
    }
}
// per-feature vector as "bias chunks" for the LDS weight stream: chunk of tile t holds, for lane (g, o),
// the float4 vec[16t + 4g .. +3] (i.e. ld_vec's layout, replicated over o); same sectioning as k_pack_vector
OARD_KERNEL __global__ void k_pack_bias_chunks(const float* __restrict__ src, float* __restrict__ dst, int sect_len, int sect_pad,
                                   int sects, int n_tiles, size_t tstride, int perm_ht) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles * 256) return;
    const int t = i >> 8, lane = (i >> 2) & 63, c = i & 3;
    const int r = 16 * t + 4 * (lane >> 4) + c;
    const int s = r / sect_pad, w = r % sect_pad;
    const int slot = perm_ht > 0 ? (t % perm_ht) * 3 + t / perm_ht : t;
    dst[(size_t)slot * tstride + (i & 255)] = (src != nullptr && s < sects && w < sect_len) ? src[s * sect_len + w] : 0.f;
}
// vector with the same row sectioning, padded with zeros; n_dst = msects * msect_pad (or MT*16)
OARD_KERNEL __global__ void k_pack_vector(const float* __restrict__ src, float* __restrict__ dst, int sect_len, int sect_pad,
                              int sects, int n_dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_dst) return;
    const int s = i / sect_pad, w = i % sect_pad;
    dst[i] = (src != nullptr && s < sects && w < sect_len) ? src[s * sect_len + w] : 0.f;
}
OARD_KERNEL __global__ void k_copy_raw(const float* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
// ---- all packing jobs of one oard_pack_weights(_bwd) call in ONE launch --------------------------------------------------------------
// The weights change every training step, so both blobs are re-packed every step: ~420 launches of a few microseconds each.  The jobs
// (matrix -> MFMA chunks, padded vectors, bias chunks, raw copies) only depend on the parameter / blob addresses, so their table is
// built once on the host, cached on the device, and one kernel walks it: block b belongs to the job whose [block0, block0 + nblocks)
// range holds it (binary search), 256 elements per block.
struct GenJob {
    int type;                  // 0 matrix, 1 vector, 2 bias chunks, 3 raw copy
    PackJob m;                 // type 0 (dst inside)
    const float* src;          // types 1..3
    size_t dst;
    int sect_len, sect_pad, sects, n;      // n: destination elements (1: n_dst, 2: n_tiles * 256, 3: n)
    size_t tstride;
    int perm_ht;
    long long block0;          // first block of this job
};
OARD_KERNEL __global__ __launch_bounds__(256) void k_pack_all(const GenJob* __restrict__ jobs, int n_jobs, float* __restrict__ blob) {
    int lo = 0, hi = n_jobs - 1;
    const long long b = blockIdx.x;
    while (lo < hi) {                                     // last job with block0 <= b
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= b) lo = mid; else hi = mid - 1;
    }
    const GenJob& J = jobs[lo];
    const size_t i = (size_t)(b - J.block0) * 256 + threadIdx.x;
    if (J.type == 0) {
        const PackJob& j = J.m;
        const size_t total = (size_t)j.MT * j.KB * 256;
        if (i >= total) return;
        const int c = (int)(i & 3), lane = (int)((i >> 2) & 63);
        const size_t ch = i >> 8;
        const int bb = (int)(ch % j.KB), t = (int)(ch / j.KB);
        const bool r4 = j.rows4 > 0 && t % j.rows4 == j.rows4 - 1;
        const bool tail = j.tail_compact && bb == j.KB - 1 && !r4;
        const int r = 16 * t + (r4 ? (lane & 3) : (lane & 15)), k = tail ? 16 * bb + (lane >> 4) : 16 * bb + 4 * (lane >> 4) + c;
        const int rs = r / j.msect_pad, rw = r % j.msect_pad;
        const int ks = k / j.ksect_pad, kw = k % j.ksect_pad;
        float v = 0.f;
        if (rs < j.msects && rw < j.msect_len && ks < j.ksects && kw < j.ksect_len && !(tail && c != 0))
            v = j.transpose ? j.src[(size_t)(ks * j.ksect_len + kw) * j.src_ld + j.col_off + rs * j.msect_len + rw]
                            : j.src[(size_t)(rs * j.msect_len + rw) * j.src_ld + j.col_off + ks * j.ksect_len + kw];
        const int slot = j.perm_ht > 0 ? (t % j.perm_ht) * 3 + t / j.perm_ht : t;
        blob[j.dst + (size_t)slot * j.tstride + (size_t)bb * j.bstride + (i & 255)] = v;
    } else if (J.type == 1) {
        if (i >= (size_t)J.n) return;
        const int s = (int)(i / J.sect_pad), w = (int)(i % J.sect_pad);
        blob[J.dst + i] = (J.src != nullptr && s < J.sects && w < J.sect_len) ? J.src[s * J.sect_len + w] : 0.f;
    } else if (J.type == 2) {
        if (i >= (size_t)J.n) return;
        const int t = (int)(i >> 8), lane = (int)((i >> 2) & 63), c = (int)(i & 3);
        const int r = 16 * t + 4 * (lane >> 4) + c;
        const int s = r / J.sect_pad, w = r % J.sect_pad;
        const int slot = J.perm_ht > 0 ? (t % J.perm_ht) * 3 + t / J.perm_ht : t;
        blob[J.dst + (size_t)slot * J.tstride + (i & 255)] = (J.src != nullptr && s < J.sects && w < J.sect_len) ? J.src[s * J.sect_len + w] : 0.f;
    } else {
        if (i < (size_t)J.n) blob[J.dst + i] = J.src[i];
    }
}

// constant state of a masked edge (leftnet.py:768-809 with dist = 0, radial_emb = 0, frame = 0):
//   [ lin3(0) x 2H | radial_lin(0) | 0 x R | pad ]
OARD_KERNEL __global__ void k_c0row(const float* __restrict__ lin3w0b /*b0*/, const float* __restrict__ lin3w2,
                        const float* __restrict__ lin3b2, const float* __restrict__ rl0b,
                        const float* __restrict__ rl2w, const float* __restrict__ rl2b, float* __restrict__ c0,
                        int H, int H4, int WP) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= WP) return;
    float v = 0.f;
    if (i < 2 * H) {
        float a = lin3b2[0];
        for (int k = 0; k < H4; ++k) { const float x = lin3w0b[k]; a += lin3w2[k] * (x / (1.0f + expf(-x))); }
        v = a;
    } else if (i < 3 * H) {
        const int f = i - 2 * H;
        float a = rl2b[f];
        for (int k = 0; k < H; ++k) { const float x = rl0b[k]; a += rl2w[(size_t)f * H + k] * (x / (1.0f + expf(-x))); }
        v = a;
    }
    c0[i] = v;
}

// u0 = W1c(layer 0) . c0row (+ nothing else): what stage S1 of the first GCL layer yields on every
// inter-object edge, whose initial state is the constant row
OARD_KERNEL __global__ void k_u0(const float* __restrict__ w_edge_mlp0 /*[H][2H+W]*/, const float* __restrict__ c0,
                     float* __restrict__ u0, int H, int W, int HP) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= HP) return;
    float a = 0.f;
    if (f < H) {
        const float* w = w_edge_mlp0 + (size_t)f * (2 * H + W) + 2 * H;
        for (int k = 0; k < W; ++k) a += w[k] * c0[k];
    }
    u0[f] = a;
}

// =====================================================================================================
// topology check: is edge_index exactly get_edges_index(combined_mask, remove_self_edge=True)?
// =====================================================================================================
// edge_index ([2, n_edges] int64, reference node ids) is a permutation of the complete per-sample edge set: see
// oard_topology_check_edge_index.  One thread per given edge; a repeated id finds its bit already set.
OARD_KERNEL __global__ void k_check_edge_set(const int* __restrict__ ref_sample, const int* __restrict__ ref_rank, const long long* __restrict__ ref_ptr,
                                 int n_nodes, const long long* __restrict__ ei, long long n_edges, unsigned* bitmap, int* ok) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_edges) return;
    const long long i = ei[p], j = ei[n_edges + p];
    if (i < 0 || j < 0 || i >= n_nodes || j >= n_nodes || i == j || ref_sample[i] != ref_sample[j]) { atomicAnd(ok, 0); return; }
    const int ri = ref_rank[i], rj = ref_rank[j];
    const long long id = ref_ptr[i] + rj - (rj > ri ? 1 : 0);
    const unsigned bit = 1u << (id & 31);
    if (atomicOr(bitmap + (id >> 5), bit) & bit) atomicAnd(ok, 0);
}

// =====================================================================================================
// wrapper prologue / epilogue (egnn_dynamics.py:91-119, 145-160; _base.py:88-109)
// =====================================================================================================
struct ObjPtrs {
    const float* xh[OARD_MAX_OBJECTS];
    float* out[OARD_MAX_OBJECTS];
    int node_nf[OARD_MAX_OBJECTS];
    size_t enc[OARD_MAX_OBJECTS], dec[OARD_MAX_OBJECTS];
};

OARD_DEV float silu_acc(float x) { return x / (1.0f + expf(-x)); }

OARD_KERNEL __global__ void k_prep(TopoDev tp, ObjPtrs op, const float* __restrict__ wb, float* __restrict__ pos,
                       float* __restrict__ hin, const float* __restrict__ t, int t_scalar,
                       const float* __restrict__ cond, int cnf, int ctime, int emb) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= tp.N) return;
    const int obj = tp.node_obj[n], row = tp.node_row[n], nf = op.node_nf[obj], d = nf - 3;
    const float* x = op.xh[obj] + (size_t)row * nf;
    pos[n * 3 + 0] = x[0]; pos[n * 3 + 1] = x[1]; pos[n * 3 + 2] = x[2];
    const float* W0 = wb + op.enc[obj];
    const float* b0 = W0 + 2 * d * d;
    const float* W1 = b0 + 2 * d;
    const float* b1 = W1 + emb * 2 * d;
    float hid[32];
    for (int o = 0; o < 2 * d; ++o) {
        float a = 0.f;
        for (int i = 0; i < d; ++i) a += W0[o * d + i] * x[3 + i];
        hid[o] = silu_acc(a + b0[o]);
    }
    float* h = hin + (size_t)n * 16;
    int c = 0;
    for (int o = 0; o < emb; ++o) {
        float a = 0.f;
        for (int i = 0; i < 2 * d; ++i) a += W1[o * 2 * d + i] * hid[i];
        h[c++] = a + b1[o];
    }
    const int ti = tp.node_tidx[n];
    if (ctime) h[c++] = t_scalar ? t[0] : t[ti];
    for (int q = 0; q < cnf; ++q) h[c++] = cond[(size_t)ti * cnf + q];
    for (; c < 16; ++c) h[c] = 0.f;
}

OARD_KERNEL __global__ void k_post(TopoDev tp, ObjPtrs op, const float* __restrict__ wb, const float* __restrict__ dpos,
                       const float* __restrict__ hout, int emb) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= tp.N) return;
    const int obj = tp.node_obj[n], row = tp.node_row[n], nf = op.node_nf[obj], d = nf - 3;
    const int q = tp.node_sample[n] * tp.n_obj + obj;
    const int g0 = tp.grp_ptr[q], g1 = tp.grp_ptr[q + 1];
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    for (int k = g0; k < g1; ++k) { m0 += dpos[k * 3]; m1 += dpos[k * 3 + 1]; m2 += dpos[k * 3 + 2]; }
    const float inv = 1.0f / (float)(g1 - g0);
    float* o = op.out[obj] + (size_t)row * nf;
    o[0] = dpos[n * 3] - m0 * inv; o[1] = dpos[n * 3 + 1] - m1 * inv; o[2] = dpos[n * 3 + 2] - m2 * inv;
    const float* W0 = wb + op.dec[obj];
    const float* b0 = W0 + 2 * d * emb;
    const float* W1 = b0 + 2 * d;
    const float* b1 = W1 + d * 2 * d;
    const float* h = hout + (size_t)n * 16;
    float hid[32];
    for (int k = 0; k < 2 * d; ++k) {
        float a = 0.f;
        for (int i = 0; i < emb; ++i) a += W0[k * emb + i] * h[i];
        hid[k] = silu_acc(a + b0[k]);
    }
    for (int k = 0; k < d; ++k) {
        float a = 0.f;
        for (int i = 0; i < 2 * d; ++i) a += W1[k * 2 * d + i] * hid[i];
        o[3 + k] = a + b1[k];
    }
}

// NaN guard of the reference without its host sync (egnn_dynamics.py:138-143: `if torch.any(torch.isnan(vel)): vel = randn_like(vel)`
// before the per-object CoM removal): when the call's status flag is set, EVERY object's velocity block is replaced by CoM-free
// N(0,1) noise the caller drew beforehand; otherwise nothing is touched.  One thread per node.
struct NanPtrs { const float* noise[OARD_MAX_OBJECTS]; float* out[OARD_MAX_OBJECTS]; int node_nf[OARD_MAX_OBJECTS]; };
OARD_KERNEL __global__ void k_nan_replace(TopoDev tp, NanPtrs np, const int* __restrict__ status) {
    if (status[0] == 0) return;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= tp.N) return;
    const int obj = tp.node_obj[n], row = tp.node_row[n], nf = np.node_nf[obj];
    const int q = tp.node_sample[n] * tp.n_obj + obj;
    const int g0 = tp.grp_ptr[q], g1 = tp.grp_ptr[q + 1];
    const float* R = np.noise[obj];
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    for (int k = g0; k < g1; ++k) { const float* r = R + (size_t)tp.node_row[k] * 3; m0 += r[0]; m1 += r[1]; m2 += r[2]; }
    const float inv = 1.0f / (float)(g1 - g0);
    float* o = np.out[obj] + (size_t)row * nf;
    const float* r = R + (size_t)row * 3;
    o[0] = r[0] - m0 * inv; o[1] = r[1] - m1 * inv; o[2] = r[2] - m2 * inv;
}

// =====================================================================================================
// sampler step (en_diffusion.py:562-702, 278-305; _utils.py:9-31): one thread per node
// =====================================================================================================
struct SamplerPtrs {
    const float* z[OARD_MAX_OBJECTS];
    const float* eh[OARD_MAX_OBJECTS];
    const float* noise[OARD_MAX_OBJECTS];
    const float* h0[OARD_MAX_OBJECTS];
    float* out[OARD_MAX_OBJECTS];
    int node_nf[OARD_MAX_OBJECTS];
};
// coef != NULL: the three schedule scalars come from device memory (a captured hipGraph replays the same launch for every
// step; the host only advances a device-side step counter that selects the row of the coefficient table)
OARD_KERNEL __global__ void k_sampler_step(TopoDev tp, SamplerPtrs sp, int mode, float a, float b, float c, const float* __restrict__ coef,
                               int zero_h) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= tp.N) return;
    if (coef != nullptr) { a = coef[0]; b = coef[1]; c = coef[2]; }
    const int obj = tp.node_obj[n], row = tp.node_row[n], nf = sp.node_nf[obj];
    const int q = tp.node_sample[n] * tp.n_obj + obj;
    const int g0 = tp.grp_ptr[q], g1 = tp.grp_ptr[q + 1];
    const float inv = 1.0f / (float)(g1 - g0);
    const float* Z = sp.z[obj]; const float* E = sp.eh[obj]; const float* R = sp.noise[obj];
    // group means: raw noise, then the un-projected update (scatter_mean order = ascending row)
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    for (int k = g0; k < g1; ++k) {
        const float* r = R + (size_t)tp.node_row[k] * nf;
        m0 += r[0]; m1 += r[1]; m2 += r[2];
    }
    m0 *= inv; m1 *= inv; m2 *= inv;
    auto upd = [&](float z, float e, float eps) -> float {
        if (mode == 0) return z / a - e * b + c * eps;
        if (mode == 1) return a * (z - b * e) + c * eps;
        if (mode >= 3) return a * z + c * eps;
        return eps;
    };
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (mode == 0 || mode == 4) {
        for (int k = g0; k < g1; ++k) {
            const size_t o = (size_t)tp.node_row[k] * nf;
            s0 += upd(Z[o], E[o], R[o] - m0); s1 += upd(Z[o + 1], E[o + 1], R[o + 1] - m1); s2 += upd(Z[o + 2], E[o + 2], R[o + 2] - m2);
        }
        s0 *= inv; s1 *= inv; s2 *= inv;
    }
    const size_t o = (size_t)row * nf;
    float* out = sp.out[obj] + o;
    out[0] = upd(Z[o], E[o], R[o] - m0) - s0;
    out[1] = upd(Z[o + 1], E[o + 1], R[o + 1] - m1) - s1;
    out[2] = upd(Z[o + 2], E[o + 2], R[o + 2] - m2) - s2;
    const float* h0 = sp.h0[obj];
    for (int j = 3; j < nf; ++j)
        out[j] = h0 ? h0[o - 3 * (size_t)row + (size_t)(j - 3)] : upd(Z[o + j], E[o + j], zero_h ? 0.f : R[o + j]);
}

// =====================================================================================================
// geometry block in float64 (leftnet.py:747-761, 707-722, 812-834)
// =====================================================================================================
// one 64-thread block per (sample, object) group: cutoff graph inside the group, the reference's
// one-hop labelling with overwrites, label-mean removal; then the node frame in its exact form.
OARD_KERNEL __global__ void k_geom(TopoDev tp, const float* __restrict__ pos, double cutoff, double* __restrict__ pf64,
                       float* __restrict__ pf32, float* __restrict__ x1, float* __restrict__ pp0,
                       int* __restrict__ labels) {
    __shared__ double sp[OARD_MAX_GROUP * 3];
    __shared__ int lab[OARD_MAX_GROUP];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int g0 = tp.grp_ptr[q], ng = tp.grp_ptr[q + 1] - g0;
    if (ng == 0) return;
    for (int j = tid; j < ng; j += 64) {
        sp[j * 3] = pos[(g0 + j) * 3]; sp[j * 3 + 1] = pos[(g0 + j) * 3 + 1]; sp[j * 3 + 2] = pos[(g0 + j) * 3 + 2];
        lab[j] = -1;
    }
    __syncthreads();
    for (int c = 0; c < ng; ++c) {               // assemble_nodemask, :713-721
        const int lc = lab[c];
        __syncthreads();
        if (lc < 0) {
            const double cx = sp[c * 3], cy = sp[c * 3 + 1], cz = sp[c * 3 + 2];
            for (int j = tid; j < ng; j += 64) {
                if (j == c) continue;
                const double dx = cx - sp[j * 3], dy = cy - sp[j * 3 + 1], dz = cz - sp[j * 3 + 2];
                if (sqrt(dx * dx + dy * dy + dz * dz) < cutoff) lab[j] = c;
            }
            if (tid == 0) lab[c] = c;
        }
        __syncthreads();
    }
    const int smp = tp.node_sample[g0];
    const int ns = tp.sample_ptr[smp + 1] - tp.sample_ptr[smp];
    for (int j = tid; j < ng; j += 64) {
        const int lj = lab[j];
        double sx = 0, sy = 0, sz = 0; int cnt = 0;
        for (int k = 0; k < ng; ++k)
            if (lab[k] == lj) { sx += sp[k * 3]; sy += sp[k * 3 + 1]; sz += sp[k * 3 + 2]; ++cnt; }
        const double ax = sp[j * 3] - sx / cnt, ay = sp[j * 3 + 1] - sy / cnt, az = sp[j * 3 + 2] - sz / cnt;
        const int n = g0 + j;
        pf64[n * 3] = ax; pf64[n * 3 + 1] = ay; pf64[n * 3 + 2] = az;
        pf32[n * 3] = (float)ax; pf32[n * 3 + 1] = (float)ay; pf32[n * 3 + 2] = (float)az;
        labels[n] = tp.node_ref[g0 + lj];
        // node frame: sum_sample(pos_frame) = 0 exactly  =>  b = -a/(n_s-1), a x b = 0, y1 = z1 = 0
        const double k = ns > 1 ? 1.0 + 1.0 / (double)(ns - 1) : 1.0;      // a - b = k * a
        const double dx = k * ax, dy = k * ay, dz = k * az;
        const double nr = sqrt(dx * dx + dy * dy + dz * dz) + EPSF;
        const double ux = dx / nr, uy = dy / nr, uz = dz / nr;
        x1[n * 3] = (float)ux; x1[n * 3 + 1] = (float)uy; x1[n * 3 + 2] = (float)uz;
        pp0[n] = (float)(ax * ux + ay * uy + az * uz);                         // pos_prjt[:,0]; [:,1:] = 0
    }
}

// one thread per inner (same-object) edge: cutoff mask from raw positions (:747-753), edge frame from
// pos_frame (:693-705), masked (:768-771), envelope (:785)
// blk_cnt (optional): [gridDim.x] number of inner edges of this block (256 rows) inside the cutoff - first pass of k_active_list
OARD_KERNEL __global__ __launch_bounds__(256) void k_edge_geo(TopoDev tp, const float* __restrict__ pos, const double* __restrict__ pf64,
                           double cutoff, float* __restrict__ geo, double* __restrict__ d64, int* __restrict__ blk_cnt) {
    const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk_cnt != nullptr) {                    // (before the early exit: every thread of the block takes part)
        int in = 0;
        if (a < tp.A) {
            const int i = tp.act_src[a], j = tp.act_tgt[a];
            const double rx = (double)pos[i * 3] - (double)pos[j * 3], ry = (double)pos[i * 3 + 1] - (double)pos[j * 3 + 1],
                         rz = (double)pos[i * 3 + 2] - (double)pos[j * 3 + 2];
            in = sqrt(rx * rx + ry * ry + rz * rz) < cutoff ? 1 : 0;
        }
        const int cnt = __syncthreads_count(in);
        if (threadIdx.x == 0) blk_cnt[blockIdx.x] = cnt;
    }
    if (a >= tp.A) return;
    const int i = tp.act_src[a], j = tp.act_tgt[a];
    const double rx = (double)pos[i * 3] - (double)pos[j * 3], ry = (double)pos[i * 3 + 1] - (double)pos[j * 3 + 1],
                 rz = (double)pos[i * 3 + 2] - (double)pos[j * 3 + 2];
    const double m = sqrt(rx * rx + ry * ry + rz * rz) < cutoff ? 1.0 : 0.0;
    const double ax = pf64[i * 3], ay = pf64[i * 3 + 1], az = pf64[i * 3 + 2];
    const double bx = pf64[j * 3], by = pf64[j * 3 + 1], bz = pf64[j * 3 + 2];
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    const double dist = sqrt(dx * dx + dy * dy + dz * dz);
    const double nr = dist + EPSF;
    double ux = dx / nr, uy = dy / nr, uz = dz / nr;
    double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
    const double cn = sqrt(cx * cx + cy * cy + cz * cz) + EPSF;
    cx /= cn; cy /= cn; cz /= cn;
    double vx = uy * cz - uz * cy, vy = uz * cx - ux * cz, vz = ux * cy - uy * cx;
    const double d = dist * m;
    const double env = 0.5 * (cos(d * M_PI / cutoff) + 1.0);
    float* g = geo + a * GEO_STRIDE;
    g[0] = (float)d; g[1] = (float)env;
    g[2] = (float)(ux * m); g[3] = (float)(uy * m); g[4] = (float)(uz * m);
    g[5] = (float)(cx * m); g[6] = (float)(cy * m); g[7] = (float)(cz * m);
    g[8] = (float)(vx * m); g[9] = (float)(vy * m); g[10] = (float)(vz * m);
    g[11] = (float)m;
    d64[a] = d;
}

// radial basis (:63-69, 781-782) in float64; one thread per (inner edge, k); writes the RBF buffer that
// EquiMessage re-reads every layer and the rbf section of the initial edge state
// ---- the frame-scalar MLP as a table (see L3T_* in oard_layout.h) --------------------------------------------------------------------
// p: raw parameter block w0[48][3] b0[48] w2[8][48] b2[8] w4[8] b4[1] (LayerOff::l3u); f(x) = lin3((x, 0, 0)) and f'(x), float64
__device__ inline void lin3u_f64(const float* __restrict__ p, double x, double& f, double& df) {
    double h[48], dh[48];
    for (int k = 0; k < 48; ++k) {
        const double w = (double)p[3 * k], z = w * x + (double)p[144 + k], s = 1.0 / (1.0 + exp(-z));
        h[k] = z * s; dh[k] = w * s * (1.0 + z * (1.0 - s));
    }
    f = (double)p[592]; df = 0.0;
    for (int j = 0; j < 8; ++j) {
        double z = (double)p[576 + j], dz = 0.0;
        for (int k = 0; k < 48; ++k) { const double w = (double)p[192 + j * 48 + k]; z += w * h[k]; dz += w * dh[k]; }
        const double s = 1.0 / (1.0 + exp(-z));
        f += (double)p[584 + j] * z * s; df += (double)p[584 + j] * dz * s * (1.0 + z * (1.0 - s));
    }
}
// Hermite evaluation exactly as the node kernel does it (float32)
OARD_DEV float lin3u_table_eval(const float* __restrict__ tab, float x) {
    const float u = x * (1.0f / L3T_H), fl = floorf(u), t = u - fl;
    const int i = (int)fl + L3T_N / 2;
    const float f0 = tab[2 * i], d0 = tab[2 * i + 1], f1 = tab[2 * i + 2], d1 = tab[2 * i + 3];
    const float t2 = t * t, t3 = t2 * t;
    return (2.f * t3 - 3.f * t2 + 1.f) * f0 + (t3 - 2.f * t2 + t) * d0 + (3.f * t2 - 2.f * t3) * f1 + (t3 - t2) * d1;
}
// Two launches over (L3T_N / 256 blocks, layers): fill the points, then check every interval at its midpoint against the float64
// function and set the flag (the last block of a layer to finish decides; tab[flag + 1 .. 3] = worst deviation, range, block counter).
// ~30 us per weight update (a training step repacks every step), spread over 4 CUs per layer.
struct L3tJobs { size_t l3u[OARD_MAX_LAYERS], l3t[OARD_MAX_LAYERS]; };
OARD_KERNEL __global__ __launch_bounds__(256) void k_lin3u_table_fill(float* __restrict__ blob, L3tJobs j) {
    const float* p = blob + j.l3u[blockIdx.y];
    float* tab = blob + j.l3t[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double f, df;
    lin3u_f64(p, ((double)i - L3T_N / 2) * (double)L3T_H, f, df);
    tab[2 * i] = (float)f; tab[2 * i + 1] = (float)(df * (double)L3T_H);
    if (i == 0) {
        lin3u_f64(p, (double)L3T_X, f, df);
        tab[2 * L3T_N] = (float)f; tab[2 * L3T_N + 1] = (float)(df * (double)L3T_H);
        tab[2 * (L3T_N + 1)] = 0.f; tab[2 * (L3T_N + 1) + 1] = 0.f; tab[2 * (L3T_N + 1) + 2] = 0.f; tab[2 * (L3T_N + 1) + 3] = 0.f;
    }
}
OARD_KERNEL __global__ __launch_bounds__(256) void k_lin3u_table_check(float* __restrict__ blob, L3tJobs j) {
    __shared__ float red_e[256], red_f[256];
    const float* p = blob + j.l3u[blockIdx.y];
    float* tab = blob + j.l3t[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x, t = threadIdx.x;
    const double x = ((double)i - L3T_N / 2 + 0.5) * (double)L3T_H;
    double f, df;
    lin3u_f64(p, x, f, df);
    red_e[t] = fabsf(lin3u_table_eval(tab, (float)x) - (float)f);
    red_f[t] = fmaxf(fabsf(tab[2 * i]), fabsf((float)f));
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (t < d) { red_e[t] = fmaxf(red_e[t], red_e[t + d]); red_f[t] = fmaxf(red_f[t], red_f[t + d]); }
        __syncthreads();
    }
    if (t == 0) {
        // non-negative floats order like their bit patterns; a NaN pattern is larger than every finite one and fails the test below
        unsigned* u = reinterpret_cast<unsigned*>(tab + 2 * (L3T_N + 1));
        atomicMax(u + 1, __float_as_uint(red_e[0]));
        atomicMax(u + 2, __float_as_uint(red_f[0]));
        __threadfence();
        if (atomicAdd(u + 3, 1u) == gridDim.x - 1) {
            __threadfence();
            const float e = __uint_as_float(atomicMax(u + 1, 0u)), fm = __uint_as_float(atomicMax(u + 2, 0u));
            tab[2 * (L3T_N + 1)] = (e <= 2e-7f * fmaxf(fm, 1e-30f) && fm < 3e38f) ? 1.0f : 0.0f;
        }
    }
}
// Zeroes up to 64 short rows in one launch (the spare rows of the training tape, oard_hip.hip: forward_impl)
struct ZeroRows { float* p[64]; int n[64]; int count; };
OARD_KERNEL __global__ __launch_bounds__(256) void k_zero_rows(ZeroRows z) {
    if ((int)blockIdx.x >= z.count) return;
    float* p = z.p[blockIdx.x];
    for (int i = threadIdx.x; i < z.n[blockIdx.x]; i += 256) p[i] = 0.f;
}
// Compaction of the inner rows inside the cutoff (ActList, oard_layout.h): block b (256 rows, the blocks of k_edge_geo) adds the
// counts of the blocks in front of it, scans its own 256 flags and writes pre[a] for every row, (row, source) for the active ones.
OARD_KERNEL __global__ __launch_bounds__(256) void k_active_list(TopoDev tp, const float* __restrict__ geo, const int* __restrict__ blk_cnt,
                                                     int* __restrict__ rows, int* __restrict__ src, int* __restrict__ pre, int* __restrict__ n_act) {
    __shared__ int red[256];
    __shared__ int wsum[4];
    int base = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) base += blk_cnt[b];
    red[threadIdx.x] = base;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    base = red[0];
    const long long a = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool on = a < tp.A && geo[a * GEO_STRIDE + 11] > 0.f;
    const unsigned long long bal = __ballot(on);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = base + in_wave;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (a < tp.A) pre[a] = off;
    if (on) { rows[off] = (int)a; src[off] = tp.act_src[a]; }
    if (a == tp.A - 1) { const int tot = off + (on ? 1 : 0); pre[tp.A] = tot; *n_act = tot; }
}
OARD_KERNEL __global__ void k_rbf(TopoDev tp, const double* __restrict__ d64, const float* __restrict__ geo,
                      const float* __restrict__ means, const float* __restrict__ betas, double cutoff,
                      float* __restrict__ rbuf, float* __restrict__ ew, int R, int RP, int H, int WP) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tp.A * RP) return;
    const long long a = i / RP;
    const int k = (int)(i % RP);
    float v = 0.f;
    if (k < R) {
        const double d = d64[a], m = (double)geo[a * GEO_STRIDE + 11];
        double rb = 0.5 * (cos(d * M_PI / cutoff) + 1.0);
        rb = d < cutoff ? rb : 0.0;
        const double q = exp(-d) - (double)means[k];
        v = (float)(rb * exp(-(double)betas[k] * q * q) * m);
        ew[(size_t)a * WP + 3 * H + k] = v;
        if (k == 0)                                  // the row's padding features: consumers multiply them by zero weights,
            for (int c = 3 * H + R; c < WP; ++c) ew[(size_t)a * WP + c] = 0.f;   // so they must be finite
    }
    rbuf[i] = v;
}

// every edge starts as the masked-edge constant row; inner edges are then overwritten
OARD_KERNEL __global__ void k_fill_edges(const float* __restrict__ c0, float* __restrict__ ew, long long E, int WP) {
    const int per = WP / 4;                    // E = number of rows to fill starting at `ew`
    const long long total = E * per;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        st_f4(ew + i * 4, ld_f4(c0 + (i % per) * 4));
}

// =====================================================================================================
// init stages on the column engine
// =====================================================================================================
// z_emb = embedding(h) (:744);  nb = LN0(neighbor_emb.embedding(h)) (:82)
template <class D>
__global__ __launch_bounds__(256) void k_node_embed(TopoDev tp, const float* __restrict__ wb, PackOff po,
                                                    const float* __restrict__ hin, float* __restrict__ zemb,
                                                    float* __restrict__ nb) {
    bool live; const ColId id = col_id(tp.N, live);
    if (!live) return;
    f4 in[1] = {ld_blk(hin, id.c, 16, 0, id.lane)};
    f4 v[D::HT];
#pragma unroll
    for (int t = 0; t < D::HT; ++t) {
        f4 z = dense_tile<1>(wb + po.emb, t, in, id.lane, ld_vec(wb + po.emb_b, t, id.lane));
        if (id.valid) st_blk(zemb, id.c, D::HP, t, id.lane, z);
        v[t] = dense_tile<1>(wb + po.nbemb, t, in, id.lane, ld_vec(wb + po.nbemb_b, t, id.lane));
    }
    layer_norm<D::HT, D::H, false>(v, nullptr, nullptr, id.lane);
    if (id.valid)
#pragma unroll
        for (int t = 0; t < D::HT; ++t) st_blk(nb, id.c, D::HP, t, id.lane, v[t]);
}

// radial_lin on inner edges (:784-786):  f = (Linear(SiLU(Linear(rbf)))) * env  -> ew[:, 2H:3H]
template <class D>
__global__ __launch_bounds__(256) void k_radial_lin(TopoDev tp, const float* __restrict__ wb, PackOff po,
                                                    const float* __restrict__ rbuf, const float* __restrict__ geo,
                                                    float* __restrict__ ew) {
    bool live; const ColId id = col_id(tp.A, live);
    if (!live) return;
    f4 rb[D::RB];
#pragma unroll
    for (int b = 0; b < D::RB; ++b) rb[b] = ld_blk(rbuf, id.c, D::RP, b, id.lane);
    f4 h1[D::HT];
    dense_regs<D::RB, D::HT, true, true>(wb + po.rl0, wb + po.rl0_b, rb, h1, id.lane);
    const float env = geo[id.c * GEO_STRIDE + 1];
    const size_t row = (size_t)id.c * D::WP + 2 * D::H;
#pragma unroll
    for (int t = 0; t < D::HT; ++t) {
        f4 f = dense_tile<D::HT>(wb + po.rl2, t, h1, id.lane, ld_vec(wb + po.rl2_b, t, id.lane));
        if (id.valid && 16 * t + 4 * id.g < D::H) st_f4(ew + row + 16 * t + 4 * id.g, f * env);
    }
}

// NeighborEmb aggregation (:83-89) + s2v.lin1 (:116):  s = z_emb + sum_m f(m->n) * nb[m];  s1 = SiLU(LN0(Linear(s)))
template <class D>
__global__ __launch_bounds__(256) void k_neighbor(TopoDev tp, const float* __restrict__ wb, PackOff po,
                                                  const float* __restrict__ zemb, const float* __restrict__ nb,
                                                  const float* __restrict__ ew, float* __restrict__ s,
                                                  float* __restrict__ s1) {
    // inter-object rows (>= A) are not materialised at this point: their state is the constant row c0row
    bool live; const ColId id = col_id(tp.N, live);
    if (!live) return;
    const int n = id.col, smp = tp.node_sample[n], s0 = tp.sample_ptr[smp], ns = tp.sample_ptr[smp + 1] - s0;
    f4 acc[D::HT];
#pragma unroll
    for (int t = 0; t < D::HT; ++t) acc[t] = f4zero();                 // (the sum on its own, z_emb added once: see k_neighbor_v1)
    const int mx = wave_max(ns);
    for (int k = 0; k < mx; ++k) {
        const int m = s0 + k;
        if (k < ns && m != n) {
            const int r = tp.edge_row[tp.edge_ptr[m] + (n - s0) - (n > m ? 1 : 0)];
            const float* frow = (r < tp.A ? ew + (size_t)r * D::WP : wb + po.c0row) + 2 * D::H;
#pragma unroll
            for (int t = 0; t < D::HT; ++t)
                if (16 * t + 4 * id.g < D::H)
                    acc[t] += ld_f4(frow + 16 * t + 4 * id.g) * ld_blk(nb, m, D::HP, t, id.lane);
        }
    }
#pragma unroll
    for (int t = 0; t < D::HT; ++t) acc[t] += ld_blk(zemb, n, D::HP, t, id.lane);
    if (id.valid)
#pragma unroll
        for (int t = 0; t < D::HT; ++t) st_blk(s, n, D::HP, t, id.lane, acc[t]);
    f4 y[D::HT];
    dense_regs<D::HT, D::HT, false, true>(wb + po.s2v, wb + po.s2v_b, acc, y, id.lane);
    layer_norm<D::HT, D::H, false>(y, nullptr, nullptr, id.lane);
    if (id.valid)
#pragma unroll
        for (int t = 0; t < D::HT; ++t) st_blk(s1, n, D::HP, t, id.lane, silu4(y[t]));
}

// CFConvS2V aggregation (:117-125):  NE1[n][x][:] = sum_{m active->n} f(m->n) * s1[m] * coord_diff(m->n)[x]
template <class D>
__global__ __launch_bounds__(256) void k_s2v_agg(TopoDev tp, const float* __restrict__ s1, const float* __restrict__ ew,
                                                 const float* __restrict__ geo, float* __restrict__ ne1) {
    bool live; const ColId id = col_id(tp.N, live);
    if (!live) return;
    const int n = id.col, a0 = tp.act_ptr[n], cnt = tp.act_ptr[n + 1] - a0;
    const int mx = wave_max(cnt);
    const bool fok = true;
    for (int t = 0; t < D::HT; ++t) {
        f4 ax = f4zero(), ay = f4zero(), az = f4zero();
        if (16 * t + 4 * id.g < D::H && fok) {
            for (int k = 0; k < mx; ++k) {
                if (k < cnt) {
                    const int a = a0 + k, m = tp.act_src[a];
                    const float* g = geo + (size_t)a * GEO_STRIDE;
                    const f4 p = ld_f4(ew + (size_t)a * D::WP + 2 * D::H + 16 * t + 4 * id.g) *
                                 ld_blk(s1, m, D::HP, t, id.lane);
                    ax += p * g[2]; ay += p * g[3]; az += p * g[4];
                }
            }
        }
        if (id.valid) {
            st_blk(ne1, (size_t)n * 3 + 0, D::HP, t, id.lane, ax);
            st_blk(ne1, (size_t)n * 3 + 1, D::HP, t, id.lane, ay);
            st_blk(ne1, (size_t)n * 3 + 2, D::HP, t, id.lane, az);
        }
    }
}

// edge scalarisation + lin3 (:792-806) on inner edges -> ew[:, 0:2H]
template <class D>
__global__ __launch_bounds__(256) void k_scalarize(TopoDev tp, const float* __restrict__ wb, PackOff po,
                                                   const float* __restrict__ ne1, const float* __restrict__ geo,
                                                   float* __restrict__ ew) {
    bool live; const ColId id = col_id(tp.A, live);
    if (!live) return;
    const long long a = id.c;
    const int ni = tp.act_src[a], nj = tp.act_tgt[a];
    const float* g = geo + (size_t)a * GEO_STRIDE;
    const float env = g[1];
    const float ux = g[2], uy = g[3], uz = g[4], cx = g[5], cy = g[6], cz = g[7], vx = g[8], vy = g[9], vz = g[10];
    const float* l3 = wb + po.lin3;
    const float* w0 = l3;
    const float* b0 = l3 + D::H4 * 3;
    const float* w2 = b0 + D::H4;
    const float b2 = w2[D::H4];
    const size_t row = (size_t)a * D::WP;
    // gridDim.y > 1 spreads the hidden tiles of an edge block over several workgroups (small launches)
    for (int t = blockIdx.y; t < D::HT; t += gridDim.y) {
        if (16 * t + 4 * id.g >= D::H) continue;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int node = side == 0 ? ni : nj;
            const f4 n0 = ld_blk(ne1, (size_t)node * 3 + 0, D::HP, t, id.lane);
            const f4 n1 = ld_blk(ne1, (size_t)node * 3 + 1, D::HP, t, id.lane);
            const f4 n2 = ld_blk(ne1, (size_t)node * 3 + 2, D::HP, t, id.lane);
            const f4 S0 = n0 * ux + n1 * uy + n2 * uz;
            f4 S1 = n0 * cx + n1 * cy + n2 * cz;
            const f4 S2 = n0 * vx + n1 * vy + n2 * vz;
            if (!po.signed_scal) S1 = (f4){fabsf(S1.x), fabsf(S1.y), fabsf(S1.z), fabsf(S1.w)};      // reflect_equiv, :794-796
            f4 o = (f4){b2, b2, b2, b2};
            for (int k = 0; k < D::H4; ++k) {
                const float wa = w0[3 * k], wbb = w0[3 * k + 1], wc = w0[3 * k + 2], bb = b0[k], ww = w2[k];
                o += silu4(S0 * wa + S1 * wbb + S2 * wc + bb) * ww;
            }
            o = (o + S0) * env;
            if (id.valid) st_f4(ew + row + side * D::H + 16 * t + 4 * id.g, o);
        }
    }
}

// =====================================================================================================
// per-layer node stages
// =====================================================================================================
// s += pos_expansion(pos_prjt) (:840-841);  xh = LN_gcl(s) (:158);  P = W1a xh + b1;  Q = W1b xh
// (edge_mlp's first Linear, :168, split by input block so its node part is evaluated once per node)
template <class D>
__global__ __launch_bounds__(256) void k_node_pre(TopoDev tp, const float* __restrict__ wb, PackOff po, LayerOff lo,
                                                  const float* __restrict__ s, const float* __restrict__ pp0,
                                                  float* __restrict__ xh, float* __restrict__ P, float* __restrict__ Q) {
    bool live; const ColId id = col_id(tp.N, live);
    if (!live) return;
    const int n = id.col;
    const float pp = pp0[n];
    f4 hid[D::PB];
#pragma unroll
    for (int b = 0; b < D::PB; ++b) {
        const int k0 = 16 * b + 4 * id.g;
        const float* w = wb + po.pe0;
        hid[b].x = k0 + 0 < D::H2 ? silu1(w[(k0 + 0) * 3] * pp) : 0.f;
        hid[b].y = k0 + 1 < D::H2 ? silu1(w[(k0 + 1) * 3] * pp) : 0.f;
        hid[b].z = k0 + 2 < D::H2 ? silu1(w[(k0 + 2) * 3] * pp) : 0.f;
        hid[b].w = k0 + 3 < D::H2 ? silu1(w[(k0 + 3) * 3] * pp) : 0.f;
    }
    f4 v[D::HT];
#pragma unroll
    for (int t = 0; t < D::HT; ++t)
        v[t] = ld_blk(s, n, D::HP, t, id.lane) + dense_tile<D::PB>(wb + po.pe1, t, hid, id.lane, f4zero());      // (summed on its own: see k_node_pre_v1)
    layer_norm<D::HT, D::H, true>(v, wb + lo.ln_g_w, wb + lo.ln_g_b, id.lane);
#pragma unroll
    for (int t = 0; t < D::HT; ++t)
        if (id.valid) st_blk(xh, n, D::HP, t, id.lane, v[t]);
    for (int t = 0; t < D::HT; t += 1) {
        f4 p = ld_vec(wb + lo.b1, t, id.lane), q = f4zero();
        p = dense_tile<D::HT>(wb + lo.W1a, t, v, id.lane, p);
        q = dense_tile<D::HT>(wb + lo.W1b, t, v, id.lane, q);
        if (id.valid) { st_blk(P, n, D::HP, t, id.lane, p); st_blk(Q, n, D::HP, t, id.lane, q); }
    }
}

// GCL node update (:172-183) and EquiMessage's node part (:245):
//   agg = mean_{e: src=n} m_e;  s = xh + node_mlp([xh, agg]);  xq = x_proj(LN_msg(s))
template <class D>
__global__ __launch_bounds__(256) void k_gcl_node(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                  const float* __restrict__ xh, const float* __restrict__ mbuf,
                                                  float* __restrict__ s, float* __restrict__ xq) {
    bool live; const ColId id = col_id(tp.N, live);
    if (!live) return;
    const int n = id.col, smp = tp.node_sample[n];
    const int deg = tp.sample_ptr[smp + 1] - tp.sample_ptr[smp] - 1;
    const size_t e0 = (size_t)tp.edge_ptr[n];
    f4 in[2 * D::HT];
#pragma unroll
    for (int t = 0; t < D::HT; ++t) { in[t] = ld_blk(xh, n, D::HP, t, id.lane); in[D::HT + t] = f4zero(); }
    const int mx = wave_max(deg);
    for (int k = 0; k < mx; ++k)
        if (k < deg)
#pragma unroll
            for (int t = 0; t < D::HT; ++t) in[D::HT + t] += ld_blk(mbuf, e0 + k, D::HP, t, id.lane);
    const float inv = 1.0f / (float)max(deg, 1);            // util_funcs.py:40-44
#pragma unroll
    for (int t = 0; t < D::HT; ++t) in[D::HT + t] *= inv;
    f4 hm[D::HT];
    dense_regs<2 * D::HT, D::HT, true, true>(wb + lo.nm0, wb + lo.nm0b, in, hm, id.lane);
    f4 v[D::HT];
#pragma unroll
    for (int t = 0; t < D::HT; ++t) {
        v[t] = in[t] + dense_tile<D::HT>(wb + lo.nm1, t, hm, id.lane, ld_vec(wb + lo.nm1b, t, id.lane));
        if (id.valid) st_blk(s, n, D::HP, t, id.lane, v[t]);
    }
    layer_norm<D::HT, D::H, true>(v, wb + lo.ln_q_w, wb + lo.ln_q_b, id.lane);
    f4 hq[D::HT];
    dense_regs<D::HT, D::HT, true, false>(wb + lo.xp0, nullptr, v, hq, id.lane);
    for (int t = 0; t < 3 * D::HT; ++t) {
        f4 o = dense_tile<D::HT>(wb + lo.xp2, t, hq, id.lane, f4zero());
        if (id.valid) st_blk(xq, n, 3 * D::HP, t, id.lane, o);
    }
}

// EquiUpdate MLP on (scalar projection, 0, 0): Linear(3,48) SiLU Linear(48,8) SiLU Linear(8,1) (:304-310, 333)
OARD_DEV float lin3u(const float* __restrict__ p, float x) {
    const float* w0 = p;             // [48][3]
    const float* b0 = p + 144;       // [48]
    const float* w2 = p + 192;       // [8][48]
    const float* b2 = p + 576;       // [8]
    const float* w4 = p + 584;       // [8]
    const float b4 = p[592];
    float h2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) h2[j] = b2[j];
    for (int k = 0; k < 48; ++k) {
        const float h = silu1(w0[3 * k] * x + b0[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) h2[j] += w2[j * 48 + k] * h;
    }
    float o = b4;
#pragma unroll
    for (int j = 0; j < 8; ++j) o += w4[j] * silu1(h2[j]);
    return o;
}

// message aggregation (:282-283, 857-859) + first half of EquiUpdate (:326-336):
//   s = (s + sum x_msg)/sqrt2;  vec += sum vec_msg;  (vec1, vec2) = vec_proj(vec);
//   scal = lin3(nodeframe^T vec1) with nodeframe = [x1, 0, 0];  vdot = <vec1, vec2>/sqrt(H)
template <class D>
__global__ __launch_bounds__(256) void k_equi_agg(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                  const float* __restrict__ xmsg, const float* __restrict__ vmsg,
                                                  const float* __restrict__ x1, float* __restrict__ s,
                                                  float* __restrict__ vec, float* __restrict__ v2buf,
                                                  float* __restrict__ scal, float* __restrict__ vdot) {
    bool live; const ColId id = col_id(tp.N, live);
    if (!live) return;
    const int n = id.col, a0 = tp.act_ptr[n], cnt = tp.act_ptr[n + 1] - a0;
    const int mx = wave_max(cnt);
    const float inv_sqrt2 = 0.70710678118654752f, inv_sqrt_h = 1.0f / sqrtf((float)D::H);
    f4 vx[3][D::HT], dx[D::HT];
#pragma unroll
    for (int t = 0; t < D::HT; ++t) { dx[t] = f4zero(); vx[0][t] = f4zero(); vx[1][t] = f4zero(); vx[2][t] = f4zero(); }
    for (int k = 0; k < mx; ++k)
        if (k < cnt) {
            const size_t a = (size_t)a0 + k;
#pragma unroll
            for (int t = 0; t < D::HT; ++t) {
                dx[t] += ld_blk(xmsg, a, D::HP, t, id.lane);
                vx[0][t] += ld_blk(vmsg, a * 3 + 0, D::HP, t, id.lane);
                vx[1][t] += ld_blk(vmsg, a * 3 + 1, D::HP, t, id.lane);
                vx[2][t] += ld_blk(vmsg, a * 3 + 2, D::HP, t, id.lane);
            }
        }
#pragma unroll
    for (int t = 0; t < D::HT; ++t) {
        const f4 sn = (ld_blk(s, n, D::HP, t, id.lane) + dx[t]) * inv_sqrt2;
        if (id.valid) st_blk(s, n, D::HP, t, id.lane, sn);
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            vx[x][t] += ld_blk(vec, (size_t)n * 3 + x, D::HP, t, id.lane);
            if (id.valid) st_blk(vec, (size_t)n * 3 + x, D::HP, t, id.lane, vx[x][t]);
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

// second half of EquiUpdate (:338-346, 861-864):
//   (a, b, c) = xvec_proj([s, scal]);  s += (a + b + vdot)/sqrt2;  vec += c * vec2
template <class D>
__global__ __launch_bounds__(256) void k_equi_upd(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                  const float* __restrict__ scal, const float* __restrict__ vdot,
                                                  const float* __restrict__ v2buf, float* __restrict__ s,
                                                  float* __restrict__ vec) {
    bool live; const ColId id = col_id(tp.N, live);
    if (!live) return;
    const int n = id.col;
    const float inv_sqrt2 = 0.70710678118654752f;
    f4 in[2 * D::HT];
#pragma unroll
    for (int t = 0; t < D::HT; ++t) {
        in[t] = ld_blk(s, n, D::HP, t, id.lane);
        in[D::HT + t] = ld_blk(scal, n, D::HP, t, id.lane);
    }
    f4 hx[D::HT];
    dense_regs<2 * D::HT, D::HT, true, false>(wb + lo.xv0, nullptr, in, hx, id.lane);
    for (int t = 0; t < D::HT; ++t) {
        f4 a = f4zero(), b = f4zero(), c = f4zero();
        dense_tile2<D::HT>(wb + lo.xv2, t, D::HT + t, hx, id.lane, a, b);
        c = dense_tile<D::HT>(wb + lo.xv2, 2 * D::HT + t, hx, id.lane, c);
        const f4 sn = ld_blk(s, n, D::HP, t, id.lane) + (a + b + ld_blk(vdot, n, D::HP, t, id.lane)) * inv_sqrt2;
        if (id.valid) st_blk(s, n, D::HP, t, id.lane, sn);
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const f4 vn = ld_blk(vec, (size_t)n * 3 + x, D::HP, t, id.lane) +
                          c * ld_blk(v2buf, (size_t)n * 3 + x, D::HP, t, id.lane);
            if (id.valid) st_blk(vec, (size_t)n * 3 + x, D::HP, t, id.lane, vn);
        }
    }
}

// =====================================================================================================
// per-layer edge stages (the hot kernels)
// =====================================================================================================
// GCLMessage edge part (:162-170) on ALL edges; columns = edges in source order.
//   h1 = SiLU(W1c ew + P[src] + Q[tgt]);  m = SiLU(W2 h1 + b2);  m *= SiLU(watt.m + batt)
//   mbuf[e] = m;   ew += SiLU(W3 m + b3)
template <class D>
__global__ __launch_bounds__(256) void k_gcl_edge(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                  const float* __restrict__ P, const float* __restrict__ Q,
                                                  float* __restrict__ ew, float* __restrict__ mbuf) {
    bool live; const ColId id = col_id(tp.E, live);
    if (!live) return;
    const size_t e = (size_t)id.c;                      // physical row
    const int src = tp.row_src[e], tgt = tp.row_tgt[e];
    float* erow = ew + e * D::WP + 4 * id.g;

    // stage 1: K-outer over the edge state
    f4 h1[D::HT];
#pragma unroll
    for (int t = 0; t < D::HT; ++t) h1[t] = f4zero();
    const float* w1 = wb + lo.W1c + (size_t)id.lane * 4;
    for (int b = 0; b < D::WB; ++b) {
        const f4 x = ld_f4(erow + 16 * b);
#pragma unroll
        for (int t = 0; t < D::HT; ++t)
            h1[t] = mma_chunk(ld_f4(w1 + ((size_t)t * D::WB + b) * 256), x, h1[t]);
    }
#pragma unroll
    for (int t = 0; t < D::HT; ++t)
        h1[t] = silu4(h1[t] + ld_blk(P, src, D::HP, t, id.lane) + ld_blk(Q, tgt, D::HP, t, id.lane));

    // stage 2 + attention gate
    f4 m[D::HT];
    dense_regs<D::HT, D::HT, true, true>(wb + lo.W2, wb + lo.b2, h1, m, id.lane);
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < D::HT; ++t) {
        const f4 w = ld_vec(wb + lo.watt, t, id.lane);
        part += m[t].x * w.x + m[t].y * w.y + m[t].z * w.z + m[t].w * w.w;
    }
    const float gate = silu1(col_reduce(part) + wb[lo.batt]);
#pragma unroll
    for (int t = 0; t < D::HT; ++t) {
        m[t] *= gate;
        if (id.valid) st_blk(mbuf, (size_t)tp.row_eid[e], D::HP, t, id.lane, m[t]);
    }

    // stage 3: edge-state residual, one output tile at a time
    for (int t = 0; t < D::WB; ++t) {
        f4 acc = dense_tile<D::HT>(wb + lo.W3, t, m, id.lane, ld_vec(wb + lo.b3, t, id.lane));
        const f4 o = ld_f4(erow + 16 * t) + silu4(acc);
        if (id.valid) st_f4(erow + 16 * t, o);
    }
}

// EquiMessage edge part (:247-272) on inner edges; columns = inner edges sorted by target.
//   d1 = SiLU(dir_proj.0 ew);  q = (dir_proj.2 d1 + b) * (rbf_proj rbf);  (x, a2, a3) = (xq[src] + xq[tgt]) * q
//   xmsg = x;   vmsg[k] = (vec[src][k] * a2/sqrt3 + a3 * coord_diff[k]) / sqrt(H)
template <class D>
__global__ __launch_bounds__(256) void k_equi_edge(TopoDev tp, const float* __restrict__ wb, LayerOff lo,
                                                   const float* __restrict__ ew, const float* __restrict__ rbuf,
                                                   const float* __restrict__ geo, const float* __restrict__ xq,
                                                   const float* __restrict__ vec, float* __restrict__ xmsg,
                                                   float* __restrict__ vmsg) {
    bool live; const ColId id = col_id(tp.A, live);
    if (!live) return;
    const size_t a = (size_t)id.c;
    const int src = tp.act_src[a], tgt = tp.act_tgt[a];
    const float* erow = ew + a * D::WP + 4 * id.g;
    const float* g = geo + a * GEO_STRIDE;
    const float ux = g[2], uy = g[3], uz = g[4];
    const float xc = lo.xcross ? 1.0f : 0.0f, cx = xc * g[5], cy = xc * g[6], cz = xc * g[7];      // reflect_equiv = False: + x (x) coord_cross
    const float inv_sqrt3 = 0.57735026918962576f, inv_sqrt_h = 1.0f / sqrtf((float)D::H);

    f4 d1[D::D1T];
#pragma unroll
    for (int t = 0; t < D::D1T; ++t) d1[t] = f4zero();
    const float* w0 = wb + lo.dp0 + (size_t)id.lane * 4;
    for (int b = 0; b < D::WB; ++b) {
        const f4 x = ld_f4(erow + 16 * b);
#pragma unroll
        for (int t = 0; t < D::D1T; ++t)
            d1[t] = mma_chunk(ld_f4(w0 + ((size_t)t * D::WB + b) * 256), x, d1[t]);
    }
#pragma unroll
    for (int t = 0; t < D::D1T; ++t) d1[t] = silu4(d1[t] + ld_vec(wb + lo.dp0b, t, id.lane));

    f4 rb[D::RB];
#pragma unroll
    for (int b = 0; b < D::RB; ++b) rb[b] = ld_blk(rbuf, a, D::RP, b, id.lane);

    for (int tt = 0; tt < D::HT; ++tt) {
        f4 q[3];
#pragma unroll
        for (int th = 0; th < 3; ++th) {
            const int t = th * D::HT + tt;
            f4 acc = dense_tile<D::D1T>(wb + lo.dp2, t, d1, id.lane, ld_vec(wb + lo.dp2b, t, id.lane));
            f4 accr = dense_tile<D::RB>(wb + lo.rbfp, t, rb, id.lane, f4zero());
            q[th] = acc * accr * (ld_blk(xq, src, 3 * D::HP, t, id.lane) + ld_blk(xq, tgt, 3 * D::HP, t, id.lane));
        }
        const f4 a2 = q[1] * inv_sqrt3, a3 = q[2];
        if (id.valid) {
            st_blk(xmsg, a, D::HP, tt, id.lane, q[0]);
            st_blk(vmsg, a * 3 + 0, D::HP, tt, id.lane,
                   (ld_blk(vec, (size_t)src * 3 + 0, D::HP, tt, id.lane) * a2 + a3 * ux + q[0] * cx) * inv_sqrt_h);
            st_blk(vmsg, a * 3 + 1, D::HP, tt, id.lane,
                   (ld_blk(vec, (size_t)src * 3 + 1, D::HP, tt, id.lane) * a2 + a3 * uy + q[0] * cy) * inv_sqrt_h);
            st_blk(vmsg, a * 3 + 2, D::HP, tt, id.lane,
                   (ld_blk(vec, (size_t)src * 3 + 2, D::HP, tt, id.lane) * a2 + a3 * uz + q[0] * cz) * inv_sqrt_h);
        }
    }
}

// =====================================================================================================
// output block (GatedEquivariantBlock :566-576, tail :878-891)
// =====================================================================================================
template <class D>
__global__ __launch_bounds__(256) void k_out(TopoDev tp, const float* __restrict__ wb, PackOff po,
                                             const float* __restrict__ s, const float* __restrict__ vec,
                                             float* __restrict__ dpos, float* __restrict__ hout, int* __restrict__ status) {
    bool live; const ColId id = col_id(tp.N, live);
    if (!live) return;
    const int n = id.col;
    f4 in[2 * D::HT];        // [ s | |vec1_proj(vec)| ]
    float v2s[3];
    {
        f4 vx[3][D::HT];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            float part = 0.f;
#pragma unroll
            for (int t = 0; t < D::HT; ++t) {
                vx[x][t] = ld_blk(vec, (size_t)n * 3 + x, D::HP, t, id.lane);
                const f4 w = ld_vec(wb + po.v2p, t, id.lane);
                part += vx[x][t].x * w.x + vx[x][t].y * w.y + vx[x][t].z * w.z + vx[x][t].w * w.w;
            }
            v2s[x] = col_reduce(part);
        }
#pragma unroll
        for (int t = 0; t < D::HT; ++t) {
            const f4 p0 = dense_tile<D::HT>(wb + po.v1p, t, vx[0], id.lane, f4zero());
            const f4 p1 = dense_tile<D::HT>(wb + po.v1p, t, vx[1], id.lane, f4zero());
            const f4 p2 = dense_tile<D::HT>(wb + po.v1p, t, vx[2], id.lane, f4zero());
            const f4 q = p0 * p0 + p1 * p1 + p2 * p2;
            in[D::HT + t] = (f4){sqrtf(q.x), sqrtf(q.y), sqrtf(q.z), sqrtf(q.w)};
            in[t] = ld_blk(s, n, D::HP, t, id.lane);
        }
    }
    f4 hu[D::HT];
    dense_regs<2 * D::HT, D::HT, true, true>(wb + po.un0, wb + po.un0_b, in, hu, id.lane);
    const f4 xg = dense_tile<D::HT>(wb + po.un2, 0, hu, id.lane, ld_vec(wb + po.un2_b, 0, id.lane));
    // rows 0,1 of tile 0 live in lane group 0: xg.x = scalar output (unused), xg.y = gate
    if (id.valid && id.g == 0) {
        const float gate = xg.y;
        const float d0 = gate * v2s[0], d1 = gate * v2s[1], d2 = gate * v2s[2];
        dpos[n * 3] = d0; dpos[n * 3 + 1] = d1; dpos[n * 3 + 2] = d2;
        if (isnan(d0) || isnan(d1) || isnan(d2)) atomicOr(status, 1);
    }
    f4 sreg[D::HT];
#pragma unroll
    for (int t = 0; t < D::HT; ++t) sreg[t] = in[t];
    const f4 ho = dense_tile<D::HT>(wb + po.embout, 0, sreg, id.lane, ld_vec(wb + po.embout_b, 0, id.lane));
    if (id.valid) st_blk(hout, n, 16, 0, id.lane, ho);
}

// =====================================================================================================
// taps (tests): copy internal buffers out in the reference's node / edge order
// =====================================================================================================
OARD_KERNEL __global__ void k_tap_nodes(TopoDev tp, const float* __restrict__ src, int src_ld, int sections, int sect_pad,
                            int sect_len, float* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cols = sections * sect_len;
    if (i >= (long long)tp.N * cols) return;
    const int n = (int)(i / cols), c = (int)(i % cols);
    dst[(size_t)tp.node_ref[n] * cols + c] = src[(size_t)n * src_ld + (c / sect_len) * sect_pad + (c % sect_len)];
}
OARD_KERNEL __global__ void k_tap_labels(TopoDev tp, const int* __restrict__ labels, float* __restrict__ dst) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < tp.N) dst[tp.node_ref[n]] = (float)labels[n];
}
OARD_KERNEL __global__ void k_tap_edges(TopoDev tp, const float* __restrict__ ew, int WP, int W, float* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tp.E * W) return;
    const long long e = i / W;
    const int c = (int)(i % W);
    const int s = tp.edge_src[e];
    const long long rp = tp.ref_edge_ptr[s] + (e - tp.edge_ptr[s]);
    dst[(size_t)rp * W + c] = ew[(size_t)tp.edge_row[e] * WP + c];
}
