// General edge lists: EGNNDynamics.forward on ANY edge_index (oa_reactdiff/dynamics/egnn_dynamics.py:63-72 accepts one;
// utils/_graph_tools.py:31-33 builds incomplete ones with `edge_cutoff`; the reference's own model tests run disconnected and cut graphs,
// tests/model/test_equiv.py:216-230, tests/model/test_subgraphs.py:285-339).  The production path (oard_forward) is built on the
// complete graph per sample - implicit edge ids, inner rows first, the exact-arithmetic node frame; this file is the other path: an
// explicit edge list, CSR gathers by target and by source, the reference's LITERAL node frame, every stage a plain one-thread-per-
// element kernel.  Throughput is not its purpose (production never leaves the complete graph, trainer/train_ts1x.py:106); parity is:
// geometry in float64 (as the production path), every dot product accumulated in float64, activations stored in float32, every
// aggregation in the reference's order (edge order, as index_add_ sums).
//
// Written once for two executors: the stage bodies are functors over a flat index; oard_general.hip launches them as HIP kernels
// (the product), tests/general_host/harness.cpp runs the same functors and the same orchestration in host loops (a CPU test of the
// formulas against the oracle - test infrastructure, never a fallback: the product entry points exist in the HIP library only).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/oard.h"

#ifdef __HIPCC__
#define G_HD __host__ __device__ inline
#else
#define G_HD inline
#endif

namespace oard_general {

constexpr double EPS = 1e-6;                       // leftnet.py:15
constexpr double PI = 3.14159265358979323846;

// canonical parameter order == oareactdiff_amd/spec.py:state_spec == reference state_dict() (the same table as ParamIdx in oard_hip.hip)
struct Params {
    int emb_w, emb_b, embout_w, embout_b, means, betas, nbemb_w, nbemb_b, s2v_w, s2v_b, rl0_w, rl0_b, rl2_w, rl2_b, lin30_w, lin30_b, lin32_w,
        lin32_b, pe0_w, pe1_w, gcl0, msg0, upd0, out0, enc0, dec0, count;
    explicit Params(const oard_config* c) {
        int i = 0;
        emb_w = i++; emb_b = i++; embout_w = i++; embout_b = i++; means = i++; betas = i++;
        nbemb_w = i++; nbemb_b = i++; s2v_w = i++; s2v_b = i++; rl0_w = i++; rl0_b = i++; rl2_w = i++; rl2_b = i++;
        lin30_w = i++; lin30_b = i++; lin32_w = i++; lin32_b = i++; pe0_w = i++; pe1_w = i++;
        i += 2;                                   // distance_embedding (unused in forward)
        gcl0 = i; i += 14 * c->num_layers;        // em0.w em0.b em1.w em1.b nm0.w nm0.b nm1.w nm1.b eot.w eot.b att.w att.b ln.w ln.b
        msg0 = i; i += 9 * c->num_layers;         // dp0.w dp0.b dp2.w dp2.b xp0.w xp2.w rbfp.w ln.w ln.b
        upd0 = i; i += 9 * c->num_layers;         // vp.w xv0.w xv2.w l30.w l30.b l32.w l32.b l34.w l34.b
        i += 2;                                   // last_layer (unused in forward)
        out0 = i; i += 6;                         // v1p.w v2p.w un0.w un0.b un2.w un2.b
        enc0 = i; i += 4 * c->n_obj;              // per object: m0.w m0.b m1.w m1.b
        dec0 = i; i += 4 * c->n_obj;
        count = i;
    }
};

// ---- graph tables -----------------------------------------------------------------------------------------------------------------
// Node ids are the reference's (rows of xh[0], then xh[1], ...).  in_*: the edges grouped by target (edge_index[1]), out_*: by source
// (edge_index[0]); inside a group the edges keep their order in edge_index - the order in which index_add_ / scatter sum them.
struct GraphHost {
    int64_t N = 0, E = 0, G = 0;                   // nodes, edges, (object, combined_mask value) groups of the velocity's CoM removal
    std::vector<int32_t> ei0, ei1, in_ptr, in_list, out_ptr, out_list, sub, node_obj, node_row, node_tidx, node_grp, grp_ptr, grp_list;
    std::vector<int64_t> obj_rows;                 // rows of xh[k]
};
struct Graph {                                     // the same tables behind raw pointers (device or host)
    int64_t N, E, G;
    const int32_t *ei0, *ei1, *in_ptr, *in_list, *out_ptr, *out_list, *sub, *node_obj, *node_row, *node_tidx, *node_grp, *grp_ptr, *grp_list;
};

// combined_mask / n_frag_switch: [N] in the reference's node order; edge_index: [2, E] row-major.  Returns OARD_EINVAL for node ids out of
// range or an n_frag_switch that is not 0 ... n_obj-1 in ascending blocks (compute_frag_index, egnn_dynamics.py:177-182, assumes that).
inline int build_graph(const oard_config* c, const int64_t* cm, const int64_t* nfs, int64_t N, const int64_t* ei, int64_t E, GraphHost& g) {
    if (N < 0 || E < 0 || N > 0x7fffffff || E > 0x7fffffff) return OARD_EINVAL;
    g.N = N; g.E = E;
    g.node_obj.resize(N); g.node_row.resize(N); g.node_tidx.resize(N); g.node_grp.resize(N);
    g.obj_rows.assign(c->n_obj, 0);
    int64_t prev = 0;
    for (int64_t n = 0; n < N; ++n) {
        const int64_t k = nfs[n];
        if (k < prev || k >= c->n_obj || cm[n] < 0 || cm[n] > 0x7fffffff) return OARD_EINVAL;
        prev = k;
        g.node_obj[n] = (int32_t)k;
        g.node_row[n] = (int32_t)g.obj_rows[k]++;
        g.node_tidx[n] = (int32_t)cm[n];
    }
    // groups of the per-object CoM removal: (object, combined_mask value), members in node order (egnn_dynamics.py:147-160, 268-271)
    {
        std::vector<std::pair<std::pair<int64_t, int64_t>, int32_t>> key(N);
        for (int64_t n = 0; n < N; ++n) key[n] = {{nfs[n], cm[n]}, (int32_t)n};
        std::vector<int32_t> order(N);
        for (int64_t n = 0; n < N; ++n) order[n] = (int32_t)n;
        // stable counting by (object, mask): N log N is fine here
        std::vector<int32_t> idx(order);
        std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return key[a].first < key[b].first; });
        g.grp_ptr.clear(); g.grp_list.resize(N);
        int64_t G = 0;
        for (int64_t p = 0; p < N; ++p) {
            if (p == 0 || key[idx[p]].first != key[idx[p - 1]].first) { g.grp_ptr.push_back((int32_t)p); ++G; }
            g.grp_list[p] = idx[p];
            g.node_grp[idx[p]] = (int32_t)(G - 1);
        }
        g.grp_ptr.push_back((int32_t)N);
        g.G = G;
    }
    g.ei0.resize(E); g.ei1.resize(E); g.sub.resize(E);
    g.in_ptr.assign(N + 1, 0); g.out_ptr.assign(N + 1, 0);
    for (int64_t e = 0; e < E; ++e) {
        const int64_t a = ei[e], b = ei[E + e];
        if (a < 0 || a >= N || b < 0 || b >= N) return OARD_EINVAL;
        g.ei0[e] = (int32_t)a; g.ei1[e] = (int32_t)b;
        g.sub[e] = nfs[a] == nfs[b] ? 1 : 0;      // get_subgraph_mask, _graph_tools.py:56-59
        ++g.out_ptr[a + 1]; ++g.in_ptr[b + 1];
    }
    for (int64_t n = 0; n < N; ++n) { g.out_ptr[n + 1] += g.out_ptr[n]; g.in_ptr[n + 1] += g.in_ptr[n]; }
    g.in_list.resize(E); g.out_list.resize(E);
    std::vector<int32_t> ci(g.in_ptr.begin(), g.in_ptr.end() - 1), co(g.out_ptr.begin(), g.out_ptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) { g.out_list[co[g.ei0[e]]++] = (int32_t)e; g.in_list[ci[g.ei1[e]]++] = (int32_t)e; }
    return OARD_OK;
}

// ---- workspace --------------------------------------------------------------------------------------------------------------------
struct Dims {
    int H, R, W, L, Cin, emb, H2, H4;
    explicit Dims(const oard_config* c)
        : H(c->hidden), R(c->num_radial), W(3 * c->hidden + c->num_radial), L(c->num_layers), Cin(c->in_hidden),
          emb(c->in_hidden - (c->condition_time ? 1 : 0) - (c->condition_nf > 0 ? c->condition_nf : 0)), H2(c->hidden / 2), H4(c->hidden / 4) {}
};
struct Workspace {
    double *pos, *pf, *lmean;                      // [N][3] each
    int32_t* labels;                               // [N]
    float *hin, *mask, *dist, *env, *cd, *cc, *cv, *rbf, *nf, *pp, *f, *eH, *m, *nb, *s, *s1, *ne1, *ew, *xh, *agg, *nH, *nH2, *xq, *e3a,
        *e3b, *e3c, *vec, *vec2, *vp, *scalar, *vdot, *xv, *n3H, *v1n, *v2, *xg, *dpos, *hout, *gmean, *att;
    size_t bytes;
};
inline Workspace carve(const oard_config* c, int64_t N, int64_t E, int64_t G, char* base) {
    const Dims d(c);
    Workspace w;
    size_t cur = 0;
    auto take = [&](size_t n, size_t elt) { const size_t o = cur; cur = (cur + n * elt + 255) & ~(size_t)255; return base ? base + o : (char*)nullptr; };
    const size_t n = (size_t)(N > 0 ? N : 1), e = (size_t)(E > 0 ? E : 1), g = (size_t)(G > 0 ? G : 1);
    const size_t H = d.H, R = d.R, W = d.W;
    w.pos = (double*)take(n * 3, 8); w.pf = (double*)take(n * 3, 8); w.lmean = (double*)take(n * 3, 8);
    w.labels = (int32_t*)take(n, 4);
#define F(name, count) w.name = (float*)take(count, 4)
    F(hin, n * d.Cin); F(mask, e); F(dist, e); F(env, e); F(cd, e * 3); F(cc, e * 3); F(cv, e * 3); F(rbf, e * R); F(nf, n * 9); F(pp, n * 3);
    F(f, e * H); F(eH, e * H); F(m, e * H); F(nb, n * H); F(s, n * H); F(s1, n * H); F(ne1, n * 3 * H); F(ew, e * W); F(xh, n * H);
    F(agg, n * H); F(nH, n * H); F(nH2, n * H); F(xq, n * 3 * H); F(e3a, e * 3 * H); F(e3b, e * 3 * H); F(e3c, e * 3 * H);
    F(vec, n * 3 * H); F(vec2, n * 3 * H); F(vp, n * 3 * 2 * H); F(scalar, n * H); F(vdot, n * H); F(xv, n * 3 * H); F(n3H, n * 3 * H);
    F(v1n, n * H); F(v2, n * 3); F(xg, n * 2); F(dpos, n * 3); F(hout, n * d.Cin); F(gmean, g * 3); F(att, e);
#undef F
    w.bytes = cur;
    return w;
}

G_HD float silu_f(float x) { return x / (1.0f + expf(-x)); }

// ---- the one dense layer: Y[row][o] = post(act(sum_seg sum_k X_seg[idx_seg[row]][k] W[o][koff_seg + k] + bias[o])) -------------------
// One thread per (4 rows, 4 outputs): every loaded weight serves four rows, every loaded input four outputs; float64 accumulation.
// Up to three input segments = torch.cat([...], dim=1) of gathered rows.
struct Seg { const float* x; int ld; int K; const int32_t* idx; };
struct Gemm {
    long long rows; int nout; Seg seg[3]; int nseg;
    const float* W; int ldw; const float* bias;
    float* Y; int ldy; int act;                     // act: 0 none, 1 SiLU
    int mode;                                       // 0: Y = v   1: Y += v   2: Y = resid + v
    const float* resid; int ldr;
    const float* rowscale;                          // optional: v *= rowscale[row] (after the activation)
    int ogroups() const { return (nout + 3) / 4; }
    long long threads() const { return ((rows + 3) / 4) * (long long)ogroups(); }
    G_HD void operator()(long long tid) const {
        const int og = (nout + 3) / 4;
        const long long rg = tid / og;
        const int o0 = (int)(tid - rg * og) * 4;
        double acc[4][4];
        for (int r = 0; r < 4; ++r) for (int q = 0; q < 4; ++q) acc[r][q] = 0.0;
        const float* w[4];
        for (int q = 0; q < 4; ++q) w[q] = W + (size_t)(o0 + q < nout ? o0 + q : nout - 1) * ldw;
        int koff = 0;
        for (int sgi = 0; sgi < nseg; ++sgi) {
            const Seg sg = seg[sgi];
            const float* xr[4];
            for (int r = 0; r < 4; ++r) {
                long long row = rg * 4 + r;
                if (row >= rows) row = rows - 1;
                const long long src = sg.idx ? (long long)sg.idx[row] : row;
                xr[r] = sg.x + (size_t)src * sg.ld;
            }
            for (int k = 0; k < sg.K; ++k) {
                const double x0 = xr[0][k], x1 = xr[1][k], x2 = xr[2][k], x3 = xr[3][k];
                for (int q = 0; q < 4; ++q) {
                    const double wk = (double)w[q][koff + k];
                    acc[0][q] += x0 * wk; acc[1][q] += x1 * wk; acc[2][q] += x2 * wk; acc[3][q] += x3 * wk;
                }
            }
            koff += sg.K;
        }
        for (int r = 0; r < 4; ++r) {
            const long long row = rg * 4 + r;
            if (row >= rows) break;
            for (int q = 0; q < 4; ++q) {
                const int o = o0 + q;
                if (o >= nout) break;
                float v = (float)(acc[r][q] + (bias ? (double)bias[o] : 0.0));
                if (act) v = silu_f(v);
                if (rowscale) v *= rowscale[row];
                float* y = Y + (size_t)row * ldy + o;
                if (mode == 1) v = *y + v;
                else if (mode == 2) v = resid[(size_t)row * ldr + o] + v;
                *y = v;
            }
        }
    }
};

// ---- wrapper input: positions, per-object encoder, time / condition columns (egnn_dynamics.py:91-119, _base.py:88-109) --------------
struct Prep {
    Graph g; oard_config c; int emb; int p_enc0;
    const float* const* P; const float* const* xh; const float* t; int t_scalar; const float* cond;
    double* pos; float* hin;
    G_HD void operator()(long long n) const {
        const int k = g.node_obj[n], row = g.node_row[n];
        const int nf = c.node_nf[k], dd = nf - 3, a = c.enc_alias[k];
        const float* x = xh[k] + (size_t)row * nf;
        for (int i = 0; i < 3; ++i) pos[n * 3 + i] = (double)x[i];
        const float *W0 = P[p_enc0 + 4 * a], *b0 = P[p_enc0 + 4 * a + 1], *W1 = P[p_enc0 + 4 * a + 2], *b1 = P[p_enc0 + 4 * a + 3];
        float hid[128];
        for (int q = 0; q < 2 * dd; ++q) {
            double acc = b0[q];
            for (int i = 0; i < dd; ++i) acc += (double)W0[q * dd + i] * (double)x[3 + i];
            hid[q] = silu_f((float)acc);
        }
        float* h = hin + (size_t)n * c.in_hidden;
        for (int o = 0; o < emb; ++o) {
            double acc = b1[o];
            for (int q = 0; q < 2 * dd; ++q) acc += (double)W1[o * 2 * dd + q] * (double)hid[q];
            h[o] = (float)acc;
        }
        int col = emb;
        const int b = g.node_tidx[n];
        if (c.condition_time) h[col++] = t_scalar ? t[0] : t[b];
        for (int q = 0; q < c.condition_nf; ++q) h[col++] = cond[(size_t)b * c.condition_nf + q];
    }
};

// ---- geometry block, float64 (leftnet.py:747-785, 693-722, 812-834) ------------------------------------------------------------------
struct DistMask {                                   // dist < cutoff, times the same-object mask (:747-753)
    Graph g; const double* pos; double cutoff; float* mask;
    G_HD void operator()(long long e) const {
        const int i = g.ei0[e], j = g.ei1[e];
        double r = 0;
        for (int x = 0; x < 3; ++x) { const double dlt = pos[i * 3 + x] - pos[j * 3 + x]; r += dlt * dlt; }
        mask[e] = (sqrt(r) < cutoff && g.sub[e]) ? 1.0f : 0.0f;
    }
};
struct Labels {                                     // assemble_nodemask (:707-722): ONE thread, the reference's sequential sweep
    Graph g; const float* mask; int32_t* labels;
    G_HD void operator()(long long) const {
        for (long long n = 0; n < g.N; ++n) labels[n] = -1;
        int ind = 0;
        for (long long c = 0; c < g.N; ++c) {
            if (labels[c] > -1) continue;
            for (int p = g.out_ptr[c]; p < g.out_ptr[c + 1]; ++p) {
                const int e = g.out_list[p];
                if (mask[e] > 0.f) labels[g.ei1[e]] = ind;
            }
            labels[c] = ind++;
        }
    }
};
struct LabelMean {                                  // scatter_mean(pos, labels) (:760): thread L sums the nodes labelled L in node order
    Graph g; const int32_t* labels; const double* pos; double* lmean;
    G_HD void operator()(long long L) const {
        double s0 = 0, s1 = 0, s2 = 0; long long cnt = 0;
        for (long long n = 0; n < g.N; ++n)
            if (labels[n] == (int32_t)L) { s0 += pos[n * 3]; s1 += pos[n * 3 + 1]; s2 += pos[n * 3 + 2]; ++cnt; }
        const double inv = cnt ? 1.0 / (double)cnt : 0.0;
        lmean[L * 3] = s0 * inv; lmean[L * 3 + 1] = s1 * inv; lmean[L * 3 + 2] = s2 * inv;
    }
};
struct PosFrame {
    const int32_t* labels; const double* pos; const double* lmean; double* pf;
    G_HD void operator()(long long n) const {
        for (int x = 0; x < 3; ++x) pf[n * 3 + x] = pos[n * 3 + x] - lmean[(size_t)labels[n] * 3 + x];
    }
};
struct EdgeGeo {                                    // scalarization (:693-705), masks (:768-771), RBF (:63-69, 781-782), envelope (:785)
    Graph g; const double* pf; const float* mask; double cutoff; int R; const float* means; const float* betas;
    float *dist, *env, *cd, *cc, *cv, *rbf;
    G_HD void operator()(long long e) const {
        const int i = g.ei0[e], j = g.ei1[e];
        const double* a = pf + (size_t)i * 3;
        const double* b = pf + (size_t)j * 3;
        const double mk = mask[e];
        double df[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
        const double radial = df[0] * df[0] + df[1] * df[1] + df[2] * df[2];
        double cr[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        const double nrm = sqrt(radial) + EPS, cn = sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) + EPS;
        double u[3], cx[3], v[3];
        for (int x = 0; x < 3; ++x) { u[x] = df[x] / nrm; cx[x] = cr[x] / cn; }
        v[0] = u[1] * cx[2] - u[2] * cx[1]; v[1] = u[2] * cx[0] - u[0] * cx[2]; v[2] = u[0] * cx[1] - u[1] * cx[0];
        const double d = sqrt(radial) * mk;
        dist[e] = (float)d;
        for (int x = 0; x < 3; ++x) { cd[e * 3 + x] = (float)(u[x] * mk); cc[e * 3 + x] = (float)(cx[x] * mk); cv[e * 3 + x] = (float)(v[x] * mk); }
        const double rb = 0.5 * (cos(d * PI / cutoff) + 1.0);
        env[e] = (float)rb;
        const double rbc = d < cutoff ? rb : 0.0, ex = exp(-d);
        for (int k = 0; k < R; ++k) {
            const double t = ex - (double)means[k];
            rbf[(size_t)e * R + k] = (float)(rbc * exp(-(double)betas[k] * t * t) * mk);
        }
    }
};
struct NodeFrame {                                  // vector() = mean of pos_frame[i] at j over ALL edges (:421-428, 812-814), frame (:823-834)
    Graph g; const double* pf; float* nf; float* pp;
    G_HD void operator()(long long n) const {
        double b[3] = {0, 0, 0};
        const int lo = g.in_ptr[n], hi = g.in_ptr[n + 1];
        for (int p = lo; p < hi; ++p) {
            const int i = g.ei0[g.in_list[p]];
            b[0] += pf[i * 3]; b[1] += pf[i * 3 + 1]; b[2] += pf[i * 3 + 2];
        }
        const double inv = hi > lo ? 1.0 / (double)(hi - lo) : 1.0;
        for (int x = 0; x < 3; ++x) b[x] *= inv;
        const double* a = pf + (size_t)n * 3;
        double x1[3] = {a[0] - b[0], a[1] - b[1], a[2] - b[2]};
        const double n1 = sqrt(x1[0] * x1[0] + x1[1] * x1[1] + x1[2] * x1[2]) + EPS;
        for (int x = 0; x < 3; ++x) x1[x] /= n1;
        double y1[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        const double n2 = sqrt(y1[0] * y1[0] + y1[1] * y1[1] + y1[2] * y1[2]) + EPS;
        for (int x = 0; x < 3; ++x) y1[x] /= n2;
        const double z1[3] = {x1[1] * y1[2] - x1[2] * y1[1], x1[2] * y1[0] - x1[0] * y1[2], x1[0] * y1[1] - x1[1] * y1[0]};
        double pr[3] = {0, 0, 0};
        for (int x = 0; x < 3; ++x) {
            nf[n * 9 + x * 3 + 0] = (float)x1[x]; nf[n * 9 + x * 3 + 1] = (float)y1[x]; nf[n * 9 + x * 3 + 2] = (float)z1[x];
            pr[0] += a[x] * x1[x]; pr[1] += a[x] * y1[x]; pr[2] += a[x] * z1[x];
        }
        for (int k = 0; k < 3; ++k) pp[n * 3 + k] = (float)pr[k];
    }
};

// ---- LayerNorm over H (eps 1e-5), optional affine, optional SiLU behind it ------------------------------------------------------------
struct LayerNorm {
    int H; const float* x; const float* gamma; const float* beta; int silu; float* y;
    G_HD void operator()(long long n) const {
        const float* r = x + (size_t)n * H;
        double mu = 0;
        for (int c = 0; c < H; ++c) mu += r[c];
        mu /= H;
        double var = 0;
        for (int c = 0; c < H; ++c) { const double dlt = r[c] - mu; var += dlt * dlt; }
        const double rs = 1.0 / sqrt(var / H + 1e-5);
        for (int c = 0; c < H; ++c) {
            double v = (r[c] - mu) * rs;
            if (gamma) v = v * (double)gamma[c] + (double)beta[c];
            float o = (float)v;
            if (silu) o = silu_f(o);
            y[(size_t)n * H + c] = o;
        }
    }
};

// ---- init head ------------------------------------------------------------------------------------------------------------------------
struct AggNeighbor {                                // NeighborEmb (:81-89): s += sum_{e: ei1 = n} f_e * nb[ei0]
    Graph g; int H; const float* f; const float* nb; float* s;
    G_HD void operator()(long long tid) const {
        const long long n = tid / H; const int c = (int)(tid - n * H);
        double acc = 0;
        for (int p = g.in_ptr[n]; p < g.in_ptr[n + 1]; ++p) {
            const int e = g.in_list[p];
            acc += (double)(f[(size_t)e * H + c] * nb[(size_t)g.ei0[e] * H + c]);
        }
        s[tid] = (float)((double)s[tid] + acc);
    }
};
struct AggS2V {                                     // CFConvS2V (:104-125): NE1[n][x] = sum_in (f_e * cd_e[x]) * s1[ei0]
    Graph g; int H; const float* f; const float* cd; const float* s1; float* ne1;
    G_HD void operator()(long long tid) const {
        const long long n = tid / H; const int c = (int)(tid - n * H);
        double acc[3] = {0, 0, 0};
        for (int p = g.in_ptr[n]; p < g.in_ptr[n + 1]; ++p) {
            const int e = g.in_list[p];
            const float fe = f[(size_t)e * H + c], sv = s1[(size_t)g.ei0[e] * H + c];
            for (int x = 0; x < 3; ++x) acc[x] += (double)((fe * cd[e * 3 + x]) * sv);
        }
        for (int x = 0; x < 3; ++x) ne1[((size_t)n * 3 + x) * H + c] = (float)acc[x];
    }
};
struct Scalarize {                                  // edge scalarisation + lin3 + assembly of the edge state (:792-809)
    Graph g; int H, R, H4, reflect; const float *ne1, *cd, *cc, *cv, *env, *f, *W0, *b0, *W2, *b2; float* ew;
    G_HD float lin3(const double (&sc)[3]) const {
        double out = b2[0];
        for (int q = 0; q < H4; ++q) {
            const double z = (double)W0[q * 3] * sc[0] + (double)W0[q * 3 + 1] * sc[1] + (double)W0[q * 3 + 2] * sc[2] + (double)b0[q];
            out += (double)W2[q] * (double)silu_f((float)z);
        }
        return (float)(out + sc[0]);
    }
    G_HD void operator()(long long tid) const {
        const long long e = tid / H; const int c = (int)(tid - e * H);
        const int W = 3 * H + R;
        float* row = ew + (size_t)e * W;
        for (int side = 0; side < 2; ++side) {
            const int node = side ? g.ei1[e] : g.ei0[e];
            double sc[3] = {0, 0, 0};
            for (int x = 0; x < 3; ++x) {
                const double v = ne1[((size_t)node * 3 + x) * H + c];
                sc[0] += v * (double)cd[e * 3 + x]; sc[1] += v * (double)cc[e * 3 + x]; sc[2] += v * (double)cv[e * 3 + x];
            }
            sc[0] = (double)(float)sc[0]; sc[1] = (double)(float)sc[1]; sc[2] = (double)(float)sc[2];
            if (reflect) sc[1] = fabs(sc[1]);
            row[side * H + c] = lin3(sc) * env[e];
        }
        row[2 * H + c] = f[(size_t)e * H + c];
    }
};
struct CopyRbf {
    int H, R; const float* rbf; float* ew;
    G_HD void operator()(long long tid) const {
        const long long e = tid / R; const int k = (int)(tid - e * R);
        ew[(size_t)e * (3 * H + R) + 3 * H + k] = rbf[tid];
    }
};

// ---- GCLMessage (:157-183) ------------------------------------------------------------------------------------------------------------
struct Gate {                                       // m *= SiLU(att_mlp(m)); att = the H -> 1 dense layer (SiLU applied), one thread per element
    int H; const float* att; float* m;
    G_HD void operator()(long long tid) const { m[tid] *= att[tid / H]; }
};
struct AggMean {                                    // unsorted_segment_sum(m, ei0) / max(count, 1) (util_funcs.py:27-45)
    Graph g; int H; const float* m; float* agg;
    G_HD void operator()(long long tid) const {
        const long long n = tid / H; const int c = (int)(tid - n * H);
        const int lo = g.out_ptr[n], hi = g.out_ptr[n + 1];
        double acc = 0;
        for (int p = lo; p < hi; ++p) acc += (double)m[(size_t)g.out_list[p] * H + c];
        agg[tid] = (float)(acc / (double)(hi > lo ? hi - lo : 1));
    }
};

// ---- EquiMessage (:244-284, 857-859): messages formed and summed per target node --------------------------------------------------------
struct EquiAgg {
    Graph g; int H, reflect; const float *xq, *rbfh, *w3, *vec, *cd, *cc; float* s; float* vec_out;
    G_HD void operator()(long long tid) const {
        const long long n = tid / H; const int c = (int)(tid - n * H);
        const double is3 = 1.0 / sqrt(3.0), ish = 1.0 / sqrt((double)H), is2 = 1.0 / sqrt(2.0);
        const float* qn = xq + (size_t)n * 3 * H;
        double dx = 0, dv[3] = {0, 0, 0};
        for (int p = g.in_ptr[n]; p < g.in_ptr[n + 1]; ++p) {
            const int e = g.in_list[p], i = g.ei0[e];
            const float* qi = xq + (size_t)i * 3 * H;
            const float* rb = rbfh + (size_t)e * 3 * H;
            const float* ww = w3 + (size_t)e * 3 * H;
            const float x_m = (qi[c] + qn[c]) * (rb[c] * ww[c]);
            const float a2 = (float)((double)((qi[H + c] + qn[H + c]) * (rb[H + c] * ww[H + c])) * is3);
            const float a3 = (qi[2 * H + c] + qn[2 * H + c]) * (rb[2 * H + c] * ww[2 * H + c]);
            dx += (double)x_m;
            for (int x = 0; x < 3; ++x) {
                double vm = (double)vec[((size_t)i * 3 + x) * H + c] * (double)a2 + (double)a3 * (double)cd[e * 3 + x];
                if (!reflect) vm += (double)x_m * (double)cc[e * 3 + x];
                dv[x] += vm * ish;
            }
        }
        s[tid] = (float)(((double)s[tid] + dx) * is2);
        for (int x = 0; x < 3; ++x) {
            const size_t o = ((size_t)n * 3 + x) * H + c;
            vec_out[o] = (float)((double)vec[o] + dv[x]);
        }
    }
};

// ---- EquiUpdate (:325-346, 861-864) ------------------------------------------------------------------------------------------------------
struct UpdScalar {                                  // frame scalars of vec1, lin3 (3 -> 48 -> 8 -> 1), <vec1, vec2>
    int H, reflect; const float* vp; const float* nf; const float *W0, *b0, *W2, *b2, *W4, *b4; float* scalar; float* vdot;
    G_HD void operator()(long long tid) const {
        const long long n = tid / H; const int c = (int)(tid - n * H);
        double v1[3], v2[3];
        for (int x = 0; x < 3; ++x) { v1[x] = vp[((size_t)n * 3 + x) * 2 * H + c]; v2[x] = vp[((size_t)n * 3 + x) * 2 * H + H + c]; }
        double sc[3] = {0, 0, 0};
        for (int x = 0; x < 3; ++x)
            for (int k = 0; k < 3; ++k) sc[k] += v1[x] * (double)nf[n * 9 + x * 3 + k];
        for (int k = 0; k < 3; ++k) sc[k] = (double)(float)sc[k];
        if (reflect) sc[1] = fabs(sc[1]);
        float h1[48];
        for (int q = 0; q < 48; ++q)
            h1[q] = silu_f((float)((double)W0[q * 3] * sc[0] + (double)W0[q * 3 + 1] * sc[1] + (double)W0[q * 3 + 2] * sc[2] + (double)b0[q]));
        double out = b4[0];
        for (int r = 0; r < 8; ++r) {
            double z = b2[r];
            for (int q = 0; q < 48; ++q) z += (double)W2[r * 48 + q] * (double)h1[q];
            out += (double)W4[r] * (double)silu_f((float)z);
        }
        scalar[tid] = (float)out;
        vdot[tid] = (float)((v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2]) / sqrt((double)H));
    }
};
struct UpdApply {
    int H; const float* xv; const float* vdot; const float* vp; const float* vec_in; float* s; float* vec_out;
    G_HD void operator()(long long tid) const {
        const long long n = tid / H; const int c = (int)(tid - n * H);
        const float* r = xv + (size_t)n * 3 * H;
        s[tid] = (float)((double)s[tid] + ((double)r[c] + (double)r[H + c] + (double)vdot[tid]) / sqrt(2.0));
        for (int x = 0; x < 3; ++x) {
            const size_t o = ((size_t)n * 3 + x) * H + c;
            vec_out[o] = vec_in[o] + r[2 * H + c] * vp[((size_t)n * 3 + x) * 2 * H + H + c];
        }
    }
};

// ---- output block (:566-576, 878-891) and the wrapper's tail (egnn_dynamics.py:137-160) ---------------------------------------------------
struct VecNorm {
    int H; const float* t; float* v1n;              // t = vec1_proj(vec) [N][3][H]
    G_HD void operator()(long long tid) const {
        const long long n = tid / H; const int c = (int)(tid - n * H);
        double q = 0;
        for (int x = 0; x < 3; ++x) { const double v = t[((size_t)n * 3 + x) * H + c]; q += v * v; }
        v1n[tid] = (float)sqrt(q);
    }
};
struct Vec2 {
    int H; const float* vec; const float* w; float* v2;
    G_HD void operator()(long long tid) const {         // tid = n * 3 + x
        double acc = 0;
        for (int c = 0; c < H; ++c) acc += (double)w[c] * (double)vec[(size_t)tid * H + c];
        v2[tid] = (float)acc;
    }
};
struct Dpos {
    const float* xg; const float* v2; float* dpos; int32_t* status;
    G_HD void operator()(long long n) const {
        const float gate = xg[n * 2 + 1];
        for (int x = 0; x < 3; ++x) {
            const float v = gate * v2[n * 3 + x];
            dpos[n * 3 + x] = v;
            if (v != v && status) status[0] = 1;    // egnn_dynamics.py:138-143: the caller applies the randn replacement
        }
    }
};
struct GroupMean {
    Graph g; const float* dpos; float* gmean;
    G_HD void operator()(long long q) const {
        double a[3] = {0, 0, 0};
        const int lo = g.grp_ptr[q], hi = g.grp_ptr[q + 1];
        for (int p = lo; p < hi; ++p) for (int x = 0; x < 3; ++x) a[x] += (double)dpos[(size_t)g.grp_list[p] * 3 + x];
        for (int x = 0; x < 3; ++x) gmean[q * 3 + x] = (float)(a[x] / (double)(hi > lo ? hi - lo : 1));
    }
};
struct Post {
    Graph g; oard_config c; int emb; int p_dec0; const float* const* P; const float* dpos; const float* gmean; const float* hout; float* const* out;
    G_HD void operator()(long long n) const {
        const int k = g.node_obj[n], row = g.node_row[n], nf = c.node_nf[k], dd = nf - 3, a = c.enc_alias[k];
        float* o = out[k] + (size_t)row * nf;
        for (int x = 0; x < 3; ++x) o[x] = dpos[n * 3 + x] - gmean[(size_t)g.node_grp[n] * 3 + x];
        const float *W0 = P[p_dec0 + 4 * a], *b0 = P[p_dec0 + 4 * a + 1], *W1 = P[p_dec0 + 4 * a + 2], *b1 = P[p_dec0 + 4 * a + 3];
        const float* h = hout + (size_t)n * c.in_hidden;           // the last 1 + condition_nf columns are dropped (:145)
        float hid[128];
        for (int q = 0; q < 2 * dd; ++q) {
            double acc = b0[q];
            for (int i = 0; i < emb; ++i) acc += (double)W0[q * emb + i] * (double)h[i];
            hid[q] = silu_f((float)acc);
        }
        for (int i = 0; i < dd; ++i) {
            double acc = b1[i];
            for (int q = 0; q < 2 * dd; ++q) acc += (double)W1[i * 2 * dd + q] * (double)hid[q];
            o[3 + i] = (float)acc;
        }
    }
};

// ---- orchestration ----------------------------------------------------------------------------------------------------------------------
// Exec: run(n, functor) executes functor(0 .. n-1) (a kernel launch / a host loop), zero(ptr, bytes), gemm(Gemm) = one dense layer
// (run(k.threads(), k), or the executor's own kernel with the same result up to the order of the float64 sum).  `P`, `xh`, `out` are tables the
// FUNCTORS dereference: device-visible tables for the HIP executor.  P_host is the same parameter table readable by this function.
inline Seg seg(const float* x, int ld, int K, const int32_t* idx = nullptr) { return Seg{x, ld, K, idx}; }
template <class Exec>
int dense(Exec& ex, long long rows, int nout, const Seg* segs, int nseg, const float* W, int ldw, const float* bias, float* Y, int ldy, int act,
          int mode = 0, const float* resid = nullptr, int ldr = 0, const float* rowscale = nullptr) {
    if (rows <= 0) return OARD_OK;
    Gemm k;
    memset(&k, 0, sizeof(k));
    k.rows = rows; k.nout = nout; k.nseg = nseg;
    for (int i = 0; i < nseg; ++i) k.seg[i] = segs[i];
    k.W = W; k.ldw = ldw; k.bias = bias; k.Y = Y; k.ldy = ldy; k.act = act; k.mode = mode; k.resid = resid; k.ldr = ldr; k.rowscale = rowscale;
    return ex.gemm(k);                                   // the executor's dense layer: Gemm itself on plain threads, or a kernel of its own
}
#define G_TRY(x) do { const int rc_ = (x); if (rc_ != OARD_OK) return rc_; } while (0)

template <class Exec>
int forward(Exec& ex, const oard_config* c, const Graph& g, const float* const* P_host, const float* const* P, const float* const* xh,
            const float* t, int t_scalar, const float* cond, float* const* out, const Workspace& w, int32_t* status) {
    const Dims d(c);
    const Params pi(c);
    const int H = d.H, R = d.R, W = d.W, L = d.L, Cin = d.Cin;
    const long long N = g.N, E = g.E;
    for (int k = 0; k < c->n_obj; ++k)
        if (c->node_nf[k] - 3 > 64 || c->node_nf[k] < 4) return OARD_EINVAL;
    if (d.emb < 1 || d.emb > 64 || d.H4 < 1) return OARD_EINVAL;
    if (status) G_TRY(ex.zero(status, sizeof(int32_t)));
    if (N == 0) return OARD_OK;
    auto Pw = [&](int i) { return P_host[i]; };
    // wrapper input + geometry
    G_TRY(ex.run(N, Prep{g, *c, d.emb, pi.enc0, P, xh, t, t_scalar, cond, w.pos, w.hin}));
    G_TRY(ex.run(E, DistMask{g, w.pos, (double)c->cutoff, w.mask}));
    G_TRY(ex.run(1, Labels{g, w.mask, w.labels}));
    G_TRY(ex.run(N, LabelMean{g, w.labels, w.pos, w.lmean}));
    G_TRY(ex.run(N, PosFrame{w.labels, w.pos, w.lmean, w.pf}));
    G_TRY(ex.run(E, EdgeGeo{g, w.pf, w.mask, (double)c->cutoff, R, Pw(pi.means), Pw(pi.betas), w.dist, w.env, w.cd, w.cc, w.cv, w.rbf}));
    G_TRY(ex.run(N, NodeFrame{g, w.pf, w.nf, w.pp}));
    // radial_lin (:784-786)
    { Seg s1[1] = {seg(w.rbf, R, R)}; G_TRY(dense(ex, E, H, s1, 1, Pw(pi.rl0_w), R, Pw(pi.rl0_b), w.eH, H, 1)); }
    { Seg s1[1] = {seg(w.eH, H, H)}; G_TRY(dense(ex, E, H, s1, 1, Pw(pi.rl2_w), H, Pw(pi.rl2_b), w.f, H, 0, 0, nullptr, 0, w.env)); }
    // z_emb (:744) + NeighborEmb (:81-89)
    { Seg s1[1] = {seg(w.hin, Cin, Cin)}; G_TRY(dense(ex, N, H, s1, 1, Pw(pi.emb_w), Cin, Pw(pi.emb_b), w.s, H, 0)); }
    { Seg s1[1] = {seg(w.hin, Cin, Cin)}; G_TRY(dense(ex, N, H, s1, 1, Pw(pi.nbemb_w), Cin, Pw(pi.nbemb_b), w.nH, H, 0)); }
    G_TRY(ex.run(N, LayerNorm{H, w.nH, nullptr, nullptr, 0, w.nb}));
    G_TRY(ex.run(N * H, AggNeighbor{g, H, w.f, w.nb, w.s}));
    // CFConvS2V (:104-125)
    { Seg s1[1] = {seg(w.s, H, H)}; G_TRY(dense(ex, N, H, s1, 1, Pw(pi.s2v_w), H, Pw(pi.s2v_b), w.nH, H, 0)); }
    G_TRY(ex.run(N, LayerNorm{H, w.nH, nullptr, nullptr, 1, w.s1}));
    G_TRY(ex.run(N * H, AggS2V{g, H, w.f, w.cd, w.s1, w.ne1}));
    // edge state (:792-809)
    G_TRY(ex.run(E * H, Scalarize{g, H, R, d.H4, c->reflect_equiv, w.ne1, w.cd, w.cc, w.cv, w.env, w.f, Pw(pi.lin30_w), Pw(pi.lin30_b),
                                  Pw(pi.lin32_w), Pw(pi.lin32_b), w.ew}));
    G_TRY(ex.run(E * R, CopyRbf{H, R, w.rbf, w.ew}));
    G_TRY(ex.zero(w.vec, (size_t)N * 3 * H * sizeof(float)));
    float* vec = w.vec;
    float* vec_b = w.vec2;
    for (int l = 0; l < L; ++l) {
        const int gp = pi.gcl0 + 14 * l, mp = pi.msg0 + 9 * l, up = pi.upd0 + 9 * l;
        // s += pos_expansion(pos_prjt) (legacy: every layer, :840-841)
        { Seg s1[1] = {seg(w.pp, 3, 3)}; G_TRY(dense(ex, N, d.H2, s1, 1, Pw(pi.pe0_w), 3, nullptr, w.nH, d.H2, 1)); }
        { Seg s1[1] = {seg(w.nH, d.H2, d.H2)}; G_TRY(dense(ex, N, H, s1, 1, Pw(pi.pe1_w), d.H2, nullptr, w.s, H, 0, 1)); }
        // GCLMessage
        G_TRY(ex.run(N, LayerNorm{H, w.s, Pw(gp + 12), Pw(gp + 13), 0, w.xh}));
        { Seg s3[3] = {seg(w.xh, H, H, g.ei0), seg(w.xh, H, H, g.ei1), seg(w.ew, W, W)};
          G_TRY(dense(ex, E, H, s3, 3, Pw(gp + 0), 2 * H + W, Pw(gp + 1), w.eH, H, 1)); }
        { Seg s1[1] = {seg(w.eH, H, H)}; G_TRY(dense(ex, E, H, s1, 1, Pw(gp + 2), H, Pw(gp + 3), w.m, H, 1)); }
        { Seg s1[1] = {seg(w.m, H, H)}; G_TRY(dense(ex, E, 1, s1, 1, Pw(gp + 10), H, Pw(gp + 11), w.att, 1, 1)); }
        G_TRY(ex.run(E * H, Gate{H, w.att, w.m}));
        G_TRY(ex.run(N * H, AggMean{g, H, w.m, w.agg}));
        { Seg s2[2] = {seg(w.xh, H, H), seg(w.agg, H, H)}; G_TRY(dense(ex, N, H, s2, 2, Pw(gp + 4), 2 * H, Pw(gp + 5), w.nH, H, 1)); }
        { Seg s1[1] = {seg(w.nH, H, H)}; G_TRY(dense(ex, N, H, s1, 1, Pw(gp + 6), H, Pw(gp + 7), w.s, H, 0, 2, w.xh, H)); }
        { Seg s1[1] = {seg(w.m, H, H)}; G_TRY(dense(ex, E, W, s1, 1, Pw(gp + 8), H, Pw(gp + 9), w.ew, W, 1, 1)); }
        // EquiMessage
        G_TRY(ex.run(N, LayerNorm{H, w.s, Pw(mp + 7), Pw(mp + 8), 0, w.nH2}));
        { Seg s1[1] = {seg(w.nH2, H, H)}; G_TRY(dense(ex, N, H, s1, 1, Pw(mp + 4), H, nullptr, w.nH, H, 1)); }
        { Seg s1[1] = {seg(w.nH, H, H)}; G_TRY(dense(ex, N, 3 * H, s1, 1, Pw(mp + 5), H, nullptr, w.xq, 3 * H, 0)); }
        { Seg s1[1] = {seg(w.rbf, R, R)}; G_TRY(dense(ex, E, 3 * H, s1, 1, Pw(mp + 6), R, nullptr, w.e3a, 3 * H, 0)); }
        { Seg s1[1] = {seg(w.ew, W, W)}; G_TRY(dense(ex, E, 3 * H, s1, 1, Pw(mp + 0), W, Pw(mp + 1), w.e3b, 3 * H, 1)); }
        { Seg s1[1] = {seg(w.e3b, 3 * H, 3 * H)}; G_TRY(dense(ex, E, 3 * H, s1, 1, Pw(mp + 2), 3 * H, Pw(mp + 3), w.e3c, 3 * H, 0)); }
        G_TRY(ex.run(N * H, EquiAgg{g, H, c->reflect_equiv, w.xq, w.e3a, w.e3c, vec, w.cd, w.cc, w.s, vec_b}));
        // EquiUpdate
        { Seg s1[1] = {seg(vec_b, H, H)}; G_TRY(dense(ex, N * 3, 2 * H, s1, 1, Pw(up + 0), H, nullptr, w.vp, 2 * H, 0)); }
        G_TRY(ex.run(N * H, UpdScalar{H, c->reflect_equiv, w.vp, w.nf, Pw(up + 3), Pw(up + 4), Pw(up + 5), Pw(up + 6), Pw(up + 7), Pw(up + 8),
                                      w.scalar, w.vdot}));
        { Seg s2[2] = {seg(w.s, H, H), seg(w.scalar, H, H)}; G_TRY(dense(ex, N, H, s2, 2, Pw(up + 1), 2 * H, nullptr, w.nH, H, 1)); }
        { Seg s1[1] = {seg(w.nH, H, H)}; G_TRY(dense(ex, N, 3 * H, s1, 1, Pw(up + 2), H, nullptr, w.xv, 3 * H, 0)); }
        G_TRY(ex.run(N * H, UpdApply{H, w.xv, w.vdot, w.vp, vec_b, w.s, vec}));
    }
    // output block
    { Seg s1[1] = {seg(vec, H, H)}; G_TRY(dense(ex, N * 3, H, s1, 1, Pw(pi.out0 + 0), H, nullptr, w.n3H, H, 0)); }
    G_TRY(ex.run(N * H, VecNorm{H, w.n3H, w.v1n}));
    G_TRY(ex.run(N * 3, Vec2{H, vec, Pw(pi.out0 + 1), w.v2}));
    { Seg s2[2] = {seg(w.s, H, H), seg(w.v1n, H, H)}; G_TRY(dense(ex, N, H, s2, 2, Pw(pi.out0 + 2), 2 * H, Pw(pi.out0 + 3), w.nH, H, 1)); }
    { Seg s1[1] = {seg(w.nH, H, H)}; G_TRY(dense(ex, N, 2, s1, 1, Pw(pi.out0 + 4), H, Pw(pi.out0 + 5), w.xg, 2, 0)); }
    G_TRY(ex.run(N, Dpos{w.xg, w.v2, w.dpos, status}));
    { Seg s1[1] = {seg(w.s, H, H)}; G_TRY(dense(ex, N, Cin, s1, 1, Pw(pi.embout_w), H, Pw(pi.embout_b), w.hout, Cin, 0)); }
    G_TRY(ex.run(g.G, GroupMean{g, w.dpos, w.gmean}));
    G_TRY(ex.run(N, Post{g, *c, d.emb, pi.dec0, P, w.dpos, w.gmean, w.hout, out}));
    return OARD_OK;
}

}  // namespace oard_general
