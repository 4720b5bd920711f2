// oard_loss.h — the training caller around the denoising call, fused.
//
// What the reference does per training step in ~150 eager torch ops (+ their autograd backward):
//   EnVariationalDiffusion.forward          oa_reactdiff/diffusion/en_diffusion.py:56-248 (helpers :250-449)
//   DDPMModule.compute_loss / training_step oa_reactdiff/trainer/pl_trainer.py:208-282, 327-347
//   AdamW(amsgrad=True) + clip_grad_norm_   pl_trainer.py:150, 391-418
// Here: k_loss_prep (normalise the batch, CoM-free noise, z_t = alpha x + sigma eps), ONE network call, k_loss_terms (per-sample
// L2 error, the t = 0 likelihood terms, nll, the logged means AND d(mean nll)/d(net output) - the loss is quadratic in the network
// output, so its gradient is closed-form and no autograd graph is needed), k_adamw (one pass over the flat parameter bucket,
// gradient clipping factor folded in).  oareactdiff_amd.loss.DiffusionLoss stays the general differentiable formulation (the
// tests compare the two term by term).
#pragma once
#include "oard_kernels.h"

struct LossPtrs {
    const float* pos[OARD_MAX_OBJECTS];          // [n_k][3]            dataset layout (dataset/base_dataset.py:55-88), reference rows
    const long long* one_hot[OARD_MAX_OBJECTS];  // [n_k][nf_k - 4]
    const long long* charge[OARD_MAX_OBJECTS];   // [n_k][1]
    const float* noise[OARD_MAX_OBJECTS];        // [n_k][nf_k]         raw N(0,1) draws
    float* z[OARD_MAX_OBJECTS];                  // [n_k][nf_k]         noised, normalised representation (network input)
    float* eps[OARD_MAX_OBJECTS];                // [n_k][nf_k]         the noise that was added (CoM-free positions)
    const float* net[OARD_MAX_OBJECTS];          // [n_k][nf_k]         network output
    float* dnet[OARD_MAX_OBJECTS];               // [n_k][nf_k]         d(mean_b nll)/d net
    int node_nf[OARD_MAX_OBJECTS];
};
struct LossCfg {
    float norm_value[3], norm_bias[3];           // _normalizer.py: (x - bias) / value for pos | one_hot | charge
    float scale[OARD_MAX_OBJECTS];               // pl_trainer.py scales
    int pos_only, fixed_mask;                    // fixed_mask bit k: object k gets zero noise (fixed_idx)
    int T;
};

OARD_DEV float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
OARD_DEV float gauss_cdf(float x) { return 0.5f * (1.0f + erff(x * 0.70710678118654752f)); }

// one thread per node: z_t, eps                                    en_diffusion.py:250-306 (noised_representation, sample_*_noise)
OARD_KERNEL __global__ void k_loss_prep(TopoDev tp, LossPtrs lp, LossCfg lc, const float* __restrict__ t_int, const float* __restrict__ gamma) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= tp.N) return;
    const int obj = tp.node_obj[n], row = tp.node_row[n], nf = lp.node_nf[obj];
    const int q = tp.node_sample[n] * tp.n_obj + obj;
    const int g0 = tp.grp_ptr[q], g1 = tp.grp_ptr[q + 1];
    const float ti = t_int[tp.node_tidx[n]];
    const float gm = gamma[(int)(ti + 0.5f)];                            // gamma_module(t) = gamma[round(t T)], t = t_int / T
    const float alpha = sqrtf(sigmoid_f(-gm)), sigma = sqrtf(sigmoid_f(gm));
    const float* R = lp.noise[obj];
    const bool fixed = (lc.fixed_mask >> obj) & 1;
    float m[3] = {0.f, 0.f, 0.f};
    for (int k = g0; k < g1; ++k) {                                      // scatter_mean order = ascending row (_utils.py:22-31)
        const float* r = R + (size_t)tp.node_row[k] * nf;
        m[0] += r[0]; m[1] += r[1]; m[2] += r[2];
    }
    const float inv = 1.0f / (float)(g1 - g0);
    float* Z = lp.z[obj] + (size_t)row * nf;
    float* Eo = lp.eps[obj] + (size_t)row * nf;
    const float* r = R + (size_t)row * nf;
    for (int c = 0; c < nf; ++c) {
        float x, e;
        if (c < 3) {
            x = (lp.pos[obj][(size_t)row * 3 + c] - lc.norm_bias[0]) / lc.norm_value[0];
            e = r[c] - m[c] * inv;
        } else if (c < nf - 1) {
            x = ((float)lp.one_hot[obj][(size_t)row * (nf - 4) + (c - 3)] - lc.norm_bias[1]) / lc.norm_value[1];
            e = lc.pos_only ? 0.f : r[c];
        } else {
            x = ((float)lp.charge[obj][row] - lc.norm_bias[2]) / lc.norm_value[2];
            e = lc.pos_only ? 0.f : r[c];
        }
        if (fixed) e = 0.f;
        Eo[c] = e;
        Z[c] = alpha * x + sigma * e;
    }
}

// one 64-thread block per sample: error terms, t = 0 likelihood terms, nll, logged terms, d(mean nll)/d(net)
//   terms [2 n_obj][B]: rows 0..n_obj-1 = normalised, scaled error per object (before the division by scales + 1e-4 of the log),
//                       rows n_obj.. = un-normalised error per object                                      pl_trainer.py:268-277
OARD_KERNEL __global__ __launch_bounds__(64) void k_loss_terms(TopoDev tp, LossPtrs lp, LossCfg lc, const float* __restrict__ t_int,
                                                   const float* __restrict__ gamma, int B, float* __restrict__ nll,
                                                   float* __restrict__ terms) {
    const int sb = blockIdx.x, lane = threadIdx.x;
    const int n_first = tp.sample_ptr[sb];
    const int b = tp.node_tidx[n_first];                                  // row of t_int / nll this sample belongs to
    const float ti = t_int[b];
    const bool tz = ti == 0.f;
    const float gm = gamma[(int)(ti + 0.5f)];
    const float sigma0 = sqrtf(sigmoid_f(gm));
    float total = 0.f;
    for (int k = 0; k < tp.n_obj; ++k) {
        const int q = sb * tp.n_obj + k, g0 = tp.grp_ptr[q], g1 = tp.grp_ptr[q + 1], nf = lp.node_nf[k], ncat = nf - 4;
        const int ncol = lc.pos_only ? 3 : nf;                            // pos_only: the feature outputs are zeroed before the loss (:212-215)
        float e_all = 0.f, e_pos = 0.f, l_cat = 0.f, l_chg = 0.f;
        for (int i = g0 + lane; i < g1; i += 64) {
            const size_t o = (size_t)tp.node_row[i] * nf;
            const float* E = lp.eps[k] + o;
            const float* Nn = lp.net[k] + o;
            for (int c = 0; c < nf; ++c) {
                const float d = E[c] - (c < ncol ? Nn[c] : 0.f);
                e_all += d * d;
                if (c < 3) e_pos += d * d;
            }
            if (tz) {                                                     // log p(h | z0), en_diffusion.py:389-449 (discretised Gaussians)
                const float* Z = lp.z[k] + o;
                const float s_cat = sigma0 * lc.norm_value[1], s_chg = sigma0 * lc.norm_value[2];
                float lpv[16], mx = -3.0e38f;
                for (int j = 0; j < ncat && j < 16; ++j) {
                    const float cen = (Z[3 + j] * lc.norm_value[1] + lc.norm_bias[1]) - 1.0f;
                    lpv[j] = logf(gauss_cdf((cen + 0.5f) / s_cat) - gauss_cdf((cen - 0.5f) / s_cat) + 1e-10f);
                    mx = fmaxf(mx, lpv[j]);
                }
                float se = 0.f;
                for (int j = 0; j < ncat && j < 16; ++j) se += expf(lpv[j] - mx);
                const float lse = mx + logf(se);
                for (int j = 0; j < ncat && j < 16; ++j)
                    l_cat += (lpv[j] - lse) * (float)lp.one_hot[k][(size_t)tp.node_row[i] * ncat + j];
                const float chg = (float)lp.charge[k][tp.node_row[i]];
                const float est = truncf(Z[nf - 1] * lc.norm_value[2] + lc.norm_bias[2]);      // .long(): truncation, as the reference does
                const float cc = chg - est;
                l_chg += logf(gauss_cdf((cc + 0.5f) / s_chg) - gauss_cdf((cc - 0.5f) / s_chg) + 1e-10f);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            e_all += __shfl_xor(e_all, d, 64); e_pos += __shfl_xor(e_pos, d, 64);
            l_cat += __shfl_xor(l_cat, d, 64); l_chg += __shfl_xor(l_chg, d, 64);
        }
        const float size = (float)(g1 - g0), width = lc.pos_only ? 3.0f : 3.0f + (float)nf;   // pl_trainer.py:236-244: (pos_dim + node_nf) * size
        const float err_t = tz ? 0.f : e_all;
        const float err_n = size > 0.f ? err_t / (width * size) * lc.scale[k] : 0.f;
        const float l0x = (tz && size > 0.f) ? 0.5f * e_pos * lc.scale[k] / (3.0f * size) : 0.f;
        total += err_n + l0x + (tz ? -(l_cat + l_chg) : 0.f);
        if (lane == 0) { terms[(size_t)k * B + b] = err_n; terms[(size_t)(tp.n_obj + k) * B + b] = err_t; }
        // d(mean_b nll)/d net: error_t path 2 (net - eps) scale / (width size), t = 0 path (net - eps) scale / (3 size) on the positions
        const float cA = (!tz && size > 0.f) ? 2.0f * lc.scale[k] / (width * size) / (float)B : 0.f;
        const float cP = (tz && size > 0.f) ? lc.scale[k] / (3.0f * size) / (float)B : 0.f;
        for (int i = g0 + lane; i < g1; i += 64) {
            const size_t o = (size_t)tp.node_row[i] * nf;
            for (int c = 0; c < nf; ++c) {
                const float d = c < ncol ? lp.net[k][o + c] - lp.eps[k][o + c] : 0.f;
                lp.dnet[k][o + c] = c < ncol ? d * (cA + (c < 3 ? cP : 0.f)) : 0.f;
            }
        }
    }
    if (lane == 0) nll[b] = total;
}

// AdamW (torch.optim.AdamW single-tensor semantics, amsgrad optional) over a flat bucket; `gscale` multiplies the gradient first
// (clip_grad_norm_'s factor; 1 = no clipping).  Every derived scalar comes from the host in double precision, rounded once, exactly
// as torch passes its Python-float scalars to the element-wise kernels: decay = 1 - lr wd, w1 = 1 - beta1, w2 = 1 - beta2,
// step_size = lr / (1 - beta1^t), inv_bc2s = 1 / sqrt(1 - beta2^t) (torch divides a tensor by a scalar as a product with its inverse).
OARD_KERNEL __global__ void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                        float* __restrict__ vmax, long long n, float decay, float w1, float beta2, float w2, float eps, float step_size,
                        float inv_bc2s, int amsgrad, float gscale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gr = g[i] * gscale;
    const float x = p[i] * decay;
    const float mi = m[i] + w1 * (gr - m[i]);                             // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * beta2 + w2 * (gr * gr);                       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    m[i] = mi; v[i] = vi;
    float den;
    if (amsgrad) { const float vm = fmaxf(vmax[i], vi); vmax[i] = vm; den = sqrtf(vm) * inv_bc2s + eps; }
    else den = sqrtf(vi) * inv_bc2s + eps;
    p[i] = x - step_size * (mi / den);                                    // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// ---- the optimiser step without a host round trip (round 4) -----------------------------------------------------------------------------
// DDPMModule.configure_gradient_clipping (pl_trainer.py:391-418) decides on the host: max_norm = 1.5 mean(history) + 3 std(history) over the
// last <= 50 gradient norms, clip if the norm exceeds it, push min(norm, max_norm) - which costs every training step a device -> host read
// of the gradient norm BEFORE the optimiser kernel can be launched, and the device then idles while the host prepares the next step
// (1.3 ms of a 69-ms step at B = 64).  Here the decision is one device thread: the history, the optimiser's step count and the count of
// skipped steps live in device memory, and k_adamw_dev takes its scalars from there.
// state (doubles): [0] items in the history, [1] optimiser steps taken, [2] steps skipped, [3] reserved, [4 .. 4 + cap) the history, newest
// first (utils/training_tools.py:6-23 inserts at the front), then 8 doubles of scratch that hold the AdamScal of the current step.
struct AdamScal { float decay, w1, beta2, w2, eps, step_size, inv_bc2s, gscale; int skip; int pad[7]; };

// numpy's pairwise summation for n <= 128 (what np.mean / np.std of the 50-entry history evaluate): bit-identical to the host path
OARD_DEV double np_sum(const double* a, int n) {
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; ++i) r += a[i]; return r; }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// out: grad_norm, max_norm (NaN when not clipping or skipped), gscale, skipped
OARD_KERNEL __global__ void k_clip_decide(double* __restrict__ st, int cap, const float* __restrict__ norm, const float* __restrict__ flag, int clip_on,
                              double lr, double beta1, double beta2, double eps, double wd, float* __restrict__ out) {
#pragma clang fp contract(off)          // the host evaluates these expressions without fused multiply-adds
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    AdamScal* sc = reinterpret_cast<AdamScal*>(st + 4 + cap);
    double* items = st + 4;
    const double g = (double)norm[0];
    const bool skip = flag[0] != 0.f || !isfinite(g);
    const double nan_ = __builtin_nan("");
    if (skip) {
        st[2] += 1.0;
        sc->skip = 1;
        out[0] = (float)g; out[1] = (float)nan_; out[2] = 1.f; out[3] = 1.f;
        return;
    }
    double gscale = 1.0, max_norm = nan_;
    if (clip_on) {
        int n = (int)st[0];
        const double mean = np_sum(items, n) / n;
        double sq[64];
        for (int i = 0; i < n; ++i) { const double d = items[i] - mean; sq[i] = d * d; }
        const double sd = sqrt(np_sum(sq, n) / n);
        max_norm = 1.5 * mean + 3 * sd;
        double push = g;
        if (g > max_norm) { gscale = max_norm / (g + 1e-6); push = max_norm; }
        for (int i = (n < cap ? n : cap - 1); i > 0; --i) items[i] = items[i - 1];
        items[0] = push;
        st[0] = (double)(n < cap ? n + 1 : cap);
    }
    st[1] += 1.0;
    const double step = st[1];
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    sc->decay = (float)(1.0 - lr * wd); sc->w1 = (float)(1.0 - beta1); sc->beta2 = (float)beta2; sc->w2 = (float)(1.0 - beta2);
    sc->eps = (float)eps; sc->step_size = (float)(lr / bc1); sc->inv_bc2s = (float)(1.0 / sqrt(bc2)); sc->gscale = (float)gscale;
    sc->skip = 0;
    out[0] = (float)g; out[1] = (float)max_norm; out[2] = (float)gscale; out[3] = 0.f;
}

// k_adamw with the scalars of the step read from device memory; a skipped step touches nothing
OARD_KERNEL __global__ void k_adamw_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ vmax, long long n, int amsgrad, const AdamScal* __restrict__ sc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || sc->skip) return;
    const float gr = g[i] * sc->gscale;
    const float x = p[i] * sc->decay;
    const float mi = m[i] + sc->w1 * (gr - m[i]);
    const float vi = v[i] * sc->beta2 + sc->w2 * (gr * gr);
    m[i] = mi; v[i] = vi;
    float den;
    if (amsgrad) { const float vm = fmaxf(vmax[i], vi); vmax[i] = vm; den = sqrtf(vm) * sc->inv_bc2s + sc->eps; }
    else den = sqrtf(vi) * sc->inv_bc2s + sc->eps;
    p[i] = x - sc->step_size * (mi / den);
}
