"""CPU oracle for the sampling loop around the denoising call.  TEST INFRASTRUCTURE ONLY.

Restates, in plain numpy/torch, what the reference does in
    oa_reactdiff/diffusion/_schedule.py:9-74, 77-187     noise schedules, gamma lookup, alpha/sigma
    oa_reactdiff/diffusion/en_diffusion.py:459-702       sample, sample_p_zs_given_zt, sample_normal,
                                                         sample_p_xh_given_z0, compute_x_pred
    oa_reactdiff/diffusion/en_diffusion.py:278-305       sample_combined_position_feature_noise
    oa_reactdiff/diffusion/_utils.py:9-42                CoM-free noise, remove_mean_batch
with the dynamics call and the random numbers injected, so that a trajectory can be replayed.
Pinned by tests/golden/g4_sampler.npz (generated from the reference by oracle/make_goldens_sampler.py)."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor


def clip_noise_schedule(alphas2, clip_value=0.001):           # _schedule.py:45-57
    alphas2 = np.concatenate([np.ones(1), alphas2], axis=0)
    alphas_step = np.clip(alphas2[1:] / alphas2[:-1], a_min=clip_value, a_max=1.0)
    return np.cumprod(alphas_step, axis=0)


def polynomial_schedule(timesteps: int, s=1e-4, power=3.0):   # _schedule.py:60-74
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    alphas2 = (1 - np.power(x / steps, power)) ** 2
    alphas2 = clip_noise_schedule(alphas2, clip_value=0.001)
    return (1 - 2 * s) * alphas2 + s


def cosine_beta_schedule(timesteps, s=0.008, raise_to_power: float = 1):   # _schedule.py:9-27
    steps = timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    ac = np.cumprod(1.0 - betas, axis=0)
    return np.power(ac, raise_to_power) if raise_to_power != 1 else ac


def gamma_table(noise_schedule: str, timesteps: int, precision: float) -> Tensor:
    """PredefinedNoiseSchedule.__init__ (_schedule.py:83-125): float32 table of T+1 gammas."""
    if "cosine" in noise_schedule:
        sp = noise_schedule.split("_")
        alphas2 = cosine_beta_schedule(timesteps, raise_to_power=1 if len(sp) == 1 else float(sp[1]))
    elif "polynomial" in noise_schedule:
        alphas2 = polynomial_schedule(timesteps, s=precision, power=float(noise_schedule.split("_")[1]))
    else:
        raise ValueError(noise_schedule)
    sigmas2 = 1 - alphas2
    return torch.from_numpy(-(np.log(alphas2) - np.log(sigmas2))).float()


def gamma_at(table: Tensor, t: Tensor, timesteps: int) -> Tensor:          # _schedule.py:127-129
    return table[torch.round(t * timesteps).long()]


def remove_mean_batch(x: Tensor, idx: Tensor) -> Tensor:                   # _utils.py:9-12
    n = int(idx.max()) + 1 if idx.numel() else 0
    s = torch.zeros(n, x.shape[1], dtype=x.dtype).index_add_(0, idx, x)
    c = torch.zeros(n, dtype=x.dtype).index_add_(0, idx, torch.ones(idx.numel(), dtype=x.dtype)).clamp(min=1)
    return x - (s / c.unsqueeze(1))[idx]


def combined_noise(raw: List[Tensor], masks: List[Tensor], pos_only: bool, pos_dim: int = 3) -> List[Tensor]:
    """en_diffusion.py:278-305 on given raw N(0,1) draws: CoM-free positions per (object, sample)."""
    out = []
    for r, m in zip(raw, masks):
        ex = remove_mean_batch(r[:, :pos_dim], m)
        eh = torch.zeros_like(r[:, pos_dim:]) if pos_only else r[:, pos_dim:]
        out.append(torch.cat([ex, eh], dim=1))
    return out


def sample(dynamics: Callable, table: Tensor, timesteps: int, masks: List[Tensor], n_samples: int,
           noise: Callable[[int], List[Tensor]], conditions: Optional[Tensor], pos_only: bool,
           h0: Optional[List[Tensor]] = None, pos_dim: int = 3, trace: Optional[list] = None):
    """en_diffusion.py:459-560 with identity normaliser.  `dynamics(zt_xh, t) -> List[Tensor]`;
    `noise(i)` returns the i-th set of raw N(0,1) draws (i = 0 initial, 1..T steps T-1..0, T+1 final).
    Returns the list of final [pos | features] per object before the argmax/round post-processing."""
    def inflate(v):   # per-sample value -> per-node column
        return v
    zt = combined_noise(noise(0), masks, pos_only, pos_dim)
    if pos_only:
        zt = [torch.cat([zt[k][:, :pos_dim], h0[k]], dim=1) for k in range(len(masks))]
    call = 1
    for s in reversed(range(timesteps)):
        s_arr = torch.full((n_samples, 1), float(s)) / timesteps
        t_arr = torch.full((n_samples, 1), float(s + 1)) / timesteps
        g_s, g_t = gamma_at(table, s_arr, timesteps), gamma_at(table, t_arr, timesteps)
        sigma2_ts = -torch.expm1(F.softplus(g_s) - F.softplus(g_t))                      # _schedule.py:165-167
        alpha_ts = torch.exp(0.5 * (F.logsigmoid(-g_t) - F.logsigmoid(-g_s)))            # :170-175
        sigma_ts = torch.sqrt(sigma2_ts)
        sigma_s, sigma_t = torch.sqrt(torch.sigmoid(g_s)), torch.sqrt(torch.sigmoid(g_t))
        eps_hat = dynamics(zt, t_arr)
        mu = [zt[k] / alpha_ts[masks[k]] - eps_hat[k] * (sigma2_ts / alpha_ts / sigma_t)[masks[k]]
              for k in range(len(masks))]                                                # en_diffusion.py:614-618
        sigma = sigma_ts * sigma_s / sigma_t                                             # :621
        eps = combined_noise(noise(call), masks, pos_only, pos_dim)
        call += 1
        zs = [mu[k] + sigma[masks[k]] * eps[k] for k in range(len(masks))]               # :645-646
        for k in range(len(masks)):
            zs[k] = torch.cat([remove_mean_batch(zs[k][:, :pos_dim], masks[k]), zs[k][:, pos_dim:]], dim=1)  # :627-631
        zt = zs
        if pos_only:
            zt = [torch.cat([zt[k][:, :pos_dim], h0[k]], dim=1) for k in range(len(masks))]   # :526-530
        if trace is not None:
            trace.append([z.clone() for z in zt])
    # sample_p_xh_given_z0 (:649-702)
    t0 = torch.zeros(n_samples, 1)
    g0 = gamma_at(table, t0, timesteps)
    sigma_x = torch.exp(-(-0.5 * g0))                                                    # SNR(-0.5 gamma_0)
    eps_hat = dynamics(zt, t0)
    sigma_0, alpha_0 = torch.sqrt(torch.sigmoid(g0)), torch.sqrt(torch.sigmoid(-g0))
    mu_x = [1.0 / alpha_0[masks[k]] * (zt[k] - sigma_0[masks[k]] * eps_hat[k]) for k in range(len(masks))]
    eps = combined_noise(noise(call), masks, pos_only, pos_dim)
    return [mu_x[k] + sigma_x[masks[k]] * eps[k] for k in range(len(masks))]


def get_repaint_schedule(resamplings, jump_length, timesteps):            # _schedule.py:206-232
    out, cur = [], 0
    while cur < timesteps:
        if cur + jump_length < timesteps:
            if len(out) > 0:
                out[-1] += jump_length
                out.extend([jump_length] * (resamplings - 1))
            else:
                out.extend([jump_length] * resamplings)
            cur += jump_length
        else:
            residual = timesteps - cur
            if len(out) > 0:
                out[-1] += residual
            else:
                out.append(residual)
            cur += residual
    return list(reversed(out))


def inpaint(dynamics: Callable, table: Tensor, timesteps: int, masks: List[Tensor], n_samples: int,
            noise: Callable[[int], List[Tensor]], conditions: Optional[Tensor], pos_only: bool,
            xh_fixed: List[Tensor], frag_fixed: Sequence[int], resamplings: int, jump_length: int, pos_dim: int = 3):
    """en_diffusion.py:722-883 with identity normaliser; `noise(i)` = i-th set of raw draws in call order."""
    K = len(masks)
    xf = [x.clone() for x in xh_fixed]
    h0 = [x[:, pos_dim:].long() for x in xf]
    for k in range(K):
        xf[k][:, :pos_dim] = remove_mean_batch(xf[k][:, :pos_dim], masks[k])
    cnt = [0]

    def draw():
        i = cnt[0]
        cnt[0] += 1
        return combined_noise(noise(i), masks, pos_only, pos_dim)

    def with_h0(z):
        return [torch.cat([z[k][:, :pos_dim], h0[k].to(z[k].dtype)], dim=1) for k in range(K)] if pos_only else z

    zt = with_h0(draw())
    s = timesteps - 1
    sched = get_repaint_schedule(resamplings, jump_length, timesteps)
    for i, nd in enumerate(sched):
        for j in range(nd):
            s_arr = torch.full((n_samples, 1), float(s)) / timesteps
            t_arr = torch.full((n_samples, 1), float(s + 1)) / timesteps
            g_s, g_t = gamma_at(table, s_arr, timesteps), gamma_at(table, t_arr, timesteps)
            alpha_s, sigma_sv = torch.sqrt(torch.sigmoid(-g_s)), torch.sqrt(torch.sigmoid(g_s))
            eps = draw()
            known = [alpha_s[masks[k]] * xf[k] + sigma_sv[masks[k]] * eps[k] for k in range(K)]        # :269-287
            sigma2_ts = -torch.expm1(F.softplus(g_s) - F.softplus(g_t))
            alpha_ts = torch.exp(0.5 * (F.logsigmoid(-g_t) - F.logsigmoid(-g_s)))
            sigma_ts = torch.sqrt(sigma2_ts)
            sigma_t = torch.sqrt(torch.sigmoid(g_t))
            eps_hat = dynamics(zt, t_arr)
            mu = [zt[k] / alpha_ts[masks[k]] - eps_hat[k] * (sigma2_ts / alpha_ts / sigma_t)[masks[k]] for k in range(K)]
            sigma = sigma_ts * sigma_sv / sigma_t
            eps = draw()
            unknown = [mu[k] + sigma[masks[k]] * eps[k] for k in range(K)]
            unknown = [torch.cat([remove_mean_batch(u[:, :pos_dim], masks[k]), u[:, pos_dim:]], dim=1)
                       for k, u in enumerate(unknown)]
            known, unknown = with_h0(known), with_h0(unknown)
            zt = [known[k] if k in frag_fixed else unknown[k] for k in range(K)]
            if j == nd - 1 and i < len(sched) - 1:
                t = s + jump_length
                tj = torch.full((n_samples, 1), float(t)) / timesteps
                g_tj = gamma_at(table, tj, timesteps)
                sigma2 = -torch.expm1(F.softplus(g_s) - F.softplus(g_tj))
                alpha = torch.exp(0.5 * (F.logsigmoid(-g_tj) - F.logsigmoid(-g_s)))
                eps = draw()
                zt = [alpha[masks[k]] * zt[k] + torch.sqrt(sigma2)[masks[k]] * eps[k] for k in range(K)]
                zt = [torch.cat([remove_mean_batch(z[:, :pos_dim], masks[k]), z[:, pos_dim:]], dim=1) for k, z in enumerate(zt)]
                s = t
            s -= 1
    t0 = torch.zeros(n_samples, 1)
    g0 = gamma_at(table, t0, timesteps)
    sigma_x = torch.exp(0.5 * g0)
    eps_hat = dynamics(zt, t0)
    sigma_0, alpha_0 = torch.sqrt(torch.sigmoid(g0)), torch.sqrt(torch.sigmoid(-g0))
    mu_x = [1.0 / alpha_0[masks[k]] * (zt[k] - sigma_0[masks[k]] * eps_hat[k]) for k in range(K)]
    eps = draw()
    return [mu_x[k] + sigma_x[masks[k]] * eps[k] for k in range(K)]
