"""Noise schedule of the diffusion sampler: the gamma lookup table and the per-step scalars the
sampler kernels need.  Same formulas and the same float32 arithmetic as the reference's
`PredefinedNoiseSchedule` / `DiffSchedule` (oa_reactdiff/diffusion/_schedule.py:9-187); evaluated once
on the host for all T+1 steps, so the sampling loop never reads a schedule value back from the device."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


def _clip_noise_schedule(alphas2: np.ndarray, clip_value: float = 0.001) -> np.ndarray:
    alphas2 = np.concatenate([np.ones(1), alphas2], axis=0)
    step = np.clip(alphas2[1:] / alphas2[:-1], a_min=clip_value, a_max=1.0)
    return np.cumprod(step, axis=0)


def _polynomial(timesteps: int, s: float, power: float) -> np.ndarray:
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    a2 = _clip_noise_schedule((1 - np.power(x / steps, power)) ** 2, 0.001)
    return (1 - 2 * s) * a2 + s


def _cosine(timesteps: int, s: float = 0.008, power: float = 1.0) -> np.ndarray:
    steps = timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    ac = np.cumprod(1.0 - betas, axis=0)
    return np.power(ac, power) if power != 1 else ac


def gamma_table(noise_schedule: str, timesteps: int, precision: float) -> torch.Tensor:
    """float32 [T+1]; `PredefinedNoiseSchedule.gamma`."""
    if "cosine" in noise_schedule:
        sp = noise_schedule.split("_")
        a2 = _cosine(timesteps, power=1.0 if len(sp) == 1 else float(sp[1]))
    elif "polynomial" in noise_schedule:
        sp = noise_schedule.split("_")
        assert len(sp) == 2
        a2 = _polynomial(timesteps, precision, float(sp[1]))
    else:
        raise ValueError(f"noise schedule {noise_schedule!r} is not implemented")
    return torch.from_numpy(-(np.log(a2) - np.log(1 - a2))).float()


@dataclass
class StepCoefficients:
    """z_s = z_t / alpha_ts - eps_hat * c_eps + sigma * eps   (en_diffusion.py:614-646)."""
    alpha_ts: float
    c_eps: float
    sigma: float


@dataclass
class FinalCoefficients:
    """x = (z_0 - sigma_0 * eps_hat) / alpha_0 + sigma_x * eps   (en_diffusion.py:649-702, 704-719)."""
    inv_alpha_0: float
    sigma_0: float
    sigma_x: float


class Schedule:
    def __init__(self, noise_schedule: str = "polynomial_2", timesteps: int = 1000, precision: float = 1e-5):
        self.timesteps = timesteps
        self.gamma = gamma_table(noise_schedule, timesteps, precision)

    def index(self, step: int, n_steps: int) -> int:
        """gamma index of time step/n_steps: round(t * T) on a float32 t (_schedule.py:127-129)."""
        t = torch.tensor(float(step), dtype=torch.float32) / n_steps
        return int(torch.round(t * self.timesteps).long())

    def step(self, s: int, n_steps: int = None) -> StepCoefficients:
        """Transition t = (s+1)/n -> s/n, float32 like the reference (which evaluates these on [B,1] tensors)."""
        n_steps = self.timesteps if n_steps is None else n_steps
        g_s, g_t = self.gamma[self.index(s, n_steps)], self.gamma[self.index(s + 1, n_steps)]
        sigma2_ts = -torch.expm1(F.softplus(g_s) - F.softplus(g_t))
        alpha_ts = torch.exp(0.5 * (F.logsigmoid(-g_t) - F.logsigmoid(-g_s)))
        sigma_ts = torch.sqrt(sigma2_ts)
        sigma_s, sigma_t = torch.sqrt(torch.sigmoid(g_s)), torch.sqrt(torch.sigmoid(g_t))
        return StepCoefficients(float(alpha_ts), float(sigma2_ts / alpha_ts / sigma_t), float(sigma_ts * sigma_s / sigma_t))

    def final(self) -> FinalCoefficients:
        g0 = self.gamma[0]
        return FinalCoefficients(float(1.0 / torch.sqrt(torch.sigmoid(-g0))), float(torch.sqrt(torch.sigmoid(g0))),
                                 float(torch.exp(0.5 * g0)))

    def alpha_sigma(self, step: int, n_steps: int = None):
        """(alpha, sigma) at time step/n_steps  (DiffSchedule.alpha / .sigma, _schedule.py:149-158)."""
        n_steps = self.timesteps if n_steps is None else n_steps
        g = self.gamma[self.index(step, n_steps)]
        return float(torch.sqrt(torch.sigmoid(-g))), float(torch.sqrt(torch.sigmoid(g)))

    def forward_jump(self, s: int, t: int, n_steps: int = None):
        """(alpha_t|s, sigma_t|s) of the forward re-noising s -> t > s  (sample_p_zt_given_zs, en_diffusion.py:1050-1074)."""
        n_steps = self.timesteps if n_steps is None else n_steps
        g_s, g_t = self.gamma[self.index(s, n_steps)], self.gamma[self.index(t, n_steps)]
        sigma2 = -torch.expm1(F.softplus(g_s) - F.softplus(g_t))
        alpha = torch.exp(0.5 * (F.logsigmoid(-g_t) - F.logsigmoid(-g_s)))
        return float(alpha), float(torch.sqrt(sigma2))


def get_repaint_schedule(resamplings: int, jump_length: int, timesteps: int):
    """Number of denoising steps before each jump back (_schedule.py:206-232)."""
    out, cur = [], 0
    while cur < timesteps:
        if cur + jump_length < timesteps:
            if out:
                out[-1] += jump_length
                out.extend([jump_length] * (resamplings - 1))
            else:
                out.extend([jump_length] * resamplings)
            cur += jump_length
        else:
            residual = timesteps - cur
            if out:
                out[-1] += residual
            else:
                out.append(residual)
            cur += residual
    return list(reversed(out))
