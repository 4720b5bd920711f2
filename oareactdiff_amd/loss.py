"""Loss terms of the diffusion model around the denoising call (row N2, forward half).

`DiffusionLoss.loss_terms` mirrors `EnVariationalDiffusion.forward`
(oa_reactdiff/diffusion/en_diffusion.py:56-248, helpers :250-449) and `DiffusionLoss.compute_loss` mirrors
`DDPMModule.compute_loss` (oa_reactdiff/trainer/pl_trainer.py:208-282): noised representation at a random
time step, ONE dynamics call (two in evaluation mode: the t = 0 term is computed separately), per-object L2
error, discretised-Gaussian likelihood of atom types and charges at t = 0, normalisation constants.

The network call is `oareactdiff_amd.EGNNDynamics` (HIP); everything around it is a handful of element-wise
and segmented-sum torch ops on [N, 9] tensors that stay on the device.  With `training=True` the terms are
differentiable: under autograd the network call runs the training-mode HIP forward and `loss.backward()` reaches
the parameters through `oareactdiff_amd.training.DynamicsFunction` (hand-written backward of the edge stages).
Evaluation (`training=False`, what `validation_step` takes) never needs gradients and runs under `torch.no_grad()`.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .graph_tools import get_edges_index, get_n_frag_switch
from .schedule import Schedule

FEATURE_MAPPING = ("pos", "one_hot", "charge")         # _normalizer.py:6


def _segment_sum(x: Tensor, index: Tensor, n: int) -> Tensor:
    """scatter_add(x.sum(-1), index, dim_size=n)  (_utils.py:38-39)"""
    return torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, index, x.sum(-1))


def _cdf(x: Tensor) -> Tensor:
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2)))    # _utils.py:42-43


class DiffusionLoss:
    def __init__(self, dynamics: Callable, noise_schedule: str = "polynomial_2", timesteps: int = 1000,
                 precision: float = 1e-5, norm_values: Sequence[float] = (1.0, 1.0, 1.0),
                 norm_biases: Sequence[float] = (0.0, 0.0, 0.0), pos_only: bool = False,
                 fixed_idx: Optional[List[int]] = None, loss_type: str = "l2",
                 scales: Sequence[float] = (1.0, 2.0, 1.0), pos_dim: int = 3, node_nfs: Optional[List[int]] = None):
        self.dynamics = dynamics
        self.schedule = Schedule(noise_schedule, timesteps, precision)
        self.T = timesteps
        self.norm_values, self.norm_biases = tuple(norm_values), tuple(norm_biases)
        self.pos_only, self.fixed_idx = pos_only, list(fixed_idx or [])
        self.loss_type, self.scales = loss_type, tuple(scales)
        self.pos_dim = getattr(dynamics, "pos_dim", pos_dim)
        self.node_nfs = list(getattr(dynamics, "node_nfs", node_nfs or []))
        self._layouts: "OrderedDict[tuple, tuple]" = OrderedDict()

    def _layout(self, masks: List[Tensor], sizes: List[Tensor], need_edges: bool = True) -> Tuple[Tensor, Optional[Tensor], Tensor]:
        """(combined_mask, edge_index, n_frag_switch) of a batch (en_diffusion.py:75-83), cached by the identity of the batch's
        `mask` / `size` tensors: a loader that hands the same tensors again (fixed-size batches) gets the same three tensors
        back, which is what lets the dynamics reuse its topology (keyed on their addresses) without a host sync.
        need_edges=False (the fused training step, whose kernels walk the implicit complete graph): no edge list is built."""
        key = tuple((t.data_ptr(), t._version, t.numel()) for t in list(masks) + list(sizes)) + (bool(need_edges),)
        hit = self._layouts.get(key)
        if hit is not None:
            self._layouts.move_to_end(key)
            return hit[0]
        if not need_edges and all(t.device.type == "cpu" for t in list(masks) + list(sizes)):
            # host tensors (DDPMTrainer.to_device's copies): numpy, not torch - on a many-core host every torch CPU op that opens an
            # OpenMP region costs milliseconds (measured on the 256-thread host of the MI355X box: 26 ms for these two lines)
            import numpy as np
            cm = np.concatenate([m.numpy().astype(np.int64, copy=False) for m in masks])
            nfs = np.repeat(np.arange(len(sizes), dtype=np.int64), [int(s.numpy().sum()) for s in sizes])
            val = (torch.from_numpy(cm), None, torch.from_numpy(nfs))
        else:
            combined_mask = torch.cat(masks)
            val = (combined_mask, get_edges_index(combined_mask, remove_self_edge=True) if need_edges else None, get_n_frag_switch(sizes))
        self._layouts[key] = (val, list(masks) + list(sizes))       # the key tensors stay alive with the entry
        while len(self._layouts) > 4:
            self._layouts.popitem(last=False)
        return val

    # ---- schedule lookups (gamma_module(t) = gamma[round(t * T)], _schedule.py:127-129) ------------------
    def _gamma(self, t: Tensor) -> Tensor:
        idx = torch.round(t * self.T).long()
        return self.schedule.gamma.to(device=t.device, dtype=t.dtype)[idx]

    # ---- noise (en_diffusion.py:283-306); `draw(shape)` is torch.randn on the device unless injected -------
    def _noise(self, masks: List[Tensor], draw: Callable, n_samples: int, dtype) -> List[Tensor]:
        out = []
        for k, m in enumerate(masks):
            n = m.numel()
            x = draw((n, self.pos_dim)).to(dtype)
            cnt = torch.zeros(n_samples, dtype=dtype, device=m.device).index_add_(0, m, torch.ones(n, dtype=dtype, device=m.device))
            mean = torch.zeros(n_samples, self.pos_dim, dtype=dtype, device=m.device).index_add_(0, m, x) / cnt.clamp(min=1).unsqueeze(1)
            x = x - mean[m]
            h = draw((n, self.node_nfs[k] - self.pos_dim)).to(dtype)
            if self.pos_only:
                h = torch.zeros_like(h)
            out.append(torch.cat([x, h], dim=1))
        for k in self.fixed_idx:
            out[k] = torch.zeros_like(out[k])
        return out

    def _noised(self, xh, masks, gamma, draw, n_samples):
        alpha, sigma = torch.sqrt(torch.sigmoid(-gamma)), torch.sqrt(torch.sigmoid(gamma))     # [B,1]
        eps = self._noise(masks, draw, n_samples, xh[0].dtype)
        z = [alpha[m] * x + sigma[m] * e for x, m, e in zip(xh, masks, eps)]
        return z, eps

    # ---- log p(x, h | z0) without constants (en_diffusion.py:332-449) ------------------------------------
    def _log_pxh_given_z0(self, reps, masks, z, eps, net, gamma, n_samples, epsilon=1e-10):
        pd = self.pos_dim
        log_px = [-0.5 * _segment_sum((e[:, :pd] - o[:, :pd]) ** 2, m, n_samples) for e, o, m in zip(eps, net, masks)]
        z = [v[:, :pd + 5 + 1] for v in z]
        sigma0 = torch.sqrt(torch.sigmoid(gamma))
        s_cat, s_chg = sigma0 * self.norm_values[1], sigma0 * self.norm_values[2]
        log_cat, log_chg = [], []
        for r, m, v in zip(reps, masks, z):
            atoms = r["one_hot"] * self.norm_values[1] + self.norm_biases[1]
            centred = (v[:, pd:-1] * self.norm_values[1] + self.norm_biases[1]) - 1
            lp = torch.log(_cdf((centred + 0.5) / s_cat[m]) - _cdf((centred - 0.5) / s_cat[m]) + epsilon)
            lp = lp - torch.logsumexp(lp, dim=1, keepdim=True)
            log_cat.append(_segment_sum(lp * atoms, m, n_samples))
            charge = r["charge"][:, :1] * self.norm_values[2] + self.norm_biases[2]
            est = (v[:, -1:] * self.norm_values[2] + self.norm_biases[2]).long()       # truncation, as the reference does
            c = charge - est
            lq = torch.log(_cdf((c + 0.5) / s_chg[m]) - _cdf((c - 0.5) / s_chg[m]) + epsilon)
            log_chg.append(_segment_sum(lq, m, n_samples))
        return log_px, log_cat, log_chg

    # ---- EnVariationalDiffusion.forward -----------------------------------------------------------------------
    def loss_terms(self, representations: List[Dict[str, Tensor]], conditions: Tensor, training: bool = False,
                   t_int: Optional[Tensor] = None, draw: Optional[Callable] = None) -> Dict:
        """`t_int` ([B,1] float) and `draw(shape) -> N(0,1) tensor` are injectable for tests; by default they are
        drawn on the device like the reference does."""
        if not training:
            with torch.no_grad():
                return self._loss_terms(representations, conditions, False, t_int, draw)
        return self._loss_terms(representations, conditions, True, t_int, draw)

    def _loss_terms(self, representations, conditions, training, t_int, draw) -> Dict:
        masks = [r["mask"] for r in representations]
        dev = representations[0]["pos"].device
        B = representations[0]["size"].size(0)
        sizes = [r["size"] for r in representations]
        n_nodes = torch.stack(sizes, dim=0).sum(dim=0)
        combined_mask, edge_index, n_frag_switch = self._layout(masks, sizes)
        # normalised copies (the reference normalises the caller's dicts in place; we do not mutate the input)
        reps = [{f: (r[f] - self.norm_biases[j]) / self.norm_values[j] for j, f in enumerate(FEATURE_MAPPING)}
                for r in representations]
        fdt = reps[0]["pos"].dtype
        if draw is None:
            def draw(shape):
                return torch.randn(shape, device=dev)
        delta_log_px = -((n_nodes.sum() - 1) * self.pos_dim) * math.log(self.norm_values[0])
        if t_int is None:
            t_int = torch.randint(0 if training else 1, self.T + 1, size=(B, 1), device=dev).float()
        t_int = t_int.detach().to(device=dev, dtype=fdt)
        t_is_zero = (t_int == 0).to(fdt)
        s, t = (t_int - 1) / self.T, t_int / self.T
        gamma_s, gamma_t = self._gamma(s), self._gamma(t)
        xh = [torch.cat([r[f] for f in FEATURE_MAPPING], dim=1) for r in reps]
        z_t, eps = self._noised(xh, masks, gamma_t, draw, B)
        net, _ = self.dynamics(xh=z_t, edge_index=edge_index, t=t, conditions=conditions, n_frag_switch=n_frag_switch,
                               combined_mask=combined_mask, edge_attr=None)
        net = [o.clone() for o in net]
        if self.pos_only:
            for o in net:
                o[:, self.pos_dim:] = 0
        error_t = [_segment_sum((e - o) ** 2, m, B) for e, o, m in zip(eps, net, masks)]
        snr_weight = (1 - torch.exp(-(gamma_s - gamma_t))).squeeze(1)
        gamma_0 = self._gamma(torch.zeros(B, 1, dtype=fdt, device=dev))
        dof = ((n_nodes - 1) * self.pos_dim).to(dev)
        neg_log_constants = -(dof * (-(0.5 * gamma_0.view(B)) - 0.5 * math.log(2 * math.pi)))
        kl_prior = torch.zeros_like(neg_log_constants)
        if training:
            lp = self._log_pxh_given_z0(reps, masks, z_t, eps, net, gamma_t, B)
            tz = t_is_zero.squeeze()
            loss_0 = [[-v * tz for v in part] for part in lp]
            error_t = [e * (1 - tz) for e in error_t]
        else:
            z_0, eps_0 = self._noised(xh, masks, gamma_0, draw, B)
            net_0, _ = self.dynamics(xh=z_0, edge_index=edge_index, t=torch.zeros_like(s), conditions=conditions,
                                     n_frag_switch=n_frag_switch, combined_mask=combined_mask, edge_attr=None)
            lp = self._log_pxh_given_z0(reps, masks, z_0, eps_0, net_0, gamma_0, B)
            loss_0 = [[-v for v in part] for part in lp]
        return {"delta_log_px": delta_log_px, "error_t": error_t, "SNR_weight": snr_weight, "loss_0_x": loss_0[0],
                "loss_0_cat": loss_0[1], "loss_0_charge": loss_0[2], "neg_log_constants": neg_log_constants,
                "kl_prior": kl_prior, "log_pN": torch.zeros_like(kl_prior), "t_int": t_int.squeeze(),
                "net_eps_xh": net, "eps_xh": eps}

    # ---- DDPMModule.compute_loss -------------------------------------------------------------------------------
    def compute_loss(self, representations: List[Dict[str, Tensor]], conditions: Tensor, training: bool = False,
                     t_int: Optional[Tensor] = None, draw: Optional[Callable] = None,
                     lazy_info: bool = False) -> Tuple[Tensor, Dict[str, float]]:
        """lazy_info: keep the logged means as 0-dim device tensors instead of calling .item() here - the reference's .item()
        (pl_trainer.py:268-277) is a host sync between the forward and the backward pass, during which the GPU runs dry while
        the host enqueues the first backward kernels; DDPMTrainer.training_step converts them after the optimiser step."""
        lt = self.loss_terms(representations, conditions, training=training, t_int=t_int, draw=draw)
        K = len(representations)
        width = [self.pos_dim if self.pos_only else self.pos_dim + self.node_nfs[k] for k in range(K)]
        denoms = [width[k] * representations[k]["size"] for k in range(K)]
        err_n = [lt["error_t"][k] / denoms[k] * self.scales[k] for k in range(K)]
        plain_l2 = self.loss_type == "l2" and training
        if plain_l2:
            loss_t = torch.stack(err_n, dim=0).sum(dim=0)
            l0x = torch.stack([lt["loss_0_x"][k] * self.scales[k] / (self.pos_dim * representations[k]["size"])
                               for k in range(K)], dim=0).sum(dim=0)
            loss_0 = l0x + torch.stack(lt["loss_0_cat"], 0).sum(0) + torch.stack(lt["loss_0_charge"], 0).sum(0)
        else:
            loss_t = torch.stack([-self.T * 0.5 * lt["SNR_weight"] * e for e in lt["error_t"]], dim=0).sum(dim=0)
            loss_0 = (torch.stack(lt["loss_0_x"], 0).sum(0) + torch.stack(lt["loss_0_cat"], 0).sum(0)
                      + torch.stack(lt["loss_0_charge"], 0).sum(0) + lt["neg_log_constants"])
        nll = loss_t + loss_0 + lt["kl_prior"]
        info = {}
        for k in range(K):
            e_n, e_u = err_n[k].detach().mean() / (self.scales[k] + 1e-4), lt["error_t"][k].detach().mean()
            info[f"error_t_{k}"] = e_n if lazy_info else e_n.item()
            info[f"unorm_error_t_{k}"] = e_u if lazy_info else e_u.item()
        if not plain_l2:
            nll = nll - lt["delta_log_px"] - lt["log_pN"]
        return nll, info
