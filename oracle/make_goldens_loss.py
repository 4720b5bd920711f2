"""Generate tests/golden/g8_loss_*.npz from the reference's EnVariationalDiffusion.forward and
DDPMModule.compute_loss arithmetic.  BUILD-CONTAINER ONLY (imports /root/reference).

For each case the reference runs twice on the same recorded randomness (t_int from torch.randint, every
torch.randn draw): in float32 — its network outputs are stored too, so the loss arithmetic can be pinned
exactly by replaying them — and in float64, the parity target of the HIP path (same protocol as the forward
goldens: the float32 reference is its own noise floor).  compute_loss (pl_trainer.py:208-282) cannot be
imported (Lightning); its few lines are restated here on the reference's loss_terms and stored as `nll`."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_stubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from oa_reactdiff.diffusion._normalizer import Normalizer  # noqa: E402
from oa_reactdiff.diffusion._schedule import DiffSchedule, PredefinedNoiseSchedule  # noqa: E402
from oa_reactdiff.diffusion.en_diffusion import EnVariationalDiffusion  # noqa: E402
from oa_reactdiff.dynamics import EGNNDynamics  # noqa: E402
from oa_reactdiff.model import LEFTNet  # noqa: E402

from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict  # noqa: E402


def make_batch(sizes, seed):
    g = torch.Generator().manual_seed(seed)
    reps = []
    B = len(sizes)
    for k in range(3):
        mask = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
        n = mask.numel()
        pos = torch.randn(n, 3, generator=g)
        mean = torch.zeros(B, 3).index_add_(0, mask, pos) / torch.tensor(sizes, dtype=torch.float32).unsqueeze(1)
        pos = pos - mean[mask]
        typ = torch.randint(0, 4, (n,), generator=g)
        one_hot = torch.zeros(n, 5, dtype=torch.long)
        one_hot[torch.arange(n), typ] = 1
        charge = torch.tensor([1, 6, 7, 8])[typ].view(n, 1)
        reps.append({"size": torch.tensor(sizes), "pos": pos, "one_hot": one_hot, "charge": charge, "mask": mask})
    return reps


def compute_loss(ddpm, lt, reps, training, pos_only, scales=(1.0, 2.0, 1.0), loss_type="l2"):
    """pl_trainer.py:208-282 on a loss_terms dict."""
    K = 3
    denoms = [(ddpm.pos_dim if pos_only else ddpm.pos_dim + ddpm.node_nfs[k]) * reps[k]["size"] for k in range(K)]
    err_n = [lt["error_t"][k] / denoms[k] * scales[k] for k in range(K)]
    if loss_type == "l2" and training:
        loss_t = torch.stack(err_n, 0).sum(0)
        l0x = torch.stack([lt["loss_0_x"][k] * scales[k] / (ddpm.pos_dim * reps[k]["size"]) for k in range(K)], 0).sum(0)
        loss_0 = l0x + torch.stack(lt["loss_0_cat"], 0).sum(0) + torch.stack(lt["loss_0_charge"], 0).sum(0)
    else:
        loss_t = torch.stack([-ddpm.T * 0.5 * lt["SNR_weight"] * e for e in lt["error_t"]], 0).sum(0)
        loss_0 = (torch.stack(lt["loss_0_x"], 0).sum(0) + torch.stack(lt["loss_0_cat"], 0).sum(0)
                  + torch.stack(lt["loss_0_charge"], 0).sum(0) + lt["neg_log_constants"])
    nll = loss_t + loss_0 + lt["kl_prior"]
    if not (loss_type == "l2" and training):
        nll = nll - lt["delta_log_px"] - lt["log_pN"]
    return nll


def run(name, sizes, training, pos_only, t_fixed, norm_values, T=100):
    cfg = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=2)
    node_nfs, cnf = [9, 9, 9], 1
    sd = synthetic_state_dict(state_spec(cfg, node_nfs, cnf), cfg, seed=42)
    B = len(sizes)
    cond = torch.zeros(B, 1)
    out = {}
    rec_randn, rec_net, rec_net64 = [], [], []
    real_randn, real_randint = torch.randn, torch.randint
    t_rec = torch.tensor(t_fixed, dtype=torch.long).view(B, 1)
    torch.set_default_dtype(torch.float32)
    batch = make_batch(sizes, 11)            # drawn once, under the float32 default (the draw depends on it)
    for dtype in (torch.float32, torch.float64):
        torch.set_default_dtype(dtype)
        dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=node_nfs, edge_nf=0,
                           condition_nf=cnf, model=LEFTNet, device=torch.device("cpu"))
        dyn.load_state_dict({k: v.to(dtype) if v.is_floating_point() else v for k, v in sd.items()}, strict=True)
        gm = PredefinedNoiseSchedule("polynomial_2", T, 1e-5)
        ddpm = EnVariationalDiffusion(dynamics=dyn, schdule=DiffSchedule(gm, norm_values),
                                      normalizer=Normalizer(norm_values, (0.0, 0.0, 0.0), 3), size_histogram=None,
                                      loss_type="l2", pos_only=pos_only)
        ddpm = ddpm.to(dtype)
        ddpm.train(training)
        reps = [dict(r) for r in batch]
        for r in reps:
            r["pos"] = r["pos"].to(dtype)
        first = dtype == torch.float32
        pos_ = [0]
        fwd = dyn.forward

        def spy_fwd(*a, **kw):
            o = fwd(*a, **kw)
            (rec_net if first else rec_net64).append([x.clone() for x in o[0]])
            return o

        def spy_randn(*a, **kw):
            if first:
                x = real_randn(*a, **kw)
                rec_randn.append(x.clone())
                return x
            x = rec_randn[pos_[0]].to(torch.float64)
            pos_[0] += 1
            return x

        def spy_randint(*a, **kw):
            return t_rec.clone()

        dyn.forward = spy_fwd
        torch.manual_seed(3)
        torch.randn, torch.randint = spy_randn, spy_randint
        try:
            with torch.no_grad():
                lt = ddpm.forward([dict(r) for r in reps], cond.to(dtype))
        finally:
            torch.randn, torch.randint = real_randn, real_randint
        nll = compute_loss(ddpm, lt, reps, training, pos_only)
        tag = "f32" if first else "f64"
        for key in ("error_t", "loss_0_x", "loss_0_cat", "loss_0_charge"):
            for k in range(3):
                out[f"{tag}_{key}{k}"] = lt[key][k].numpy()
        for key in ("SNR_weight", "neg_log_constants", "kl_prior", "t_int"):
            out[f"{tag}_{key}"] = lt[key].numpy()
        out[f"{tag}_delta_log_px"] = np.array(float(lt["delta_log_px"]))
        out[f"{tag}_nll"] = nll.numpy()
        for k in range(3):
            out[f"{tag}_net{k}"] = lt["net_eps_xh"][k].numpy()
    torch.set_default_dtype(torch.float32)
    for k, r in enumerate(batch):
        for f in ("size", "pos", "one_hot", "charge", "mask"):
            out[f"rep{k}_{f}"] = r[f].numpy()
    for i, x in enumerate(rec_randn):
        out[f"randn{i}"] = x.numpy()
    for c, o in enumerate(rec_net):
        for k in range(3):
            out[f"net_call{c}_{k}"] = o[k].numpy()
            out[f"net64_call{c}_{k}"] = rec_net64[c][k].numpy()
    out["meta"] = np.array(json.dumps(dict(name=name, sizes=sizes, training=training, pos_only=pos_only, t_int=t_fixed,
                                            norm_values=list(norm_values), T=T, model_config=cfg, n_randn=len(rec_randn),
                                            n_net_calls=len(rec_net))))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
    print(name, "randn draws", len(rec_randn), "net calls", len(rec_net), "nll f32", out["f32_nll"], "f64", out["f64_nll"])


if __name__ == "__main__":
    run("g8_loss_train", [5, 7, 4], True, False, [37, 0, 100], (2.0, 4.0, 10.0))
    run("g8_loss_eval", [5, 7], False, False, [12, 88], (1.0, 4.0, 10.0))
    run("g8_loss_eval_posonly", [6, 3], False, True, [1, 64], (1.0, 1.0, 1.0))
