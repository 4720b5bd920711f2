"""Generate tests/golden/g9_grad_*.npz: parameter gradients of one training step of the REFERENCE.
BUILD-CONTAINER ONLY (imports /root/reference).

A training step is DDPMModule.training_step (oa_reactdiff/trainer/pl_trainer.py:327-347): loss = compute_loss(batch)[0].mean(0)
with EnVariationalDiffusion.forward in training mode (oa_reactdiff/diffusion/en_diffusion.py:56-248), then loss.backward().
t_int and every torch.randn draw are fixed / recorded (as in make_goldens_loss.py) so the HIP path can replay the step.
The reference runs in float64 (the parity target, same protocol as the forward goldens) and in float32 (its own
noise floor, stored for information: `f32_grad.*`).  The oracle's autograd gradients (oracle/leftnet_oracle.py under
torch autograd, float64) are checked against the reference's here: this pins the oracle's backward.
compute_loss cannot be imported (Lightning); its few lines are restated in make_goldens_loss.compute_loss."""
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
sys.path.insert(0, HERE)

from oa_reactdiff.diffusion._normalizer import Normalizer  # noqa: E402
from oa_reactdiff.diffusion._schedule import DiffSchedule, PredefinedNoiseSchedule  # noqa: E402
from oa_reactdiff.diffusion.en_diffusion import EnVariationalDiffusion  # noqa: E402
from oa_reactdiff.dynamics import EGNNDynamics  # noqa: E402
from oa_reactdiff.model import LEFTNet  # noqa: E402

import leftnet_oracle as oracle  # noqa: E402
from make_goldens_loss import compute_loss, make_batch  # noqa: E402
from oareactdiff_amd.loss import DiffusionLoss  # noqa: E402
from oareactdiff_amd.spec import PRODUCTION_LEFTNET_CONFIG, state_spec, synthetic_state_dict  # noqa: E402


FULL, SAMPLE = 4096, 2048


def sample_index(key, n):
    g = torch.Generator().manual_seed(abs(hash_name(key)) % (2 ** 31))
    return torch.randperm(n, generator=g)[:SAMPLE].sort().values


def hash_name(s):
    h = 0
    for c in s.encode():
        h = (h * 131 + c) % 1000000007
    return h


ONLY = [a for a in sys.argv[1:] if not a.startswith("--")]       # optional case names: generate only those


def run(name, sizes, t_fixed, norm_values, cfg, pos_scale=1.0, T=100, pos_only=False):
    if ONLY and name not in ONLY:
        return
    node_nfs, cnf = [9, 9, 9], 1
    sd = synthetic_state_dict(state_spec(cfg, node_nfs, cnf), cfg, seed=42)
    B = len(sizes)
    cond = torch.zeros(B, 1)
    out = {}
    rec_randn = []
    real_randn, real_randint = torch.randn, torch.randint
    t_rec = torch.tensor(t_fixed, dtype=torch.long).view(B, 1)
    torch.set_default_dtype(torch.float32)
    batch = make_batch(sizes, 11)
    for r in batch:
        r["pos"] = r["pos"] * pos_scale
    grads = {}
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
        ddpm.train(True)
        reps = [dict(r) for r in batch]
        for r in reps:
            r["pos"] = r["pos"].to(dtype)
        first = dtype == torch.float32
        pos_ = [0]

        def spy_randn(*a, **kw):
            if first:
                x = real_randn(*a, **kw)
                rec_randn.append(x.clone())
                return x
            x = rec_randn[pos_[0]].to(torch.float64)
            pos_[0] += 1
            return x

        torch.manual_seed(3)
        torch.randn, torch.randint = spy_randn, (lambda *a, **kw: t_rec.clone())
        try:
            lt = ddpm.forward([dict(r) for r in reps], cond.to(dtype))
        finally:
            torch.randn, torch.randint = real_randn, real_randint
        nll = compute_loss(ddpm, lt, reps, True, pos_only)
        loss = nll.mean(0)                                              # pl_trainer.py:329
        loss.backward()
        tag = "f32" if first else "f64"
        out[f"{tag}_loss"] = loss.detach().numpy()
        out[f"{tag}_nll"] = nll.detach().numpy()
        g = {}
        for k, p in dyn.named_parameters():
            if p.grad is not None:
                g[k] = p.grad.detach().clone()
        grads[tag] = g
        if not first:
            # compact fixture: tensors up to FULL entries in full; larger ones as row sums, column sums, the flat
            # L2 norm and SAMPLE pseudo-randomly chosen entries (any wrong entry moves a row sum and a column sum)
            for k, v in g.items():
                v = v.double()
                out[f"gnorm.{k}"] = np.array([float(v.norm()), float(v.abs().max())])
                if v.numel() <= FULL:
                    out[f"gfull.{k}"] = v.numpy()
                else:
                    m2 = v.reshape(v.shape[0], -1)
                    out[f"grow.{k}"] = m2.sum(dim=1).numpy()
                    out[f"gcol.{k}"] = m2.sum(dim=0).numpy()
                    idx = sample_index(k, v.numel())
                    out[f"gidx.{k}"] = idx.numpy()
                    out[f"gval.{k}"] = v.reshape(-1)[idx].numpy()
        if not first:
            for k in range(3):
                out[f"{tag}_net{k}"] = lt["net_eps_xh"][k].detach().numpy()
    torch.set_default_dtype(torch.float32)

    # ---- pin the oracle's backward: the same step with oracle.dynamics_forward under autograd, float64 ----------
    sd64 = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "radial_emb" not in k else v) for k, v in sd.items()}
    it = iter(rec_randn)

    def oracle_dyn(xh, edge_index, t, conditions, n_frag_switch, combined_mask, edge_attr=None):
        return oracle.dynamics_forward(sd64, cfg, xh, edge_index, t, conditions, n_frag_switch, combined_mask, cnf,
                                       nodeframe="literal"), None
    oracle_dyn.pos_dim, oracle_dyn.node_nfs = 3, node_nfs
    dl = DiffusionLoss(oracle_dyn, "polynomial_2", T, 1e-5, norm_values=norm_values, pos_only=pos_only)
    reps64 = [dict(r) for r in batch]
    for r in reps64:
        r["pos"] = r["pos"].double()
    with torch.enable_grad():
        nll_o, _ = dl.compute_loss(reps64, cond.double(), training=True, t_int=t_rec.double(),
                                   draw=lambda shape: next(it).double())
        loss_o = nll_o.mean(0)
    worst, num, den = 0.0, 0.0, 0.0
    loss_o.backward()
    for k, gref in grads["f64"].items():
        go = sd64[k].grad
        if go is None:
            go = torch.zeros_like(gref)
        worst = max(worst, float((go - gref).abs().max() / gref.abs().max().clamp(min=1e-300)))
        num += float(((go - gref) ** 2).sum())
        den += float((gref ** 2).sum())
    glob = (num / den) ** 0.5
    # per tensor the float64 gradients themselves carry ~1e-7 of summation-order noise on a few tensors whose gradient
    # is the small remainder of a large cancelling sum (model.lin3.*: |grad| ~ 1e-6); the flat gradient agrees to ~1e-10
    print(name, "oracle-autograd vs reference-autograd (f64): worst per-tensor max|d|/max|ref| =", worst,
          "whole gradient |d|_2/|ref|_2 =", glob, "loss", float(loss_o.detach()), float(out["f64_loss"]))
    assert worst < 1e-6 and glob < 1e-8, (worst, glob)   # the literal node frame is noise at the 1e-10 level even in float64
    # reference float32 vs float64, per tensor (the reference's own noise floor on gradients)
    noise = {k: float((grads["f32"][k].double() - grads["f64"][k]).abs().max() / grads["f64"][k].abs().max().clamp(min=1e-300))
             for k in grads["f64"]}
    worst_noise = max(noise.values())
    print(name, "reference f32-vs-f64 gradient gap: worst per-tensor", worst_noise, "loss f32", float(out["f32_loss"]),
          "f64", float(out["f64_loss"]))

    for k, r in enumerate(batch):
        for f in ("size", "pos", "one_hot", "charge", "mask"):
            out[f"rep{k}_{f}"] = r[f].numpy()
    for i, x in enumerate(rec_randn):
        out[f"randn{i}"] = x.numpy()
    out["meta"] = np.array(json.dumps(dict(name=name, sizes=sizes, t_int=t_fixed, norm_values=list(norm_values), T=T,
                                            model_config=cfg, pos_only=pos_only, n_randn=len(rec_randn), oracle_vs_ref_f64=worst, oracle_vs_ref_f64_global=glob,
                                            ref_f32_vs_f64=noise)))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)


if __name__ == "__main__":
    if "--n23" in sys.argv:
        # the shape `bench.py --mode train` launches: 23-atom objects, all 6 layers, pos_only training (train_ts1x.py:99-108);
        # two reactions are what float64 autograd through the reference fits in the build box - the GPU test embeds them in
        # a B = 64 batch so that the gradients come out of the benched launch (tests/test_grad.py)
        run("g9_grad_prod_n23", [23, 23], [412, 57], (1.0, 4.0, 10.0), dict(PRODUCTION_LEFTNET_CONFIG), T=1000, pos_only=True)
        sys.exit(0)
    prod2 = dict(PRODUCTION_LEFTNET_CONFIG, num_layers=2)
    run("g9_grad_prod_l2", [5, 7, 4], [37, 0, 100], (1.0, 4.0, 10.0), prod2)
    run("g9_grad_prod_cutoff", [6, 3], [12, 70], (1.0, 4.0, 10.0), dict(PRODUCTION_LEFTNET_CONFIG, num_layers=3, cutoff=5.0),
        pos_scale=3.0)
    run("g9_grad_h32", [4, 6, 1, 3], [5, 50, 99, 1], (1.0, 4.0, 10.0),
        dict(PRODUCTION_LEFTNET_CONFIG, num_layers=2, hidden_channels=32, num_radial=8))
    # reflect_equiv = False: the adjoints of the message's x (x) coord_cross term and of the signed edge scalarisation
    run("g9_grad_h32_noreflect", [4, 6, 1, 3], [5, 50, 99, 1], (1.0, 4.0, 10.0),
        dict(PRODUCTION_LEFTNET_CONFIG, num_layers=2, hidden_channels=32, num_radial=8, reflect_equiv=False))
