"""Loss terms around the denoising call (row N2, forward half) against the reference's own
EnVariationalDiffusion.forward / compute_loss on recorded randomness (tests/golden/g8_loss_*.npz, made by
oracle/make_goldens_loss.py).

CPU: the loss arithmetic alone, with the reference's recorded network outputs replayed - float32 against the
reference's float32 run, float64 against its float64 run (tight).  GPU: the same with the HIP dynamics in the
loop, against the float64 reference."""
import json
import os

import numpy as np
import pytest
import torch

from oareactdiff_amd.loss import DiffusionLoss

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["g8_loss_train", "g8_loss_eval", "g8_loss_eval_posonly"]
TERMS = ("error_t", "loss_0_x", "loss_0_cat", "loss_0_charge")


def _load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z, json.loads(str(z["meta"]))


def _reps(z, dtype, dev="cpu"):
    reps = []
    for k in range(3):
        r = {f: torch.from_numpy(z[f"rep{k}_{f}"]).to(dev) for f in ("size", "pos", "one_hot", "charge", "mask")}
        r["pos"] = r["pos"].to(dtype)
        reps.append(r)
    return reps


def _draws(z, meta, dtype, dev="cpu"):
    it = iter(range(meta["n_randn"]))

    def draw(shape):
        x = torch.from_numpy(z[f"randn{next(it)}"]).to(device=dev, dtype=dtype)
        assert tuple(x.shape) == tuple(shape)
        return x
    return draw


def _close(a, b, rtol, atol):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), (a, b)


def _check(lt, nll, z, tag, rtol, atol):
    for key in TERMS:
        for k in range(3):
            _close(lt[key][k], z[f"{tag}_{key}{k}"], rtol, atol)
    for key in ("SNR_weight", "neg_log_constants", "kl_prior", "t_int"):
        _close(lt[key], z[f"{tag}_{key}"], rtol, atol)
    _close(float(lt["delta_log_px"]), float(z[f"{tag}_delta_log_px"]), rtol, atol)
    _close(nll, z[f"{tag}_nll"], rtol, atol)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tag,dtype,rtol,atol", [("f32", torch.float32, 2e-5, 2e-4), ("f64", torch.float64, 1e-11, 1e-11)])
def test_loss_arithmetic_matches_reference(name, tag, dtype, rtol, atol):
    z, meta = _load(name)
    calls = iter(range(meta["n_net_calls"]))
    prefix = "net_call" if tag == "f32" else "net64_call"

    class Replay:                      # the reference's own network outputs, so only the loss arithmetic is under test
        pos_dim, node_nfs = 3, [9, 9, 9]

        def __call__(self, **kw):
            c = next(calls)
            return [torch.from_numpy(z[f"{prefix}{c}_{k}"]).to(dtype) for k in range(3)], None

    loss = DiffusionLoss(Replay(), "polynomial_2", meta["T"], 1e-5, norm_values=meta["norm_values"], pos_only=meta["pos_only"])
    t_int = torch.tensor(meta["t_int"], dtype=dtype).view(-1, 1)
    cond = torch.zeros(len(meta["sizes"]), 1, dtype=dtype)
    for training_api in ("terms", "loss"):
        calls = iter(range(meta["n_net_calls"]))
        if training_api == "terms":
            lt = loss.loss_terms(_reps(z, dtype), cond, training=meta["training"], t_int=t_int, draw=_draws(z, meta, dtype))
        else:
            nll, info = loss.compute_loss(_reps(z, dtype), cond, training=meta["training"], t_int=t_int, draw=_draws(z, meta, dtype))
    _check(lt, nll, z, tag, rtol, atol)
    assert set(info) == {f"{p}_{k}" for p in ("error_t", "unorm_error_t") for k in range(3)}


def test_loss_draws_its_own_randomness_and_does_not_mutate_the_batch():
    z, meta = _load("g8_loss_eval")

    class Zero:
        pos_dim, node_nfs = 3, [9, 9, 9]

        def __call__(self, xh, **kw):
            return [torch.zeros_like(x) for x in xh], None

    reps = _reps(z, torch.float32)
    keep = [{k: v.clone() for k, v in r.items()} for r in reps]
    loss = DiffusionLoss(Zero(), "polynomial_2", meta["T"], 1e-5, norm_values=meta["norm_values"])
    torch.manual_seed(0)
    nll, _ = loss.compute_loss(reps, torch.zeros(2, 1), training=False)
    assert nll.shape == (2,) and bool(torch.isfinite(nll).all())
    for r, k in zip(reps, keep):
        for f in r:
            assert torch.equal(r[f], k[f])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_loss_with_hip_dynamics_matches_reference_f64(name):
    from oareactdiff_amd.dynamics import EGNNDynamics
    from oareactdiff_amd.spec import state_spec, synthetic_state_dict
    z, meta = _load(name)
    dev = torch.device("cuda:0")
    cfg = dict(meta["model_config"])
    dyn = EGNNDynamics(model_config=dict(cfg), fragment_names=["R", "TS", "P"], node_nfs=[9, 9, 9], edge_nf=0,
                       condition_nf=1, device=dev)
    dyn.load_state_dict(synthetic_state_dict(state_spec(cfg, [9, 9, 9], 1), cfg, seed=42), strict=True)
    loss = DiffusionLoss(dyn, "polynomial_2", meta["T"], 1e-5, norm_values=meta["norm_values"], pos_only=meta["pos_only"])
    t_int = torch.tensor(meta["t_int"], dtype=torch.float32, device=dev).view(-1, 1)
    cond = torch.zeros(len(meta["sizes"]), 1, device=dev)
    reps = _reps(z, torch.float32, dev)
    lt = loss.loss_terms(reps, cond, training=meta["training"], t_int=t_int, draw=_draws(z, meta, torch.float32, dev))
    nll, _ = loss.compute_loss(reps, cond, training=meta["training"], t_int=t_int, draw=_draws(z, meta, torch.float32, dev))
    # network outputs: the usual gate; loss terms: relative to the largest entry of each term
    for k in range(3):
        ref = torch.from_numpy(z[f"f64_net{k}"])
        assert float((lt["net_eps_xh"][k].cpu().double() - ref).abs().max() / ref.abs().max()) <= 1e-5
    for key in TERMS:
        for k in range(3):
            ref = torch.from_numpy(z[f"f64_{key}{k}"])
            got = lt[key][k].cpu().double()
            assert float((got - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1.0), (key, k, got, ref)
    ref = torch.from_numpy(z["f64_nll"])
    assert float(((nll.cpu().double() - ref).abs() / ref.abs().clamp(min=1.0)).max()) <= 2e-5, (nll, ref)
