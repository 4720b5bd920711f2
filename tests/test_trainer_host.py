"""Host-side pieces of the round-4 training step that need no GPU: the lazy `info` mapping of the step without a host sync, the batch
mover that keeps the host copies of mask / size (the layout of a never-seen batch then costs no device -> host copy), the layout
cache without an edge list, and the refusal of `host_sync=False` where the fused step does not apply."""
from collections.abc import Mapping

import pytest
import torch

from test_trainer_gloo import _oracle_dynamics


def test_lazy_info_reads_like_the_eager_dict():
    from oareactdiff_amd.trainer import LazyInfo
    K, scales = 3, [1.0, 2.0, 1.0]
    stats = torch.tensor([0.25, 0.0, 1.5] + [0.1, 0.2, 0.3] + [1.0, 2.0, 3.0])       # norm, flag, loss, err_n[k], err_t[k]
    out4 = torch.tensor([0.25, 4500.0, 1.0, 0.0])                                     # grad_norm, max_norm, gscale, skipped
    info = LazyInfo(stats, out4, K, scales, clip=True)
    assert isinstance(info, Mapping) and info._d is None                              # nothing fetched yet
    assert info["loss"] == 1.5 and info["skipped"] == 0 and info["grad_norm"] == 0.25 and info["max_grad_norm"] == 4500.0
    assert info["error_t_1"] == pytest.approx(0.2 / (2.0 + 1e-4)) and info["unorm_error_t_2"] == 3.0
    assert set(info) == {"loss", "skipped", "grad_norm", "max_grad_norm"} | {f"error_t_{k}" for k in range(K)} | {f"unorm_error_t_{k}" for k in range(K)}
    assert dict(info)["loss"] == 1.5 and len(info) == 10 and info.get("nope") is None
    skipped = LazyInfo(stats, torch.tensor([float("nan"), float("nan"), 1.0, 1.0]), K, scales, clip=False)
    assert skipped["skipped"] == 1 and "grad_norm" not in skipped


def test_to_device_keeps_the_host_copies_of_the_layout():
    from oareactdiff_amd.trainer import DDPMTrainer
    size = torch.tensor([3, 2])
    mask = torch.repeat_interleave(torch.arange(2), size)
    reps = [{"size": size.clone(), "pos": torch.randn(5, 3), "one_hot": torch.zeros(5, 5, dtype=torch.long),
             "charge": torch.ones(5, 1, dtype=torch.long), "mask": mask.clone()} for _ in range(3)]
    moved, cond = DDPMTrainer.to_device((reps, torch.zeros(2, 1)), "cpu")
    for r, m in zip(reps, moved):
        assert torch.equal(m["mask_host"], r["mask"]) and torch.equal(m["size_host"], r["size"]) and m["mask_host"].device.type == "cpu"
        assert torch.equal(m["pos"], r["pos"]) and set(m) == set(r) | {"mask_host", "size_host"}
    assert cond.shape == (2, 1)


def test_layout_without_an_edge_list_is_cached_separately():
    from oareactdiff_amd.loss import DiffusionLoss
    dyn = _oracle_dynamics()
    ls = DiffusionLoss(dyn, timesteps=50)
    size = torch.tensor([3, 2])
    masks = [torch.repeat_interleave(torch.arange(2), size) for _ in range(3)]
    sizes = [size.clone() for _ in range(3)]
    cm, ei, nfs = ls._layout(masks, sizes, need_edges=False)
    assert ei is None and cm.numel() == 15 and nfs.numel() == 15
    cm2, ei2, _ = ls._layout(masks, sizes)                       # the same tensors with edges: its own entry
    assert ei2 is not None and ei2.shape == (2, 9 * 8 + 6 * 5) and torch.equal(cm, cm2)
    assert ls._layout(masks, sizes, need_edges=False)[0] is cm   # hit


def test_host_sync_false_is_a_mode_of_the_fused_step_only():
    from oareactdiff_amd.trainer import DDPMTrainer
    with pytest.raises(ValueError, match="host_sync=False"):
        DDPMTrainer(_oracle_dynamics(), timesteps=50, host_sync=False)
