"""Checkpoint adapter: load the hot-path weights of a reference
Lightning checkpoint (`DDPMModule`, oa_reactdiff/trainer/pl_trainer.py:55-147; e.g.
`pretrained-ts1x-diff.ckpt`) into the MI355X `EGNNDynamics`.

The Lightning `state_dict` stores the dynamics under the prefix `ddpm.dynamics.`; the constructor
arguments are in `hyper_parameters` (`save_hyperparameters()`, pl_trainer.py:147)."""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Tuple

import torch

from .dynamics import EGNNDynamics

PREFIXES = ("ddpm.dynamics.", "dynamics.", "")


def extract_dynamics_state(state_dict: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Tensors of the dynamics module, prefix stripped (`model.*`, `encoders.*`, `decoders.*`)."""
    for pre in PREFIXES:
        sub = {k[len(pre):]: v for k, v in state_dict.items()
               if k.startswith(pre) and k[len(pre):].split(".")[0] in ("model", "encoders", "decoders")}
        if sub:
            return sub
    raise KeyError("no dynamics tensors (model.* / encoders.* / decoders.*) found in the state dict")


def dynamics_from_checkpoint(ckpt: Mapping, device: torch.device = torch.device("cuda"),
                             overrides: Optional[dict] = None) -> Tuple[EGNNDynamics, dict]:
    """Builds `EGNNDynamics` from a loaded Lightning checkpoint dict (`torch.load(path, map_location="cpu")`)
    and loads its weights with strict=True.  Returns (dynamics, hyper_parameters)."""
    hp = dict(ckpt.get("hyper_parameters", {}))
    if overrides:
        hp.update(overrides)
    try:
        model_config, node_nfs = dict(hp["model_config"]), list(hp["node_nfs"])
    except KeyError as e:
        raise KeyError(f"checkpoint hyper_parameters lack {e}; pass them through `overrides`") from None
    dyn = EGNNDynamics(
        model_config=model_config,
        fragment_names=list(hp.get("fragment_names", [f"frag{k}" for k in range(len(node_nfs))])),
        node_nfs=node_nfs, edge_nf=int(hp.get("edge_nf", 0)), condition_nf=int(hp.get("condition_nf", 0)),
        pos_dim=int(hp.get("pos_dim", 3)), update_pocket_coords=bool(hp.get("update_pocket_coords", True)),
        condition_time=bool(hp.get("condition_time", True)), edge_cutoff=hp.get("edge_cutoff"),
        device=device, enforce_same_encoding=hp.get("enforce_same_encoding"),
    )
    dyn.load_state_dict(extract_dynamics_state(ckpt["state_dict"]), strict=True)
    return dyn, hp
