"""oareactdiff_amd — MI355X (gfx950) backend for OA-ReactDiff's per-timestep denoising call.

    from oareactdiff_amd import EGNNDynamics        # drop-in for oa_reactdiff.dynamics.EGNNDynamics(model=LEFTNet)

See DESIGN.md (path, kernels, measurement) and INTEGRATION.md (binding)."""
from .dynamics import EGNNDynamics  # noqa: F401
from .sampler import DiffusionSampler  # noqa: F401
from .loss import DiffusionLoss  # noqa: F401
from .trainer import DDPMTrainer  # noqa: F401
from .graph_tools import get_edges_index, get_mask_for_frag, get_n_frag_switch, get_subgraph_mask  # noqa: F401

__all__ = ["EGNNDynamics", "DiffusionSampler", "DiffusionLoss", "DDPMTrainer", "get_edges_index", "get_mask_for_frag", "get_n_frag_switch", "get_subgraph_mask"]
