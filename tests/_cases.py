"""Shared helpers for the parity tests: load a golden fixture, rebuild its weights."""
import contextlib
import json
import os

import numpy as np
import torch

from oareactdiff_amd.spec import state_spec, synthetic_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL_CASES = ["g1_wrapper_small", "g2_prod_b2_n23", "g2s_prod_b1_n5", "g3_cutoff_ragged",
             "g3p_prod_cutoff", "g6_h32_r32",
             "g10_noreflect_h32", "g10p_noreflect_prod"]      # reflect_equiv = False (leftnet.py:268-272, 794-796)


def rel(a, b):
    """max|a-b| / max|b| (the metric of SURVEY.md section 8c)."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    if b.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-300))


class Case:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.z = z
        self.name = name
        self.meta = json.loads(str(z["meta"]))
        self.cfg = dict(self.meta["model_config"])
        self.node_nfs = self.meta["node_nfs"]
        self.cnf = self.meta["condition_nf"]
        self.n_obj = len(self.node_nfs)
        self.xh = [torch.from_numpy(z[f"xh{k}"]) for k in range(self.n_obj)]
        self.edge_index = torch.from_numpy(z["edge_index"])
        self.t = torch.from_numpy(z["t"])
        self.conditions = torch.from_numpy(z["conditions"])
        self.n_frag_switch = torch.from_numpy(z["n_frag_switch"])
        self.combined_mask = torch.from_numpy(z["combined_mask"])
        self.ref64 = [torch.from_numpy(z[f"ref64_out{k}"]) for k in range(self.n_obj)]
        self.ref32 = [torch.from_numpy(z[f"ref32_out{k}"]) for k in range(self.n_obj)]
        self.spec = state_spec(self.cfg, self.node_nfs, self.cnf)

    def state_dict(self, dtype=torch.float32):
        return synthetic_state_dict(self.spec, self.cfg, seed=42, dtype=dtype)

    def split(self, outs, pos_dim=3):
        """-> (all velocities flattened, all decoded features flattened)"""
        nz = [k for k in range(self.n_obj) if self.xh[k].size(0)]
        vel = torch.cat([torch.as_tensor(outs[k])[:, :pos_dim].double().reshape(-1) for k in nz])
        h = torch.cat([torch.as_tensor(outs[k])[:, pos_dim:].double().reshape(-1) for k in nz])
        return vel, h


LIB_AUTO = dict(auto_small=4, auto_tiny=8, npb=0)     # the library's default launch-shape heuristics (conftest turns them off)
# the throughput launch shapes, pinned: tests of properties that only those kernels have (split-precision forms, bit-exact skipping of
# masked edges - the row-lane gathers of small launches sum in an order that depends on the list position) set them explicitly, so that
# they also hold when the whole suite runs under OARD_TEST_SHAPES=auto (tools/soak_tests.sh)
THROUGHPUT = dict(auto_small=0, auto_tiny=0, npb=16)


@contextlib.contextmanager
def debug_options(**kw):
    """Set oard_debug_option values for the duration of a block, then restore the suite's options."""
    from conftest import SUITE_OPTIONS
    from oareactdiff_amd import _capi
    lib = _capi.lib()
    try:
        for k, v in kw.items():
            assert lib.oard_debug_option(k.encode(), v) == 0, k
        yield lib
    finally:
        for k in kw:
            lib.oard_debug_option(k.encode(), SUITE_OPTIONS[k])
