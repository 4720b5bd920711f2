"""Replay of one recorded training step (tests/golden/g9_grad_*.npz, made by oracle/make_goldens_grad.py from the
reference) and comparison of a {name: gradient} dict with the compact fixture."""
import json
import os

import numpy as np
import torch

from oareactdiff_amd.loss import DiffusionLoss
from oareactdiff_amd.spec import state_spec, synthetic_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRAD_CASES = ["g9_grad_h32", "g9_grad_prod_l2", "g9_grad_prod_cutoff", "g9_grad_prod_n23", "g9_grad_h32_noreflect"]
#: g9_grad_prod_n23: two 23-atom reactions, production dims, all 6 layers, pos_only loss (oracle/make_goldens_grad.py --n23)
NODE_NFS, CNF = [9, 9, 9], 1


class GradCase:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(str(self.z["meta"]))
        self.cfg = dict(self.meta["model_config"])
        self.names = [k[len("gnorm."):] for k in self.z.files if k.startswith("gnorm.")]

    def state_dict(self, dtype=torch.float32):
        return synthetic_state_dict(state_spec(self.cfg, NODE_NFS, CNF), self.cfg, seed=42, dtype=dtype)

    def reps(self, dtype, dev="cpu"):
        out = []
        for k in range(3):
            r = {f: torch.from_numpy(self.z[f"rep{k}_{f}"]).to(dev) for f in ("size", "pos", "one_hot", "charge", "mask")}
            r["pos"] = r["pos"].to(dtype)
            out.append(r)
        return out

    def loss(self, dynamics, dtype, dev="cpu"):
        """nll.mean(0) of DDPMModule.training_step (pl_trainer.py:327-329) on the recorded t_int and noise."""
        it = iter(range(self.meta["n_randn"]))
        dl = DiffusionLoss(dynamics, "polynomial_2", self.meta["T"], 1e-5, norm_values=self.meta["norm_values"], node_nfs=NODE_NFS,
                           pos_only=self.meta.get("pos_only", False))
        t_int = torch.tensor(self.meta["t_int"], dtype=dtype, device=dev).view(-1, 1)
        cond = torch.zeros(len(self.meta["sizes"]), 1, dtype=dtype, device=dev)
        nll, _ = dl.compute_loss(self.reps(dtype, dev), cond, training=True, t_int=t_int,
                                 draw=lambda shape: torch.from_numpy(self.z[f"randn{next(it)}"]).to(device=dev, dtype=dtype))
        return nll.mean(0)

    def embedded(self, B, slots, dtype, dev, seed=5):
        """The recorded step embedded in a batch of B equally sized reactions: reaction j of the fixture sits in slot
        slots[j], the other slots hold random reactions / time steps / noise.  -> (reps, cond, t_int, draw) for
        DiffusionLoss.compute_loss; per-sample terms of the occupied slots must equal the fixture's."""
        sizes = self.meta["sizes"]
        nf = sizes[0]
        assert all(s == nf for s in sizes) and len(slots) == len(sizes)
        g = torch.Generator().manual_seed(seed)
        n = B * nf
        mask = torch.repeat_interleave(torch.arange(B), nf)
        rows = torch.cat([torch.arange(s * nf, (s + 1) * nf) for s in slots])
        reps = []
        for k in range(3):
            pos = torch.randn(n, 3, generator=g)
            pos = pos - (torch.zeros(B, 3).index_add_(0, mask, pos) / nf)[mask]
            typ = torch.randint(0, 4, (n,), generator=g)
            one_hot = torch.zeros(n, 5, dtype=torch.long)
            one_hot[torch.arange(n), typ] = 1
            charge = torch.tensor([1, 6, 7, 8])[typ].view(n, 1)
            pos[rows] = torch.from_numpy(self.z[f"rep{k}_pos"])
            one_hot[rows] = torch.from_numpy(self.z[f"rep{k}_one_hot"])
            charge[rows] = torch.from_numpy(self.z[f"rep{k}_charge"])
            reps.append({"size": torch.full((B,), nf, dtype=torch.long).to(dev), "pos": pos.to(dev, dtype), "one_hot": one_hot.to(dev),
                         "charge": charge.to(dev), "mask": mask.to(dev)})
        t_int = torch.randint(0, self.meta["T"] + 1, (B, 1), generator=g).to(dtype)
        t_int[torch.tensor(slots), 0] = torch.tensor(self.meta["t_int"], dtype=dtype)
        draws = []
        for i in range(self.meta["n_randn"]):
            rec = torch.from_numpy(self.z[f"randn{i}"])
            x = torch.randn(n, rec.shape[1], generator=g)
            x[rows] = rec
            draws.append(x.to(dev, dtype))
        it = iter(draws)
        return reps, torch.zeros(B, 1, dtype=dtype, device=dev), t_int.to(dev), (lambda shape: next(it))

    def compare(self, grads):
        """-> ({name: error}, flat error).  Per tensor: max deviation of the entries (all of them, or the sampled ones)
        relative to the tensor's largest reference entry, and of its row / column sums relative to the largest sum;
        flat: L2 over every stored number relative to the reference's L2."""
        z, errs, num, den = self.z, {}, 0.0, 0.0
        for n in self.names:
            gmax = float(z["gnorm." + n][1])
            g = grads.get(n)
            g = torch.zeros(1) if g is None else g.detach().double().cpu()
            if "gfull." + n in z.files:
                ref = torch.from_numpy(z["gfull." + n])
                g = g.reshape(ref.shape) if g.numel() == ref.numel() else torch.zeros_like(ref)
                d = (g - ref)
                e = float(d.abs().max()) / max(gmax, 1e-300)
            else:
                rows, cols = torch.from_numpy(z["grow." + n]), torch.from_numpy(z["gcol." + n])
                g2 = g.reshape(rows.numel(), -1) if g.numel() == rows.numel() * cols.numel() else torch.zeros(rows.numel(), cols.numel(), dtype=torch.float64)
                ref = torch.from_numpy(z["gval." + n])
                d = g2.reshape(-1)[torch.from_numpy(z["gidx." + n])] - ref
                # sums are measured against the larger of the largest sum and gmax * sqrt(terms): behind a LayerNorm
                # without affine part, for instance, every column sum is exactly zero in exact arithmetic
                rs = max(float(rows.abs().max()), gmax * cols.numel() ** 0.5)
                cs = max(float(cols.abs().max()), gmax * rows.numel() ** 0.5)
                e = max(float(d.abs().max()) / max(gmax, 1e-300), float((g2.sum(1) - rows).abs().max()) / rs,
                        float((g2.sum(0) - cols).abs().max()) / cs)
            errs[n] = e
            num += float((d ** 2).sum())
            den += float((ref ** 2).sum())
        return errs, (num / max(den, 1e-300)) ** 0.5
