"""Host-side helpers either side of the sampler (row N4 of DESIGN.md): feature assembly for a list of
element symbols and the xyz writer for generated R/TS/P triples.

Same names, arguments and file format as the reference's `oa_reactdiff/utils/sampling_tools.py:64-149`
(`assemble_sample_inputs`, `write_single_xyz`, `write_tmp_xyz`); each object's rows cross the PCIe bus once
instead of once per atom.
"""
import os
from typing import List, Sequence

import torch

_ELEMENTS = ("H", "C", "N", "O", "F")
_CHARGE = {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9}
_SYMBOL = {z: s for s, z in _CHARGE.items()}
_OBJECT_TAG = {0: "react", 1: "ts", 2: "prod"}


def _feature_row(symbol: str, extra: Sequence[int] = ()) -> List[int]:
    """5-way one-hot over (H, C, N, O, F) followed by the nuclear charge (and an optional fragment-type flag)."""
    row = [int(symbol == e) for e in _ELEMENTS]
    row.append(_CHARGE[symbol])
    row.extend(extra)
    return row


def assemble_sample_inputs(atoms: List[str], device: torch.device = torch.device("cuda"), n_samples: int = 1,
                           frag_type: bool = False) -> List[torch.Tensor]:
    """h0 for `DiffusionSampler.sample(..., h0=...)`: three int64 tensors [n_samples * len(atoms), 6 (7)].

    Reference: sampling_tools.py:64-108.  With `frag_type` the objects alternate a trailing 0/1 flag
    (R: 0, TS: 1, P: 0)."""
    out = []
    for k in range(3):
        extra = (k % 2,) if frag_type else ()
        block = torch.tensor([_feature_row(a, extra) for a in atoms], dtype=torch.int64)
        out.append(block.repeat(n_samples, 1).to(device))
    return out


def write_single_xyz(xyzfile: str, natoms: int, out: torch.Tensor) -> None:
    """One molecule: rows of `out` are [x, y, z, one-hot(5), charge, ...]; element from column 8
    (sampling_tools.py:111-127)."""
    rows = out[:, :9].detach().cpu()
    xyz = rows[:, :3].numpy()
    charge = rows[:, 8].long().tolist()
    with open(xyzfile, "w") as fo:
        fo.write(f"{natoms}\n\n")
        for z, p in zip(charge, xyz):
            fo.write(_SYMBOL[z] + " " + " ".join(str(v) for v in p) + "\n")


def write_tmp_xyz(fragments_nodes, out_samples, idx=(0,), prefix: str = "gen", localpath: str = "tmp",
                  ex_ind: int = 0) -> List[str]:
    """`{localpath}/{prefix}_{sample + ex_ind}_{react|ts|prod}.xyz` for every sample of the objects in `idx`
    (sampling_tools.py:130-149).  Returns the paths written."""
    sizes = [int(n) for n in torch.as_tensor(fragments_nodes[0]).tolist()]
    paths = []
    for k in idx:
        rows = out_samples[k].detach().cpu()
        start = 0
        for j, n in enumerate(sizes):
            path = os.path.join(localpath, f"{prefix}_{j + ex_ind}_{_OBJECT_TAG[k]}.xyz")
            write_single_xyz(path, n, rows[start:start + n])
            paths.append(path)
            start += n
    return paths
