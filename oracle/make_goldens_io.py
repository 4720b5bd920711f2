"""Generates tests/golden/g7_io.npz: outputs of the REFERENCE's sampling_tools helpers (imported from
/root/reference in the build container) on seeded inputs — the expected h0 tensors and xyz file texts.
Test infrastructure only.  Run:  python oracle/make_goldens_io.py"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_stubs"))
sys.path.insert(0, "/root/reference")
from oa_reactdiff.utils import sampling_tools as ref  # noqa: E402


def main():
    g = torch.Generator().manual_seed(5)
    atoms = ["C", "H", "H", "O", "N", "F", "H"]
    out = {}
    for ft in (False, True):
        h0 = ref.assemble_sample_inputs(atoms, device=torch.device("cpu"), n_samples=2, frag_type=ft)
        for k, h in enumerate(h0):
            out[f"h0_ft{int(ft)}_{k}"] = h.numpy()
    sizes = torch.tensor([3, 4])
    samples = []
    for k in range(3):
        pos = torch.randn(7, 3, generator=g)
        typ = torch.randint(0, 5, (7,), generator=g)
        feat = torch.zeros(7, 6)
        feat[torch.arange(7), typ] = 1.0
        feat[:, 5] = torch.tensor([1.0, 6.0, 7.0, 8.0, 9.0])[typ]
        samples.append(torch.cat([pos, feat], 1))
        out[f"sample{k}"] = samples[-1].numpy()
    texts = {}
    with tempfile.TemporaryDirectory() as d:
        ref.write_tmp_xyz([sizes, sizes, sizes], samples, idx=[0, 1, 2], prefix="gen", localpath=d, ex_ind=3)
        for fn in sorted(os.listdir(d)):
            texts[fn] = open(os.path.join(d, fn)).read()
    out["meta"] = json.dumps({"atoms": atoms, "sizes": sizes.tolist(), "texts": texts})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g7_io.npz"), **out)
    print("wrote g7_io.npz:", sorted(texts))


if __name__ == "__main__":
    main()
