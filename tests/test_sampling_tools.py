"""Host-side I/O helpers (row N4) against outputs of the reference's own helpers (tests/golden/g7_io.npz,
made by oracle/make_goldens_io.py)."""
import json
import os

import numpy as np
import torch

from oareactdiff_amd.sampling_tools import assemble_sample_inputs, write_single_xyz, write_tmp_xyz

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_io.npz")


def test_assemble_sample_inputs_matches_reference():
    z = np.load(GOLDEN)
    meta = json.loads(str(z["meta"]))
    for ft in (False, True):
        h0 = assemble_sample_inputs(meta["atoms"], device=torch.device("cpu"), n_samples=2, frag_type=ft)
        assert len(h0) == 3
        for k, h in enumerate(h0):
            want = torch.from_numpy(z[f"h0_ft{int(ft)}_{k}"])
            assert h.dtype == want.dtype and torch.equal(h, want)


def test_xyz_files_match_reference(tmp_path):
    z = np.load(GOLDEN)
    meta = json.loads(str(z["meta"]))
    sizes = torch.tensor(meta["sizes"])
    samples = [torch.from_numpy(z[f"sample{k}"]) for k in range(3)]
    paths = write_tmp_xyz([sizes, sizes, sizes], samples, idx=[0, 1, 2], prefix="gen", localpath=str(tmp_path), ex_ind=3)
    assert sorted(os.path.basename(p) for p in paths) == sorted(meta["texts"])
    for p in paths:
        assert open(p).read() == meta["texts"][os.path.basename(p)]
    one = tmp_path / "one.xyz"
    write_single_xyz(str(one), 3, samples[0][:3])
    assert one.read_text() == meta["texts"]["gen_3_react.xyz"]
