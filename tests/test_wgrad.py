"""oard_wgrad through the C ABI on shapes the training step does not exercise: ragged row counts (fewer rows than one
prefetch step, partial last steps, several chunks), partial 64-column blocks of the wide operand, 16-column tiles beyond
the narrow operand, section padding, SiLU-on-load, and the small-output path.  Oracle: the same product in float64
(nn.Linear backward, dW = dY^T act(X), db = column sums of dY)."""
import pytest
import torch

from oareactdiff_amd import training

pytestmark = pytest.mark.gpu


class _Owner:
    pass


def _case(rows, ncY, o_len, o_pad, MO, ncX, i_len, i_pad, MI, x_silu, want_bias, seed, ld_extra=0):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    dY = torch.randn(rows + 1, ncY + ld_extra, generator=g).to(dev)          # + a spare row, wider leading dimension
    X = torch.randn(rows + 1, ncX + ld_extra, generator=g).to(dev)
    dW, db = training._wgrad(dY, ncY, o_len, o_pad, MO, X, ncX, x_silu, i_len, i_pad, MI, rows, want_bias, _Owner(),
                             torch.cuda.current_stream().cuda_stream)
    op = torch.tensor([(o // o_len) * o_pad + o % o_len for o in range(MO)])
    ip = torch.tensor([(i // i_len) * i_pad + i % i_len for i in range(MI)])
    y64, x64 = dY[:rows].double().cpu()[:, op], X[:rows].double().cpu()[:, ip]
    if x_silu:
        x64 = x64 * torch.sigmoid(x64)
    ref_w, ref_b = y64.t() @ x64, y64.sum(dim=0)
    scale = max(float(ref_w.abs().max()), 1e-30)
    assert float((dW.double().cpu() - ref_w).abs().max()) <= 2e-6 * max(scale, rows ** 0.5), (rows, ncY, ncX)
    if want_bias:
        assert float((db.double().cpu() - ref_b).abs().max()) <= 2e-6 * max(float(ref_b.abs().max()), rows ** 0.5)


@pytest.mark.parametrize("rows", [1, 3, 4, 7, 8, 9, 15, 16, 17, 31, 1000, 4099])
def test_wgrad_ragged_rows(rows):
    # wide operand dY (688 columns: 10 full 64-blocks + one of 48), narrow X (208 = 13 tiles: 7 + 6 of a 7-tile group)
    _case(rows, 688, 684, 684, 684, 208, 196, 196, 196, False, True, seed=rows)
    # transposed product (X wider than dY), no bias
    _case(rows, 208, 196, 196, 196, 688, 684, 684, 684, False, False, seed=100 + rows)


def test_wgrad_sections_silu_and_leading_dimension():
    # three 196-wide thirds stored 208 apart on the dY side, SiLU on load of a 592-column X (37 tiles), ld > nc
    _case(513, 624, 196, 208, 588, 592, 588, 588, 588, True, True, seed=7, ld_extra=8)
    # radial projection: 96-column X
    _case(777, 624, 196, 208, 588, 96, 96, 96, 96, False, False, seed=8)


@pytest.mark.parametrize("rows", [1, 63, 64, 65, 5000])
def test_wgrad_small_outputs(rows):
    # the frame-scalar MLP layers: 48 x 4 (weight | bias column), 8 x 48, 1 x 12
    _case(rows, 48, 48, 48, 48, 4, 4, 4, 1, False, True, seed=rows)
    _case(rows, 8, 8, 8, 8, 48, 48, 48, 48, False, True, seed=rows + 1)


@pytest.mark.parametrize("rows", [16384, 16385, 20011, 40000])
def test_wgrad_lds_panel_kernel(rows):
    """rows >= 16 384 with at least two 16-tile groups on the narrow side run k_wgrad_lds (row panels of 32 rows shared through
    LDS by the eight (P block, Q group) tasks of a workgroup): ragged last groups / chunks, partial 64-column blocks, partial
    tile groups (37 tiles = 4 full groups of 8 + 5), section padding, SiLU applied when the panel is staged, column sums."""
    _case(rows, 688, 684, 684, 684, 208, 196, 196, 196, False, True, seed=rows)                    # edge_out_trans: dz3 x m
    _case(rows, 208, 196, 196, 196, 688, 684, 684, 684, False, False, seed=rows + 1)               # edge_mlp.0 (transposed product)
    _case(rows, 624, 196, 208, 588, 592, 588, 588, 588, True, True, seed=rows + 2, ld_extra=8)     # dir_proj.2: SiLU on load, thirds
    _case(rows, 592, 588, 588, 588, 688, 684, 684, 684, False, True, seed=rows + 3)                # dir_proj.0 (transposed, bias on the Q side)
    _case(rows, 208, 196, 196, 196, 208, 196, 196, 196, True, True, seed=rows + 4)                 # edge_mlp.1


def test_wgrad_lds_and_per_wave_kernels_agree():
    """A/B of the two kernels on one shape through the debug option (deterministic each, equal up to summation order)."""
    from oareactdiff_amd import _capi
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    rows = 30000
    dY, X = torch.randn(rows, 688, generator=g).to(dev), torch.randn(rows, 208, generator=g).to(dev)
    out = {}
    for mode in (256, 0):
        assert _capi.lib().oard_debug_option(b"wgrad_lds", mode) == 0
        out[mode] = training._wgrad(dY, 688, 684, 684, 684, X, 208, False, 196, 196, 196, rows, True, _Owner(), torch.cuda.current_stream().cuda_stream)
        again = training._wgrad(dY, 688, 684, 684, 684, X, 208, False, 196, 196, 196, rows, True, _Owner(), torch.cuda.current_stream().cuda_stream)
        assert torch.equal(out[mode][0], again[0]) and torch.equal(out[mode][1], again[1])       # run-to-run deterministic
    _capi.lib().oard_debug_option(b"wgrad_lds", 256)
    assert float((out[256][0] - out[0][0]).abs().max()) <= 2e-6 * float(out[0][0].abs().max())
    assert float((out[256][1] - out[0][1]).abs().max()) <= 2e-6 * float(out[0][1].abs().max())
