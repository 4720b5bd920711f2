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
