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


@pytest.fixture(params=["t16", "lds"])
def long_kernel(request):
    """Which kernel serves the long contractions: "t16" = k_wgrad_t16 (round 4, the default: 16 x 16 tiles on both operands, rows
    streamed by LDS-DMA into a three-buffer ring, SiLU applied in LDS); "lds" = round 3's k_wgrad_lds / k_wgrad (debug option
    wgrad_t16 = 0), kept as the cross-check and for shapes the tile kernel has no instantiation for."""
    from oareactdiff_amd import _capi
    assert _capi.lib().oard_debug_option(b"wgrad_t16", 256 if request.param == "t16" else 0) == 0
    yield request.param
    _capi.lib().oard_debug_option(b"wgrad_t16", 256)


@pytest.mark.parametrize("rows", [16384, 16385, 20011, 40000])
def test_wgrad_long_contractions(rows, long_kernel):
    """rows >= 16 384 on the five shapes of a layer: ragged last 16-row groups / chunks (rows beyond the chunk are zero-filled in LDS),
    workgroup tiles of 22 + 21 and 13 + 12 + 12 tiles, section padding (three 196-wide thirds stored 208 apart), SiLU applied to the
    staged panel, column sums on the P side and on the Q side, a leading dimension wider than the operand."""
    _case(rows, 688, 684, 684, 684, 208, 196, 196, 196, False, True, seed=rows)                    # edge_out_trans: dz3 x m
    _case(rows, 208, 196, 196, 196, 688, 684, 684, 684, False, False, seed=rows + 1)               # edge_mlp.0 (transposed product)
    _case(rows, 624, 196, 208, 588, 592, 588, 588, 588, True, True, seed=rows + 2, ld_extra=8)     # dir_proj.2: SiLU on load, thirds
    _case(rows, 592, 588, 588, 588, 688, 684, 684, 684, False, True, seed=rows + 3)                # dir_proj.0 (transposed, bias on the Q side)
    _case(rows, 208, 196, 196, 196, 208, 196, 196, 196, True, True, seed=rows + 4)                 # edge_mlp.1
    _case(rows, 624, 196, 208, 588, 96, 96, 96, 96, False, False, seed=rows + 5)                   # rbf_proj: 96-column X, no bias


def test_wgrad_three_kernels_agree():
    """A/B of the three kernels on one shape through the debug options (deterministic each, equal up to summation order)."""
    from oareactdiff_amd import _capi
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    rows = 30000
    dY, X = torch.randn(rows, 688, generator=g).to(dev), torch.randn(rows, 208, generator=g).to(dev)
    out = {}
    try:
        for name, t16, lds in (("t16", 256, 256), ("lds", 0, 256), ("wave", 0, 0)):
            assert _capi.lib().oard_debug_option(b"wgrad_t16", t16) == 0 and _capi.lib().oard_debug_option(b"wgrad_lds", lds) == 0
            out[name] = training._wgrad(dY, 688, 684, 684, 684, X, 208, False, 196, 196, 196, rows, True, _Owner(), torch.cuda.current_stream().cuda_stream)
            again = training._wgrad(dY, 688, 684, 684, 684, X, 208, False, 196, 196, 196, rows, True, _Owner(), torch.cuda.current_stream().cuda_stream)
            assert torch.equal(out[name][0], again[0]) and torch.equal(out[name][1], again[1])       # run-to-run deterministic
    finally:
        _capi.lib().oard_debug_option(b"wgrad_t16", 256)
        _capi.lib().oard_debug_option(b"wgrad_lds", 256)
    for name in ("t16", "lds"):
        assert float((out[name][0] - out["wave"][0]).abs().max()) <= 2e-6 * float(out["wave"][0].abs().max())
        assert float((out[name][1] - out["wave"][1]).abs().max()) <= 2e-6 * float(out["wave"][1].abs().max())
    assert not torch.equal(out["t16"][0], out["lds"][0])                 # really different kernels (summation order)


def test_wgrad_tile_kernel_speed_report():
    """Not a gate: prints the time of the five long products of one layer at B = 64 under both kernels."""
    from oareactdiff_amd import _capi
    dev = torch.device("cuda:0")
    E, A = 300288, 97152
    shapes = [("edge_out_trans", E, 688, 684, 684, 684, 208, 196, 196, 196, False, True),
              ("edge_mlp.1", E, 208, 196, 196, 196, 208, 196, 196, 196, True, True),
              ("edge_mlp.0", E, 208, 196, 196, 196, 688, 684, 684, 684, False, False),
              ("dir_proj.2", A, 624, 196, 208, 588, 592, 588, 588, 588, True, True),
              ("dir_proj.0", A, 592, 588, 588, 588, 688, 684, 684, 684, False, True)]
    st = torch.cuda.current_stream().cuda_stream
    try:
        for name, rows, ncY, ol, op, MO, ncX, il, ip, MI, silu, bias in shapes:
            dY, X = torch.randn(rows, ncY, device=dev), torch.randn(rows, ncX, device=dev)
            line = []
            for kern, t16 in (("t16", 256), ("lds", 0)):
                _capi.lib().oard_debug_option(b"wgrad_t16", t16)
                own = _Owner()
                for _ in range(2):
                    training._wgrad(dY, ncY, ol, op, MO, X, ncX, silu, il, ip, MI, rows, bias, own, st)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    training._wgrad(dY, ncY, ol, op, MO, X, ncX, silu, il, ip, MI, rows, bias, own, st)
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 5
                line.append(f"{kern} {ms:.3f} ms = {2.0 * rows * MO * MI / ms / 1e9:.1f} TF/s")
            print(f"wgrad {name:15s} rows {rows}: " + " | ".join(line))
    finally:
        _capi.lib().oard_debug_option(b"wgrad_t16", 256)
