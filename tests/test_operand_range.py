"""Range stress of the half-based operand storages: precision "fp16", "ref" and the pair-operand ViT of
"hybrid" hold operands as IEEE halves / (hi, lo) pairs of halves -- 5 exponent bits.  The reference's own inference entry points run
under fp16 autocast (R: groma/eval/run_groma.py:82-83), where an activation beyond 65504 is inf; real LLaMA residual streams carry
massive activations, so the behaviour at the edges is pinned here instead of being left to N(0, 0.02) test weights:

  * a pair reconstructs to max(2^-22 |x|, 2^-25) for |x| <= 65504 (the documented bound of ops.split_pack), over 1e-7 .. 6e4;
  * weights beyond the storage's range are REPORTED at load time (ops.OperandOverflow), never silently saturated;
  * on the device (GPU tests): a pair GEMM with activations of magnitude 1e-7 .. 1e4 and weight outliers at +-6e4 equals the
    float64 product of the stored operands; conversions of out-of-range ACTIVATIONS saturate at +-65504 (no inf / nan);
    the bf16 build (8 exponent bits) is unaffected."""
import pytest
import torch

from groma_amd import ops


def _wide(shape, lo, hi, seed):
    """random signs, magnitudes log-uniform in [lo, hi]"""
    g = torch.Generator().manual_seed(seed)
    mag = torch.exp(torch.empty(shape).uniform_(float(torch.log(torch.tensor(lo))), float(torch.log(torch.tensor(hi))), generator=g))
    return mag * (torch.randint(0, 2, shape, generator=g) * 2 - 1)


def test_pair_reconstruction_bound_over_the_whole_range():
    x = _wide((64, 4096), 1e-7, 6.0e4, 0)
    back = ops.unsplit(ops.split_pack(x))
    err = (back.double() - x.double()).abs()
    bound = torch.maximum(x.double().abs() * 2.0 ** -22, torch.full_like(err, 2.0 ** -25))
    assert bool((err <= bound).all()), float((err / bound).max())
    # where lo is a normal half (|x| >= 2^-3) the pair carries 22 mantissa bits; at |w| ~ 0.02 (the scale of real weights) ~19
    big = x.abs() >= 0.125
    assert float((err[big] / x.double().abs()[big]).max()) <= 2.0 ** -22
    w = x[(x.abs() > 0.01) & (x.abs() < 0.04)]
    assert float(((ops.unsplit(ops.split_pack(x)) - x)[(x.abs() > 0.01) & (x.abs() < 0.04)].abs() / w.abs()).max()) <= 2.0 ** -18
    # between 65504 and 131008 hi is saturated and lo carries the rest: 11 bits of the remainder
    y = torch.tensor([[7.0e4, -1.2e5, 65504.0, 131008.0] + [0.0] * 28])
    by = ops.unsplit(ops.split_pack(y))
    assert float((by - y).abs().max()) <= 32.0 and torch.isfinite(by).all()
    # beyond that the pair saturates (and says so when asked)
    z = torch.tensor([[3.0e5] + [0.0] * 31])
    assert float(ops.unsplit(ops.split_pack(z))[0, 0]) == 131008.0
    with pytest.raises(ops.OperandOverflow):
        ops.split_pack(z, on_overflow="raise", what="w")
    with pytest.warns(UserWarning):
        ops.split_pack(z, on_overflow="warn", what="w")


def test_weight_overflow_is_reported_at_load_time():
    """weights.bf (every packed GEMM / conv weight) raises for a value the build cannot hold; bf16 holds anything finite"""
    from groma_amd import weights
    w = torch.randn(8, 64) * 0.02
    w[3, 5] = 7.0e4
    assert torch.isfinite(weights.bf(w).float()).all()                 # bf16 (the default build): in range
    with ops.precision("fp16"):
        with pytest.raises(ops.OperandOverflow, match="weight"):
            weights.bf(w)
        assert float(ops.to_h16(w)[3, 5]) == 65504.0                     # the explicit saturating form (what the device does)
    with ops.precision("ref"):
        assert abs(float(ops.unsplit(weights.bf(w))[3, 5]) - 7.0e4) <= 32.0   # a pair still holds it
        w[3, 5] = 2.0e5
        with pytest.raises(ops.OperandOverflow):
            weights.bf(w)
    w[0, 0] = float("nan")
    with ops.precision("fp16"):
        with pytest.raises(ops.OperandOverflow):
            weights.bf(w)


@pytest.mark.gpu
def test_pair_gemm_with_massive_and_tiny_operands(dev):
    """gemm_pair_256_kernel / gemm_pair_kernel on operands far from unit scale: against the float64 product of the STORED operands
    the only error is the dropped lo.lo term and fp32 accumulation -- relative to sum |a||w|, as for well-scaled inputs"""
    M, N, K = 300, 512, 2048
    a = _wide((M, K), 1e-7, 1.0e4, 1)
    a[:, ::97] = _wide((M, len(range(0, K, 97))), 1.0e3, 1.0e4, 2)     # massive-activation columns
    w = torch.randn(N, K, generator=torch.Generator().manual_seed(3)) * 0.02
    w[::31, ::53] = 6.0e4 * torch.sign(w[::31, ::53])                    # outlier weights near the half maximum
    with ops.precision("ref"):
        A, W = ops.split_pack(a).to(dev), ops.split_pack(w).to(dev)
        sa, sw = ops.unsplit(A.cpu()).double(), ops.unsplit(W.cpu()).double()
        ref = sa @ sw.T
        scale = sa.abs() @ sw.abs().T
        for tile in (128, 256):
            out = ops.gemm(A, W, out_f32=True, tile=tile).cpu().double()
            assert torch.isfinite(out).all()
            rel = float(((out - ref).abs() / scale).max())
            print(f"pair GEMM, wide-range operands, tile {tile}: max |err| / sum|a||w| = {rel:.2e}")
            assert rel < 2e-6
        # and against the fp32 inputs themselves: the documented storage bound, summed over K
        ref32 = a.double() @ w.double().T
        bound = (a.double().abs() * 2.0 ** -22 + 2.0 ** -25) @ w.double().abs().T + a.double().abs() @ (w.double().abs() * 2.0 ** -22 + 2.0 ** -25).T
        assert bool(((out - ref32).abs() <= bound + 2e-6 * scale).all())


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp16", "ref", "bf16"])
def test_out_of_range_activations_saturate_on_the_device(dev, precision):
    """a 16-bit OUTPUT beyond the storage's range: the half builds saturate at +-65504 (never inf / nan), bf16 holds it"""
    M, N, K = 64, 256, 256
    with ops.precision(precision):
        a = ops.to_h16(torch.full((M, K), 30.0)).to(dev)
        w = ops.to_h16(torch.full((N, K), 30.0)).to(dev)                 # every output = 256 * 900 = 230 400 > 65504
        out = ops.from_h16(ops.gemm(a, w).cpu())
        assert torch.isfinite(out).all()
        if precision == "bf16":
            assert float((out - 230400.0).abs().max()) <= 230400.0 * 2.0 ** -8
        elif precision == "fp16":
            assert float(out.min()) == 65504.0 and float(out.max()) == 65504.0
        else:
            assert float(out.max()) <= ops.PAIR_MAX and float(out.min()) >= 65504.0
        x = torch.full((M, K), 1.0e5, device=dev)
        y = ops.from_h16(ops.rmsnorm(x, torch.full((K,), 9.0e4, device=dev), 1e-6).cpu())   # normalised row = 1, x gamma = 9e4
        assert torch.isfinite(y).all()
        if precision == "fp16":
            assert float(y.max()) == 65504.0
        elif precision == "bf16":
            assert float((y - 9.0e4).abs().max()) <= 9.0e4 * 2.0 ** -8
