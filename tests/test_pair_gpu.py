"""GPU parity of the one-pass linear backward (csrc/pair.hip, through the C ABI) against a float64 torch evaluation of what autograd
computes for `y = linear(x)` (models/IntEL/IntEL.py:186-187, 195-196): dx = (dy @ w) [* (x > 0)], dw = dy^T x, db = colsum(dy).

Floating point: tolerance 2e-5 (data gradient) / 5e-5 (weight, bias gradient) relative to the output scale -- the bars of the separate kernels in
tests/test_ops_gpu.py (fp32 re-association only: six bf16 plane products = fp32 accuracy)."""
import pytest
import torch

from tests.helpers import KernelTrace

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


def _close(got, ref, tol, name):
    got = got.detach().cpu().double()
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err <= tol * scale, '%s: max err %.3e (scale %.3e)' % (name, err, scale)


# tile edges of both widths (32-row tiles at d = 128, 64-row tiles at d = 64): one tile, a ragged last tile, fewer tiles than workgroups, several
# tiles per workgroup (> 256 tiles), a single row
@pytest.mark.parametrize('M,d', [(32, 128), (64, 64), (1, 128), (1, 64), (50, 128), (50, 64), (200, 128), (333, 64), (4099, 128), (4099, 64),
                                 (20480, 128), (40000, 64), (33, 128), (65, 64)])
@pytest.mark.parametrize('mask', [False, True])
def test_linear_bwd_pair(M, d, mask):
    from intel_sigir2023_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M * 13 + d + int(mask))
    x = torch.randn(M, d, generator=g)
    if mask:
        x = torch.relu(x)
    w = torch.randn(d, d, generator=g) / d ** 0.5
    dy = torch.randn(M, d, generator=g)
    with KernelTrace() as kt:
        dx, dw, db = ops.linear_bwd(dy.to(dev), x.to(dev), w.to(dev), relu_mask=mask)
    kt.check(present=['linear_bwd_pair_kernel'], absent=['wgrad_b3_kernel', 'gemm_rows_b3_kernel'])
    ref = dy.double() @ w.double()
    if mask:
        ref = ref * (x > 0).double()
    _close(dx, ref, 2e-5, 'dx')
    _close(dw, dy.double().t() @ x.double(), 5e-5, 'dw')
    _close(db, dy.double().sum(0), 5e-5, 'db')


def test_linear_bwd_pair_matches_the_separate_kernels_bitwise_in_dx():
    """The data gradient is the same six plane products in the same order as gemm_rows_b3's: identical bits."""
    from intel_sigir2023_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    M, d = 4096, 128
    x = torch.relu(torch.randn(M, d, generator=g)).to(dev)
    w = (torch.randn(d, d, generator=g) / d ** 0.5).to(dev)
    dy = torch.randn(M, d, generator=g).to(dev)
    dx, dw, db = ops.linear_bwd(dy, x, w, relu_mask=False)
    dx2 = ops.linear_dgrad(dy, w)
    dw2, db2 = ops.linear_wgrad(dy, x)
    assert float((dx - dx2).abs().max()) <= 2e-6 * float(dx2.abs().max())
    assert float((dw - dw2).abs().max()) <= 2e-5 * float(dw2.abs().max())
    assert float((db - db2).abs().max()) <= 2e-5 * float(db2.abs().max())


# the fused q/k/v projection's backward (nb = 3; nb = 2 at d = 128): 32-row tiles at d = 128, 64-row tiles at d = 64 -- one tile, ragged tiles, more tiles than workgroups
@pytest.mark.parametrize('M,d,nb', [(32, 128, 3), (64, 64, 3), (1, 128, 3), (1, 64, 3), (50, 128, 3), (50, 64, 3), (333, 64, 3), (4099, 128, 3), (20480, 128, 3), (40000, 64, 3),
                                    (33, 128, 2), (4099, 128, 2), (9000, 128, 2)])
@pytest.mark.parametrize('with_res,with_bias', [(True, False), (False, True)])
def test_linear_bwd_qkv(M, d, nb, with_res, with_bias):
    from intel_sigir2023_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M * 17 + d + nb)
    x = torch.randn(M, d, generator=g)
    w = torch.randn(nb * d, d, generator=g) / d ** 0.5
    dy = torch.randn(M, nb * d, generator=g)
    res = torch.randn(M, d, generator=g) if with_res else None
    with KernelTrace() as kt:
        dx, dw, db = ops.linear_bwd_qkv(dy.to(dev), x.to(dev), w.to(dev), res=None if res is None else res.to(dev), want_bias=with_bias)
    kt.check(present=['linear_bwd_qkv_kernel'], absent=['wgrad_b3_kernel', 'gemm_rows_b3k_kernel'])
    ref = dy.double() @ w.double()
    if res is not None:
        ref = ref + res.double()
    _close(dx, ref, 2e-5, 'dx')
    _close(dw, dy.double().t() @ x.double(), 5e-5, 'dw')
    if with_bias:
        _close(db, dy.double().sum(0), 5e-5, 'db')
