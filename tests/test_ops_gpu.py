"""GPU parity of the building-block kernels (through the C ABI) against torch fp32/fp64 on CPU.

These are floating-point kernels: the comparison is against a float64 torch evaluation of the same
op with tolerance 2e-5 relative to the output scale (fp32 re-association only; the MFMA used is the
exact-fp32 v_mfma_f32_16x16x4_f32).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


def _close(got, ref, tol=2e-5, name=''):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err <= tol * scale, '%s: max err %.3e (scale %.3e)' % (name, err, scale)


@pytest.mark.parametrize('M,K,N', [(64, 16, 16), (200, 128, 128), (130, 3, 64), (77, 30, 64), (300, 384, 30),
                                   (1000, 128, 384), (65, 1071, 16), (50, 64, 200), (700, 384, 128), (4100, 256, 64), (130, 192, 64), (333, 320, 32),
                                   # the branch-free DMA weight-gradient kernel: M % 32 == 0, N and K multiples of 32, every tile-count pair
                                   (256, 128, 128), (96, 64, 64), (4096, 64, 128), (320, 96, 32), (64, 128, 64), (128, 32, 32), (8192, 128, 128),
                                   (2048, 32, 128), (1024, 64, 32), (32, 256, 384), (20480, 64, 64),
                                   # the 16-row workgroups of the B-row chains (N <= 32 split-K, or a ragged K <= 128)
                                   (4096, 384, 30), (4096, 320, 3), (4096, 30, 384), (4096, 3, 320), (4099, 30, 128), (1000, 130, 30), (37, 1000, 3),
                                   (515, 30, 30), (16, 3, 3), (4096, 127, 64)])
def test_linear_fwd_dgrad_wgrad(M, K, N):
    from intel_sigir2023_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + K * 3 + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    y = ops.linear(x.to(dev), w.to(dev), b.to(dev))
    _close(y, x.double() @ w.double().t() + b.double(), name='linear')
    y = ops.linear(x.to(dev), w.to(dev), None, relu=True)
    _close(y, torch.relu(x.double() @ w.double().t()), name='linear+relu')
    dx = ops.linear_dgrad(dy.to(dev), w.to(dev))
    _close(dx, dy.double() @ w.double(), name='dgrad')
    dw, db = ops.linear_wgrad(dy.to(dev), x.to(dev))
    _close(dw, dy.double().t() @ x.double(), tol=5e-5, name='wgrad')
    _close(db, dy.double().sum(0), tol=5e-5, name='bgrad')


def _attn_ref(qkv, B, T, d, heads, key_len):
    dk = d // heads
    x = qkv.double().view(B, T, 3, heads, dk)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2) / dk ** 0.5
    if key_len is not None:
        mask = torch.arange(T)[None, :] < key_len[:, None]
        s = s.masked_fill(~mask[:, None, None, :], float('-inf'))
    p = torch.softmax(s, -1)
    o = p @ v
    return o.transpose(1, 2).reshape(B * T, d)


@pytest.mark.parametrize('B,T,d,heads,masked', [(3, 50, 128, 1, False), (2, 50, 64, 2, False), (4, 20, 128, 2, True),
                                                (2, 200, 128, 2, False), (2, 200, 128, 2, True), (3, 37, 32, 2, True),
                                                (2, 70, 32, 1, False), (2, 100, 96, 2, True), (3, 9, 16, 2, False),
                                                # whole-sequence kernels (T <= 64, head dim 64 / 128): every tile count, ragged pair counts
                                                (5, 20, 128, 2, True), (3, 16, 64, 1, False), (7, 10, 128, 1, True), (3, 33, 128, 1, True),
                                                (2, 48, 128, 2, True), (5, 64, 64, 1, True), (3, 50, 256, 2, False), (2, 20, 64, 1, False),
                                                (9, 50, 128, 1, True), (6, 1, 64, 1, False),
                                                # general kernels on the bf16 pipe (attn_p3.hip: T > 64, head dim 64 / 128): tile boundaries, one and several 64-row blocks
                                                (2, 100, 128, 1, True), (3, 65, 256, 2, True), (2, 97, 128, 1, False), (1, 130, 64, 1, True),
                                                (2, 128, 128, 2, False), (3, 129, 128, 1, True), (2, 96, 64, 1, True)])
def test_attention_fwd_bwd(B, T, d, heads, masked):
    from intel_sigir2023_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + d + heads)
    qkv = torch.randn(B * T, 3 * d, generator=g)
    key_len = None
    if masked:
        key_len = torch.randint(1, T + 1, (B,), generator=g)
        key_len[0] = T
        key_len[-1] = 1
    dout = torch.randn(B * T, d, generator=g)
    qkv_r = qkv.clone().double().requires_grad_(True)
    ref = _attn_ref(qkv_r, B, T, d, heads, key_len)
    (ref * dout.double()).sum().backward()
    kl = key_len.to(torch.int32).to(dev) if masked else None
    out, lse = ops.attention(qkv.to(dev), B, T, d, heads, kl)
    _close(out, ref, name='attn fwd')
    dqkv = ops.attention_bwd(qkv.to(dev), out, dout.to(dev), lse, B, T, d, heads, kl)
    _close(dqkv, qkv_r.grad, tol=5e-5, name='attn bwd')


@pytest.mark.parametrize('M,N', [(10, 32), (130, 128), (77, 64), (5, 200), (1, 256), (4099, 128), (16, 64)])
def test_add_layernorm(M, N):
    from intel_sigir2023_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + N)
    x, r = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    gamma, beta = torch.randn(N, generator=g), torch.randn(N, generator=g)
    y, xhat, rstd = ops.add_layernorm(x.to(dev), r.to(dev), gamma.to(dev), beta.to(dev), stash=True)
    ref = torch.nn.functional.layer_norm((x + r).double(), (N,), gamma.double(), beta.double(), 1e-5)
    _close(y, ref, name='ln')
    _close(xhat * gamma.to(dev) + beta.to(dev), ref, name='xhat')


def test_dispatcher_ops_match_torch_and_differentiate():
    """The torch.library registration (`torch.ops.intel_mi355x.*`, ops.py): the dispatcher-visible ops give the same values as
    the ctypes wrappers, and autograd through them (linear with fused relu, attention) matches torch autograd of the same math."""
    import torch.nn.functional as F
    from intel_sigir2023_amd import ops
    assert ops.REGISTERED_OPS, getattr(ops, '_REGISTER_ERROR', None)
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(200, 64, generator=g)
    w = (torch.randn(128, 64, generator=g) / 8).requires_grad_(True)
    b = torch.randn(128, generator=g).requires_grad_(True)
    xd = x.to(dev).requires_grad_(True)
    wd, bd = w.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    y = torch.ops.intel_mi355x.linear(xd, wd, bd, True)
    xr = x.clone().requires_grad_(True)
    yr = torch.relu(F.linear(xr, w, b))
    _close(y, yr, name='linear')
    dy = torch.randn(200, 128, generator=g)
    y.backward(dy.to(dev))
    yr.backward(dy)
    _close(xd.grad, xr.grad, tol=5e-5, name='dx')
    _close(wd.grad, w.grad, tol=5e-5, name='dw')
    _close(bd.grad, b.grad, tol=5e-5, name='db')
    # attention: [q | k | v] rows of 3 sessions x 20 rows, 2 heads of 32
    B, T, d, heads = 3, 20, 64, 2
    qkv = torch.randn(B * T, 3 * d, generator=g)
    qd = qkv.to(dev).requires_grad_(True)
    out, lse = torch.ops.intel_mi355x.attention(qd, B, T, d, heads)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = [t.view(B, T, heads, d // heads).transpose(1, 2) for t in qr.split(d, dim=1)]
    ref = torch.softmax(q @ k.transpose(-1, -2) / (d // heads) ** 0.5, dim=-1) @ v
    ref = ref.transpose(1, 2).reshape(B * T, d)
    _close(out, ref, name='attention')
    go = torch.randn(B * T, d, generator=g)
    out.backward(go.to(dev))
    ref.backward(go)
    _close(qd.grad, qr.grad, tol=5e-5, name='dqkv')
    # shape inference without running a kernel
    with torch._subclasses.FakeTensorMode():
        fx = torch.empty(10, 64, device='cuda')
        fy = torch.ops.intel_mi355x.linear(fx, torch.empty(32, 64, device='cuda'), torch.empty(32, device='cuda'), False)
        assert tuple(fy.shape) == (10, 32)


def test_whole_model_op_matches_the_reference_fixture():
    """torch.ops.intel_mi355x.intel_forward: IntEL.forward (models/IntEL/IntEL.py:117-124) as ONE dispatcher-visible custom op over the
    module's parameters, intel_backward registered as its autograd formula.  Fixture `default`: forward outputs (3e-5), the
    IntBPRloss value (1e-5) and every parameter gradient of the reference's autograd (2e-4 of the tensor's largest element)."""
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd import ops
    from tests.helpers import Fixture, build_model
    assert 'intel_forward' in ops.REGISTERED_OPS and hasattr(torch.ops.intel_mi355x, 'intel_forward')
    dev = _dev()
    fx = Fixture('default')
    model, args = build_model(fx, dev)
    model.train()
    batch = fx.batch(dev)
    out = ops.model_forward(model, batch)
    ref = fx.group('out')
    for k in ('weights', 'ens_score', 'intents'):
        _close(out[k], torch.from_numpy(ref[k]), tol=3e-5, name=k)
    batch['bpr_noise'] = torch.from_numpy(fx['bpr/noise']).to(dev)
    args.cal_diversity = 0
    loss, _, _ = LS.IntBPRloss(args)(out, batch)
    loss.backward()
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    # the same through the module's default path (autograd.Function over the same C entry points): identical kernels, so equal
    model.zero_grad()
    out2 = model(batch)
    loss2, _, _ = LS.IntBPRloss(args)(out2, batch)
    loss2.backward()
    assert abs(float(loss) - float(loss2)) < 1e-6
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        assert k in got, k
        tol = 1e-7 + 2e-6 * float(p.grad.abs().max())
        assert float((got[k] - p.grad).abs().max()) <= tol, k
    # opcheck-style: the op is visible to the dispatcher with a fake (shape) function
    with torch.no_grad():
        w, e, i = torch.ops.intel_mi355x.intel_forward(ops.model_handle(model), False, batch['i_id_s'], batch['i_class_c'], batch['scores'],
                                                       batch['session_len'], batch['u_id_c'], batch['context_mh'], batch['his_context_mh'],
                                                       batch['his_intents'], batch['history_len'], batch['his_item_id'], batch.get('his_item_idx'),
                                                       batch.get('his_item_int') if 'his_item_idx' not in batch else None, batch['history_item_len'],
                                                       -1, -1, [p for _, _, p in model.slot_items()])
    assert w.shape == out['weights'].shape and e.shape == out['ens_score'].shape and i.shape == out['intents'].shape
