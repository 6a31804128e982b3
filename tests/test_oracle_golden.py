"""Pins the CPU oracle (oracle/intel_oracle.py) to the reference's own outputs.

The fixtures were produced by tests/golden/make_golden.py, which runs the unmodified reference.
Tolerances: forward 1e-5 abs (fp32 re-association), losses 1e-5 abs,
NDCG/HR metrics exact to 1e-12 (numpy float64 on identical inputs).
"""
import json

import numpy as np
import pytest
import torch

from oracle import intel_oracle as O
from tests.helpers import CONFIG_NAMES, Fixture, GOLDEN, grad_projection

LOSS_TOL = 1e-5


def _cfg(fx, **over):
    kw = dict(fx.args)
    kw.update(over)
    return O.Config(**kw)


@pytest.fixture(scope='module', params=CONFIG_NAMES)
def fx(request):
    return Fixture(request.param)


def test_forward_matches_reference(fx):
    sd, batch = fx.state_dict(), fx.batch()
    with torch.no_grad():
        out = O.forward(sd, batch, _cfg(fx))
    for k in ('weights', 'ens_score', 'intents'):
        ref = fx['out/' + k]
        got = out[k].numpy()
        assert got.shape == ref.shape and got.dtype == ref.dtype
        np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5, err_msg=k)


def test_losses_match_reference(fx):
    batch = fx.batch()
    out = {k: torch.from_numpy(fx['out/' + k]) for k in ('weights', 'ens_score', 'intents')}
    noise = torch.from_numpy(fx['bpr/noise'])
    a = fx.args
    for div in (0, 1):
        l = O.bpr_loss(out['ens_score'], batch['ranking'], batch['session_len'], noise,
                       batch['scores'], out['weights'], div, a['diversity_alpha'])
        assert l.dtype == torch.float32
        assert abs(float(l) - float(fx['bpr/loss%d' % div])) < LOSS_TOL
        l = O.list_loss(out['ens_score'], batch['ranking'], batch['session_len'],
                        batch['scores'], out['weights'], div, a['diversity_alpha'])
        assert l.dtype == torch.float32
        assert abs(float(l) - float(fx['pl/loss%d' % div])) < LOSS_TOL
    il, ce, kl = O.intent_loss(out['intents'], batch['intents'], a['kl_weight'], a['kl_temp'])
    assert il.dtype == torch.float64
    for got, key in ((il, 'int/loss'), (ce, 'int/ce'), (kl, 'int/kl')):
        assert abs(float(got) - float(fx[key])) < LOSS_TOL
    il, ce, kl = O.intent_loss(torch.from_numpy(fx['intz/pred']), batch['intents'], a['kl_weight'], a['kl_temp'])
    for got, key in ((il, 'intz/loss'), (ce, 'intz/ce'), (kl, 'intz/kl')):
        assert abs(float(got) - float(fx[key])) < LOSS_TOL


@pytest.mark.parametrize('tag', ['bpr', 'pl'])
def test_total_loss_and_grads_match_reference(fx, tag):
    if fx.detail == 'bpr' and tag == 'pl':
        pytest.skip('fixture keeps IntBPRloss grads only')
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in fx.state_dict().items()}
    batch = fx.batch()
    cfg = _cfg(fx, cal_diversity=1)
    out = O.forward(sd, batch, cfg)
    if tag == 'bpr':
        loss, ens, itl = O.int_bpr_loss(out, batch, cfg, torch.from_numpy(fx['bpr/noise']))
    else:
        loss, ens, itl = O.int_list_loss(out, batch, cfg)
    assert loss.dtype == torch.float64
    assert abs(float(loss) - float(fx['int%s/loss' % tag])) < LOSS_TOL
    assert abs(float(ens) - float(fx['int%s/ens' % tag])) < LOSS_TOL
    assert abs(float(itl) - float(fx['int%s/int' % tag])) < LOSS_TOL
    loss.backward()
    if fx.detail == 'proj':
        for name, ref in fx.group('gradproj_' + tag).items():
            g = sd[name].grad
            g = np.zeros(sd[name].shape, np.float32) if g is None else g.numpy()
            got = grad_projection(g)
            scale = max(1e-6, ref[1])
            assert abs(got[0] - ref[0]) < 2e-4 * scale + 1e-7, name
            assert abs(got[1] - ref[1]) < 2e-4 * scale + 1e-7, name
        return
    rows = fx.group('grad_%s_rows' % tag)
    for name, ref in fx.group('grad_' + tag).items():
        g = sd[name].grad
        g = torch.zeros_like(sd[name]) if g is None else g
        if name in rows:
            r = torch.from_numpy(rows[name])
            mask = torch.ones(g.shape[0], dtype=torch.bool)
            mask[r] = False
            assert float(g[mask].abs().max()) == 0.0
            g = g[r]
        tol = 1e-6 + 1e-4 * float(np.abs(ref).max())
        np.testing.assert_allclose(g.numpy(), ref, atol=tol, rtol=1e-4, err_msg=name)


def check_adam_result(fx, name, got, ref, lr, steps=2):
    """Adam divides by |g|: where the true gradient is ~0 (e.g. the key bias of a softmax
    attention, whose gradient is analytically zero) the update direction is rounding noise in the
    reference too, so those elements are only required to stay within ``steps*lr`` of it."""
    tol = np.full(ref.shape, 5e-6)
    if 'k_linear.bias' in name:
        tol[:] = 2.2 * steps * lr
    elif ('grad_bpr/' + name) in fx.z.files and ('adam_rows/' + name) not in fx.z.files:
        g = fx['grad_bpr/' + name]
        tol = np.where(np.abs(g) < 2e-7, 2.2 * steps * lr, np.where(np.abs(g) < 1e-5, 2e-4, tol))
    err = np.abs(got - ref) - 1e-4 * np.abs(ref)
    assert (err <= tol).all(), (name, float((err - tol).max()))


def test_adam_two_steps_match_reference(fx):
    if fx.detail != 'full':
        pytest.skip('fixture has no Adam section')
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in fx.state_dict().items()}
    batch = fx.batch()
    cfg = _cfg(fx, cal_diversity=1)
    lr, l2 = [float(x) for x in fx['adam/lr_l2']]
    params = [(k, v) for k, v in sd.items() if v.requires_grad]
    opt = torch.optim.Adam(O.adam_groups(params, l2), lr=lr)
    for step in range(2):
        opt.zero_grad()
        out = O.forward(sd, batch, cfg)
        loss, _, _ = O.int_bpr_loss(out, batch, cfg, torch.from_numpy(fx['adam/noise%d' % step]))
        assert abs(float(loss) - float(fx['adam/losses'][step])) < LOSS_TOL
        loss.backward()
        opt.step()
    rows = fx.group('adam_rows')
    for name, ref in fx.group('adam').items():
        if name in ('losses', 'lr_l2', 'noise0', 'noise1'):
            continue
        got = sd[name].detach()
        if name in rows:
            got = got[torch.from_numpy(rows[name])]
        check_adam_result(fx, name, got.numpy(), ref, lr)


def test_evaluate_method_matches_reference():
    z = np.load(GOLDEN + '/metrics.npz')
    n = int(z['n'])
    preds = [z['pred/%d' % i] for i in range(n)]
    ranks = [z['rank/%d' % i] for i in range(n)]
    pos = {k: z['pos/' + k] for k in ('c_paynum_i', 'c_favnum_i', 'c_clicknum_i')}
    res = O.evaluate_method(preds, ranks, pos, [int(k) for k in z['topk']], ['NDCG', 'HR'], z['session_len'])
    keys = json.loads(str(z['keys']))
    assert sorted(res.keys()) == keys and len(keys) == 25
    for k in keys:
        assert abs(float(res[k]) - float(z['metric/' + k])) < 1e-12, k
