"""GPU parity of the whole hot path against the golden fixtures (reference outputs) and the oracle.

Everything goes through the product's public interface (IntEL module + loss classes, i.e. through the C
ABI).  Tolerances (fp32 path): forward 3e-5 relative to the output scale, losses 1e-5 abs (BASELINE.md
§2), NDCG@3 1e-4 abs, gradients 2e-4 relative to each tensor's max |g| (hand-derived backward vs.
the reference's autograd in different summation order).
"""
import copy

import numpy as np
import pytest
import torch

from oracle import intel_oracle as O
from tests.helpers import CONFIG_NAMES, Fixture, build_model, grad_projection

pytestmark = pytest.mark.gpu
LOSS_TOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


@pytest.fixture(scope='module', params=CONFIG_NAMES)
def fx(request):
    return Fixture(request.param)


def _close(got, ref, tol, name):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    assert err <= tol * scale, '%s: max err %.3e (scale %.3e)' % (name, err, scale)


def test_state_dict_keys_match_reference(fx):
    model, _ = build_model(fx, _dev())
    assert sorted(model.state_dict().keys()) == sorted(fx.group('sd').keys())
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(fx['sd/' + k].shape), k


def test_forward_matches_reference(fx):
    dev = _dev()
    model, _ = build_model(fx, dev)
    model.eval()
    with torch.no_grad():
        out = model(fx.batch(dev))
    for k in ('weights', 'ens_score', 'intents'):
        assert out[k].dtype == torch.float32 and tuple(out[k].shape) == fx['out/' + k].shape
        _close(out[k].cpu().numpy(), fx['out/' + k], 3e-5, k)
    # NDCG@3 of the HIP scores vs the reference scores (north_star: within 1e-4)
    b = fx.batch()
    ref_ndcg = O.ndcg_at_k(fx['out/ens_score'], b['ranking'].numpy(), b['session_len'].numpy(), 3)
    got_ndcg = O.ndcg_at_k(out['ens_score'].cpu().numpy(), b['ranking'].numpy(), b['session_len'].numpy(), 3)
    assert abs(ref_ndcg - got_ndcg) <= 1e-4


def test_losses_match_reference(fx):
    from intel_sigir2023_amd import loss as LS
    dev = _dev()
    _, args = build_model(fx, dev)
    batch = fx.batch(dev)
    batch['bpr_noise'] = torch.from_numpy(fx['bpr/noise']).to(dev)
    out = {k: torch.from_numpy(fx['out/' + k]).to(dev) for k in ('weights', 'ens_score', 'intents')}
    for div in (0, 1):
        a = copy.copy(args)
        a.cal_diversity = div
        l, _, _ = LS.BPRloss(a)(out, batch)
        assert l.dtype == torch.float32
        assert abs(float(l) - float(fx['bpr/loss%d' % div])) < LOSS_TOL, 'bpr div=%d' % div
        l, _, _ = LS.Listloss(a)(out, batch)
        assert abs(float(l) - float(fx['pl/loss%d' % div])) < LOSS_TOL, 'pl div=%d' % div
    crit = LS.IntBPRloss(args)
    il, ce, kl = crit.get_intloss(out, batch)
    assert il.dtype == torch.float64
    for got, key in ((il, 'int/loss'), (ce, 'int/ce'), (kl, 'int/kl')):
        assert abs(float(got) - float(fx[key])) < LOSS_TOL, key
    outz = dict(out)
    outz['intents'] = torch.from_numpy(fx['intz/pred']).to(dev)
    il, ce, kl = crit.get_intloss(outz, batch)
    for got, key in ((il, 'intz/loss'), (ce, 'intz/ce'), (kl, 'intz/kl')):
        assert abs(float(got) - float(fx[key])) < LOSS_TOL, key
    # the sampled negatives are bit-exact (integer work)
    sel = LS.bpr_select_index(out['ens_score'], batch, batch['bpr_noise']).cpu().numpy()
    cb = fx.batch()
    ref_sel = O.bpr_select(cb['ranking'], cb['session_len'], torch.from_numpy(fx['bpr/noise'])).numpy()
    assert (sel == ref_sel).all()


@pytest.mark.parametrize('tag', ['bpr', 'pl'])
def test_total_loss_and_grads_match_reference(fx, tag):
    from intel_sigir2023_amd import loss as LS
    if fx.detail == 'bpr' and tag == 'pl':
        pytest.skip('fixture keeps IntBPRloss grads only')
    dev = _dev()
    model, args = build_model(fx, dev)
    model.train()
    args.cal_diversity = 1
    batch = fx.batch(dev)
    batch['bpr_noise'] = torch.from_numpy(fx['bpr/noise']).to(dev)
    crit = (LS.IntBPRloss if tag == 'bpr' else LS.IntListloss)(args)
    out = model(batch)
    loss, ens, itl = crit(out, batch)
    assert loss.dtype == torch.float64
    assert abs(float(loss) - float(fx['int%s/loss' % tag])) < LOSS_TOL
    assert abs(float(ens) - float(fx['int%s/ens' % tag])) < LOSS_TOL
    assert abs(float(itl) - float(fx['int%s/int' % tag])) < LOSS_TOL
    loss.backward()
    named = dict(model.named_parameters())
    if fx.detail == 'proj':
        for name, ref in fx.group('gradproj_' + tag).items():
            g = named[name].grad
            g = np.zeros(tuple(named[name].shape), np.float32) if g is None else g.cpu().numpy()
            got = grad_projection(g)
            scale = max(1e-6, ref[1])
            assert abs(got[0] - ref[0]) < 5e-4 * scale + 1e-7, (name, got, ref)
            assert abs(got[1] - ref[1]) < 5e-4 * scale + 1e-7, (name, got, ref)
        return
    rows = fx.group('grad_%s_rows' % tag)
    for name, ref in fx.group('grad_' + tag).items():
        g = named[name].grad
        g = torch.zeros_like(named[name]) if g is None else g
        g = g.cpu()
        if name in rows:
            r = torch.from_numpy(rows[name])
            mask = torch.ones(g.shape[0], dtype=torch.bool)
            mask[r] = False
            assert float(g[mask].abs().max()) == 0.0, name
            g = g[r]
        tol = 1e-6 + 2e-4 * float(np.abs(ref).max())
        err = float(np.abs(g.numpy() - ref).max())
        assert err <= tol, '%s: grad err %.3e > %.3e' % (name, err, tol)


def test_adam_two_steps_match_reference(fx):
    """helpers/BaseRunner.py:283-289 with torch.optim.Adam driving the HIP forward/backward."""
    from intel_sigir2023_amd import loss as LS
    from tests.test_oracle_golden import check_adam_result
    if fx.detail != 'full':
        pytest.skip('fixture has no Adam section')
    dev = _dev()
    model, args = build_model(fx, dev)
    model.train()
    args.cal_diversity = 1
    lr, l2 = [float(x) for x in fx['adam/lr_l2']]
    opt = torch.optim.Adam(model.customize_parameters(), lr=lr, weight_decay=l2)
    crit = LS.IntBPRloss(args)
    batch = fx.batch(dev)
    for step in range(2):
        opt.zero_grad()
        batch['bpr_noise'] = torch.from_numpy(fx['adam/noise%d' % step]).to(dev)
        loss, _, _ = crit(model(batch), batch)
        assert abs(float(loss) - float(fx['adam/losses'][step])) < LOSS_TOL
        loss.backward()
        opt.step()
    rows = fx.group('adam_rows')
    named = dict(model.named_parameters())
    for name, ref in fx.group('adam').items():
        if name in ('losses', 'lr_l2', 'noise0', 'noise1'):
            continue
        got = named[name].detach().cpu()
        if name in rows:
            got = got[torch.from_numpy(rows[name])]
        check_adam_result(fx, name, got.numpy(), ref, lr)
