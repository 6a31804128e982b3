"""The fusion-weight variants on the HIP path: the aWELv_IntEL class against the reference's own outputs and gradients
(tests/golden/intel_awelv.npz), and IntEL's `--weight_norm softmax` switch (no reference counterpart)
against the oracle's autograd, with and without cross attention."""
import numpy as np
import pytest
import torch

from oracle import intel_oracle as O
from tests.helpers import Fixture, build_model

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


def test_awelv_forward_loss_and_grads_match_reference():
    from intel_sigir2023_amd import loss as LS
    fx = Fixture('awelv')
    dev = _dev()
    model, args = build_model(fx, dev)
    model.eval()
    with torch.no_grad():
        out = model(fx.batch(dev))
    for k in ('weights', 'ens_score', 'intents'):
        ref = fx['out/' + k]
        err = float(np.abs(out[k].cpu().numpy() - ref).max())
        assert err <= 3e-5 * max(1.0, float(np.abs(ref).max())), (k, err)
    model.train()
    batch = fx.batch(dev)
    out = model(batch)
    loss, ens, itl = LS.IntListloss(args)(out, batch)
    assert abs(float(loss) - float(fx['intpl/loss'])) < 1e-5 and abs(float(ens) - float(fx['intpl/ens'])) < 1e-5
    loss.backward()
    named = dict(model.named_parameters())
    rows = fx.group('grad_pl_rows')
    for name, ref in fx.group('grad_pl').items():
        g = named[name].grad
        g = (torch.zeros_like(named[name]) if g is None else g).cpu()
        if name in rows:
            r = torch.from_numpy(rows[name])
            mask = torch.ones(g.shape[0], dtype=torch.bool)
            mask[r] = False
            assert float(g[mask].abs().max()) == 0.0, name
            g = g[r]
        tol = 1e-6 + 2e-4 * float(np.abs(ref).max())
        err = float(np.abs(g.numpy() - ref).max())
        assert err <= tol, '%s: grad err %.3e > %.3e' % (name, err, tol)


def test_awelv_engine_step_trains():
    from intel_sigir2023_amd.engine import IntELEngine
    fx = Fixture('awelv')
    dev = _dev()
    model, args = build_model(fx, dev)
    model.train()
    eng = IntELEngine(model, 'IntListloss', args, lr=1e-3, l2=1e-4)
    batch = fx.batch(dev)
    losses = [float(eng.train_step(batch)[0]) for _ in range(12)]
    assert abs(losses[0] - float(fx['intpl/loss'])) < 1e-5
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


@pytest.mark.parametrize('name', ['default', 'noxatt', 'gru_bpr'])
def test_weight_norm_softmax_matches_oracle_autograd(name):
    """IntEL --weight_norm softmax: weights = softmax_K(weight_embeddings(.)) for every row (valid rows and pad rows each
    carry their own vector with cross attention; per-item weights without)."""
    from intel_sigir2023_amd import loss as LS
    from intel_sigir2023_amd.model import IntEL
    from tests.helpers import make_args, make_corpus
    fx = Fixture(name)
    dev = _dev()
    a = dict(fx.args)
    a['weight_norm'] = 'softmax'
    a['cal_diversity'] = 1
    args = make_args(a, dev)
    model = IntEL(args, make_corpus(fx.shape))
    model.load_state_dict(fx.state_dict(), strict=True)
    model = model.to(dev)
    model.train()
    batch = fx.batch(dev)
    out = model(batch)
    assert float((out['weights'].sum(-1) - 1).abs().max()) < 1e-5
    loss, _, _ = LS.IntListloss(args)(out, batch)
    loss.backward()
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in fx.state_dict().items()}
    cfg = O.Config(**a)
    cb = fx.batch()
    ref = O.forward(sd, cb, cfg)
    rl, _, _ = O.int_list_loss(ref, cb, cfg)
    rl.backward()
    for k in ('weights', 'ens_score', 'intents'):
        err = float((out[k].detach().cpu() - ref[k].detach()).abs().max())
        assert err <= 3e-5 * max(1.0, float(ref[k].detach().abs().max())), (k, err)
    assert abs(float(loss) - float(rl)) < 1e-5
    for k, p in model.named_parameters():
        g = p.grad.cpu() if p.grad is not None else torch.zeros(p.shape)
        r = sd[k].grad if sd[k].grad is not None else torch.zeros(p.shape)
        tol = 1e-6 + 2e-4 * float(r.abs().max())
        assert float((g - r).abs().max()) <= tol, k
