"""Dropout in the HIP path: with the reference's keep masks the training-mode forward, IntMSEloss and every parameter
gradient match the reference (tests/golden/intel_dropout.npz); with the built-in generator the keep rate is 1 - p and
evaluation never drops."""
import numpy as np
import pytest
import torch

from tests.helpers import Fixture, build_model

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def test_training_step_with_pinned_dropout_matches_reference():
    from intel_sigir2023_amd import loss as LS
    dev = _dev()
    fx = Fixture('dropout')
    model, args = build_model(fx, dev)
    model.train()
    keep = np.concatenate([fx['keep_i'].ravel(), fx['keep_s'].ravel()]).astype(np.float32)
    model._dropout_keep = torch.from_numpy(keep).to(dev)
    batch = fx.batch(dev)
    out = model(batch)
    for k in ('weights', 'ens_score', 'intents'):
        ref = fx['out/' + k]
        err = float(np.abs(out[k].detach().cpu().numpy() - ref).max())
        assert err <= 3e-5 * max(1.0, float(np.abs(ref).max())), (k, err)
    loss, el, il = LS.IntMSEloss(args)(out, batch)
    assert abs(float(loss) - float(fx['loss'])) < 1e-5 and abs(float(el) - float(fx['loss_ens'])) < 1e-5
    loss.backward()
    named = dict(model.named_parameters())
    for name, ref in fx.group('grad').items():
        g = named[name].grad
        g = np.zeros(ref.shape, np.float32) if g is None else g.cpu().numpy()
        tol = 1e-6 + 2e-4 * float(np.abs(ref).max())
        assert float(np.abs(g - ref).max()) <= tol, name


def test_builtin_generator_keep_rate_and_eval_mode():
    dev = _dev()
    fx = Fixture('dropout')
    model, args = build_model(fx, dev)
    batch = fx.batch(dev)
    model.eval()
    with torch.no_grad():
        e1 = model(batch)['ens_score'].clone()
        e2 = model(batch)['ens_score'].clone()
    assert torch.equal(e1, e2)                     # evaluation: no dropout, deterministic
    model.train()
    torch.manual_seed(1)
    t1 = model(batch)['ens_score'].detach().clone()
    t2 = model(batch)['ens_score'].detach().clone()
    assert float((t1 - t2).abs().max()) > 1e-4     # a fresh mask per forward
    assert float((t1 - e1).abs().max()) > 1e-4
    torch.manual_seed(1)
    t3 = model(batch)['ens_score'].detach().clone()
    assert torch.equal(t1, t3)                     # reproducible under torch.manual_seed


def test_engine_trains_int_mse_with_dropout():
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = _dev()
    args = synth.make_args('tiny', dev, dropout=0.5)
    corpus, _ = synth.make_corpus('tiny')
    model = IntEL(args, corpus).to(dev)
    model.train()
    eng = IntELEngine(model, 'IntMSEloss', args)
    batch = synth.make_batch('tiny', 32, dev, seed=5, ragged=True)
    losses = [float(eng.train_step(batch)[0]) for _ in range(30)]
    assert all(np.isfinite(losses)) and np.mean(losses[-5:]) < np.mean(losses[:5])
