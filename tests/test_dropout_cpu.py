"""nn.Dropout of the tower layers (the paper's IntEL-MSE runs use --dropout 0.5): the CPU restatement
with the keep masks of tests/golden/intel_dropout.npz reproduces the reference's TRAINING-mode forward, IntMSEloss and
parameter gradients (fixture made by tests/golden/make_dropout_golden.py with the Bernoulli draw pinned)."""
import numpy as np
import torch

from oracle import intel_oracle as O
from tests.helpers import Fixture


def load():
    fx = Fixture('dropout')
    cfg = O.Config(**fx.args)
    keep = ([torch.from_numpy(m) for m in fx['keep_i']], [torch.from_numpy(m) for m in fx['keep_s']])
    return fx, cfg, keep


def test_oracle_training_forward_with_pinned_dropout_matches_reference():
    fx, cfg, keep = load()
    sd = {k: v.clone().requires_grad_(True) for k, v in fx.state_dict().items()}
    batch = fx.batch()
    out = O.forward(sd, batch, cfg, dropout_keep=keep)
    for k in ('weights', 'ens_score', 'intents'):
        np.testing.assert_allclose(out[k].detach().numpy(), fx['out/' + k], atol=2e-5, rtol=1e-5, err_msg=k)
    total, el, il = O.int_mse_loss(out, batch, cfg)
    assert abs(float(total) - float(fx['loss'])) < 1e-5 and abs(float(el) - float(fx['loss_ens'])) < 1e-5
    total.backward()
    for name, ref in fx.group('grad').items():
        g = sd[name].grad
        g = np.zeros(ref.shape, np.float32) if g is None else g.numpy()
        assert np.abs(g - ref).max() <= 1e-6 + 2e-4 * np.abs(ref).max(), name


def test_evaluation_forward_ignores_dropout():
    fx, cfg, keep = load()
    with torch.no_grad():
        a = O.forward(fx.state_dict(), fx.batch(), cfg)
        b = O.forward(fx.state_dict(), fx.batch(), cfg, dropout_keep=keep)
    assert float((a['ens_score'] - b['ens_score']).abs().max()) > 1e-3
