"""The aWELv_IntEL softmax-weight variant (reference models/supervise/aWELv_IntEL.py:149-203): the oracle's
restatement against the fixture the unmodified reference class produced (tests/golden/make_awelv_golden.py): forward
outputs, IntListloss (+ diversity) and the autograd gradient of every parameter."""
import numpy as np
import torch

from oracle import intel_oracle as O
from tests.helpers import Fixture, build_model


def test_oracle_awelv_forward_loss_and_grads_match_reference():
    fx = Fixture('awelv')
    assert fx.args['model_name'] == 'aWELv_IntEL'
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in fx.state_dict().items()}
    batch = fx.batch()
    cfg = O.Config(**fx.args)
    out = O.forward(sd, batch, cfg)
    for k in ('weights', 'ens_score', 'intents'):
        np.testing.assert_allclose(out[k].detach().numpy(), fx['out/' + k], atol=2e-5, rtol=1e-5, err_msg=k)
    w = out['weights'].detach()
    assert float((w.sum(-1) - 1).abs().max()) < 1e-5                     # a softmax over the K weights ...
    assert float((w - w[:, :1]).abs().max()) == 0.0                       # ... repeated over the list, pads included
    loss, ens, itl = O.int_list_loss(out, batch, cfg)
    assert abs(float(loss) - float(fx['intpl/loss'])) < 1e-5 and abs(float(ens) - float(fx['intpl/ens'])) < 1e-5
    loss.backward()
    rows = fx.group('grad_pl_rows')
    for name, ref in fx.group('grad_pl').items():
        g = sd[name].grad
        g = torch.zeros_like(sd[name]) if g is None else g
        if name in rows:
            g = g[torch.from_numpy(rows[name])]
        tol = 1e-6 + 1e-4 * float(np.abs(ref).max())
        np.testing.assert_allclose(g.numpy(), ref, atol=tol, rtol=1e-4, err_msg=name)


def test_awelv_class_has_the_reference_state_dict_and_flags():
    import argparse
    from intel_sigir2023_amd.model import aWELv_IntEL
    fx = Fixture('awelv')
    model, _ = build_model(fx, torch.device('cpu'))            # strict load of the reference class's state_dict
    assert isinstance(model, aWELv_IntEL) and model._desc.pool_mean == 1 and model._desc.weight_norm == 2
    assert sorted(model.state_dict().keys()) == sorted(fx.group('sd').keys())
    p = aWELv_IntEL.parse_model_args(argparse.ArgumentParser())
    flags = {a.dest for a in p._actions}
    assert 'cross_attention' not in flags and 'weight_norm' not in flags and {'num_heads', 'cross_attn_qsize', 'encoder'} <= flags
