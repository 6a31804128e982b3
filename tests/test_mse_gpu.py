"""MSE / IntMSE loss kernels against the reference's values and gradients (tests/golden/mse_loss.npz); |d loss| < 1e-5."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mse_loss.npz'))


def case(ci):
    return {k.split('/', 1)[1]: G[k] for k in G.files if k.startswith('c%d/' % ci) and k.count('/') == 1}


@pytest.mark.parametrize('ci', range(int(G['n_cases'])))
@pytest.mark.parametrize('div', [0, 1])
def test_mse_kernels_match_reference(ci, div):
    from intel_sigir2023_amd import loss as LM
    dev = torch.device('cuda:0')
    c = case(ci)
    args = argparse.Namespace(cal_diversity=div, diversity_alpha=0.01, intent_weight=0.1, ensemble_weight=1.0, kl_temp=2.0, kl_weight=0.5)
    ens = torch.tensor(c['ens'], device=dev, requires_grad=True)
    w = torch.tensor(c['weights'], device=dev, requires_grad=True)
    pi = torch.tensor(c['pred_int'], device=dev, requires_grad=True)
    batch = {'ranking': torch.tensor(c['ranking'], device=dev), 'session_len': torch.tensor(c['session_len'], device=dev),
             'scores': torch.tensor(c['scores'], device=dev), 'intents': torch.tensor(c['intents'], device=dev),
             'intentloss_w': torch.tensor(c['intentloss_w'], device=dev)}
    out = {'ens_score': ens, 'weights': w, 'intents': pi}
    loss, _, _ = LM.MSEloss(args)(out, batch)
    loss.backward()
    assert abs(float(loss) - float(G['c%d/div%d/mse_loss' % (ci, div)])) < 1e-5
    np.testing.assert_allclose(ens.grad.cpu().numpy(), G['c%d/div%d/mse_d_ens' % (ci, div)], rtol=2e-5, atol=1e-7)
    if div:
        np.testing.assert_allclose(w.grad.cpu().numpy(), G['c%d/div%d/mse_d_w' % (ci, div)], rtol=2e-5, atol=1e-8)
    ens.grad = None
    total, el, il = LM.IntMSEloss(args)(out, batch)
    total.backward()
    assert abs(float(total) - float(G['c%d/div%d/int_total' % (ci, div)])) < 1e-5
    assert abs(float(el) - float(G['c%d/div%d/int_ens' % (ci, div)])) < 1e-5
    assert abs(float(il) - float(G['c%d/div%d/int_intent' % (ci, div)])) < 1e-5
    np.testing.assert_allclose(ens.grad.cpu().numpy(), G['c%d/div%d/int_d_ens' % (ci, div)], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(pi.grad.cpu().numpy(), G['c%d/div%d/int_d_pred' % (ci, div)], rtol=2e-4, atol=1e-7)


def test_engine_trains_with_int_mse():
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    dev = torch.device('cuda:0')
    args = synth.make_args('tiny', dev)
    corpus, _ = synth.make_corpus('tiny')
    model = IntEL(args, corpus).to(dev)
    eng = IntELEngine(model, 'IntMSEloss', args)
    batch = synth.make_batch('tiny', 32, dev, seed=5, ragged=True)
    losses = [float(eng.train_step(batch)[0]) for _ in range(10)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
