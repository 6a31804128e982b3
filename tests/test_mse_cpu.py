"""MSE / IntMSE loss: the CPU restatement against the values and autograd gradients the reference's
loss/MSEloss.py and loss/IntMSEloss.py produced (tests/golden/mse_loss.npz, made by tests/golden/make_mse_golden.py)."""
import argparse
import os

import numpy as np
import torch

from oracle import intel_oracle as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mse_loss.npz'))


def case(ci):
    return {k.split('/', 1)[1]: G[k] for k in G.files if k.startswith('c%d/' % ci) and k.count('/') == 1}


def test_oracle_mse_matches_reference():
    for ci in range(int(G['n_cases'])):
        c = case(ci)
        for div in (0, 1):
            ens = torch.tensor(c['ens'], requires_grad=True)
            w = torch.tensor(c['weights'], requires_grad=True)
            loss = O.mse_loss(ens, torch.tensor(c['ranking']), torch.tensor(c['session_len']), torch.tensor(c['scores']), w, div, 0.01)
            loss.backward()
            assert abs(float(loss) - float(G['c%d/div%d/mse_loss' % (ci, div)])) < 1e-6
            np.testing.assert_allclose(ens.grad.numpy(), G['c%d/div%d/mse_d_ens' % (ci, div)], rtol=1e-5, atol=1e-7)
            if div:
                np.testing.assert_allclose(w.grad.numpy(), G['c%d/div%d/mse_d_w' % (ci, div)], rtol=1e-5, atol=1e-8)


def test_oracle_int_mse_matches_reference():
    for ci in range(int(G['n_cases'])):
        c = case(ci)
        for div in (0, 1):
            cfg = argparse.Namespace(cal_diversity=div, diversity_alpha=0.01, intent_weight=0.1, ensemble_weight=1.0, kl_temp=2.0, kl_weight=0.5)
            out = {'ens_score': torch.tensor(c['ens']), 'weights': torch.tensor(c['weights']), 'intents': torch.tensor(c['pred_int'])}
            batch = {'ranking': torch.tensor(c['ranking']), 'session_len': torch.tensor(c['session_len']), 'scores': torch.tensor(c['scores']),
                     'intents': torch.tensor(c['intents'])}
            total, el, il = O.int_mse_loss(out, batch, cfg)
            assert abs(float(total) - float(G['c%d/div%d/int_total' % (ci, div)])) < 1e-6
            assert abs(float(el) - float(G['c%d/div%d/int_ens' % (ci, div)])) < 1e-6
            assert abs(float(il) - float(G['c%d/div%d/int_intent' % (ci, div)])) < 1e-6
