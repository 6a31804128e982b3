"""Generates tests/golden/mse_loss.npz by running the reference's MSEloss / IntMSEloss (loss/MSEloss.py,
loss/IntMSEloss.py, imported from /root/reference -- available in the build container only) on small seeded
inputs.  Inputs, losses and autograd gradients are stored; nothing of the reference's source is.

    python tests/golden/make_mse_golden.py
"""
import argparse
import os
import sys

import numpy as np
import torch

REF = '/root/reference/IntEL/src'
HERE = os.path.dirname(os.path.abspath(__file__))


def case(rng, B, L, K, I):
    slen = rng.integers(1, L + 1, size=B)
    slen[0] = L
    ranking = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        n = slen[b]
        lab = rng.choice([3, 2, 1, 0, -1], size=n, p=[0.05, 0.05, 0.2, 0.6, 0.1])
        ranking[b, :n] = lab
    scores = rng.random((B, L, K))
    for b in range(B):
        scores[b, slen[b]:] = 0
    ens = rng.normal(size=(B, L)).astype(np.float32)
    w = rng.normal(size=(B, L, K)).astype(np.float32)
    pred_int = rng.random((B, I)).astype(np.float32)
    pred_int /= pred_int.sum(1, keepdims=True)
    true_int = rng.random((B, I))
    true_int /= true_int.sum(1, keepdims=True)
    return dict(session_len=slen.astype(np.int64), ranking=ranking, scores=scores, ens=ens, weights=w, pred_int=pred_int, intents=true_int,
                intentloss_w=np.ones((B, I)) / I)


def main():
    sys.path.insert(0, REF)
    from loss.MSEloss import MSEloss
    from loss.IntMSEloss import IntMSEloss
    out = {}
    shapes = [(5, 12, 3, 7), (3, 50, 3, 30), (4, 9, 5, 10)]
    for ci, (B, L, K, I) in enumerate(shapes):
        rng = np.random.default_rng(100 + ci)
        c = case(rng, B, L, K, I)
        for k, v in c.items():
            out['c%d/%s' % (ci, k)] = v
        for div in (0, 1):
            args = argparse.Namespace(cal_diversity=div, diversity_alpha=0.01, intent_weight=0.1, ensemble_weight=1.0, kl_temp=2.0, kl_weight=0.5)
            ens = torch.tensor(c['ens'], requires_grad=True)
            w = torch.tensor(c['weights'], requires_grad=True)
            pi = torch.tensor(c['pred_int'], requires_grad=True)
            od = {'ens_score': ens, 'weights': w, 'intents': pi}
            ib = {'scores': torch.tensor(c['scores']), 'ranking': torch.tensor(c['ranking']), 'session_len': torch.tensor(c['session_len']),
                  'intents': torch.tensor(c['intents']), 'intentloss_w': torch.tensor(c['intentloss_w'])}
            loss, _, _ = MSEloss(args)(od, ib)
            loss.backward()
            out['c%d/div%d/mse_loss' % (ci, div)] = loss.detach().numpy()
            out['c%d/div%d/mse_d_ens' % (ci, div)] = ens.grad.numpy().copy()
            out['c%d/div%d/mse_d_w' % (ci, div)] = (w.grad.numpy().copy() if w.grad is not None else np.zeros_like(c['weights']))
            ens.grad = None
            if w.grad is not None:
                w.grad = None
            total, e_l, i_l = IntMSEloss(args)(od, ib)
            total.backward()
            out['c%d/div%d/int_total' % (ci, div)] = total.detach().numpy()
            out['c%d/div%d/int_ens' % (ci, div)] = e_l.detach().numpy()
            out['c%d/div%d/int_intent' % (ci, div)] = i_l.detach().numpy()
            out['c%d/div%d/int_d_ens' % (ci, div)] = ens.grad.numpy().copy()
            out['c%d/div%d/int_d_pred' % (ci, div)] = pi.grad.numpy().copy()
    out['n_cases'] = np.array(len(shapes))
    np.savez_compressed(os.path.join(HERE, 'mse_loss.npz'), **out)
    print('wrote mse_loss.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
