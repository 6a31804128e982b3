#!/usr/bin/env python3
"""Generates tests/golden/intel_awelv.npz by running the reference's aWELv_IntEL model
(/root/reference/IntEL/src/models/supervise/aWELv_IntEL.py, imported unmodified -- available in the build container
only) with IntListloss on a small seeded batch in collate_batch's layout: state_dict, batch, forward outputs, loss and
the autograd gradient of every parameter.  Same file layout as make_golden.py's fixtures (tests/helpers.Fixture reads it);
nothing of the reference's source is stored.

    python tests/golden/make_awelv_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (shims, synthetic batch, helpers)


def main():
    G.install_shims()
    from models.supervise.aWELv_IntEL import aWELv_IntEL
    from loss.IntListloss import IntListloss
    # the published aWELv+IntEL flags' structure (script/baselines.sh:47): 2 heads, 2 tied layers, qsize 64 -- BERT4Rec
    # encoders here (GRU4Rec is covered by intel_gru_bpr.npz), dropout 0 (evaluation-mode parity; dropout has its own fixture)
    over = dict(num_heads=2, num_layers=2, cross_attn_qsize=64, context_emb_size=32, intent_emb_size=32, s_emb_size=32,
                u_emb_size=16, cal_diversity=1, diversity_alpha=1e-2)
    shape = dict(B=4, L=50, lens=[50, 50, 37, 12], I=30, H=20, items=3000, users=400, classes=60, ctx=100)
    seed = 11
    args = G.make_args(over)
    del args.cross_attention                       # the reference class has no such flag
    corpus = G.make_corpus(shape)
    torch.manual_seed(seed)
    model = aWELv_IntEL(args, corpus)
    model.eval()
    rng = np.random.default_rng(seed)
    batch = G.make_batch(shape, args.model_num, rng, args.history_max)
    out = {}
    a = {k: v for k, v in vars(args).items() if k != 'device'}
    a['model_name'] = 'aWELv_IntEL'
    out['cfg'] = np.array(json.dumps(dict(args=a, shape=shape, seed=seed)))
    out['detail'] = np.array('pl')
    for k, v in model.state_dict().items():
        out['sd/' + k] = v.detach().numpy().copy()
    for k, v in batch.items():
        out['in/' + k] = v
    tb = G.to_torch(batch)
    with torch.no_grad():
        o = model(tb)
    for k in ('weights', 'ens_score', 'intents'):
        out['out/' + k] = o[k].numpy()
    model.zero_grad()
    oo = model(tb)
    loss, ens, itl = IntListloss(args)(oo, tb)
    loss.backward()
    out['intpl/loss'], out['intpl/ens'], out['intpl/int'] = loss.detach().numpy(), ens.detach().numpy(), itl.detach().numpy()
    for pn, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        rows = G.table_rows_touched(batch, pn)
        if rows is not None:
            mask = torch.ones(g.shape[0], dtype=torch.bool)
            mask[torch.from_numpy(rows)] = False
            assert float(g[mask].abs().max()) == 0.0
            out['grad_pl_rows/' + pn] = rows
            out['grad_pl/' + pn] = g[torch.from_numpy(rows)].numpy()
        else:
            out['grad_pl/' + pn] = g.numpy().copy()
    path = os.path.join(HERE, 'intel_awelv.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0), 'loss', float(loss))


if __name__ == '__main__':
    main()
