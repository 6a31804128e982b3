#!/usr/bin/env python3
"""Long-horizon golden fixture: the REFERENCE's own training loop run for 40 optimizer steps (helpers/BaseRunner.py:279-290:
zero_grad -> model(batch) -> loss -> backward -> optimizer.step, torch.optim.Adam over BaseModel.customize_parameters with coupled
L2, BaseRunner.py:182-188) over three alternating batches, with the learning rate halved after step 20 the way the runner's
StepLR does it (BaseRunner.py:200-203: scheduler.step() once per epoch).  Build container only (needs /root/reference); the tests
read the committed trajectory_*.npz.

Stored: initial state_dict, the three batches, per-step (loss, ensemble loss, intent loss), the final dense parameters (weight matrices of more than 4096 elements as a random
projection + norm of their 40-step update), and for the embedding tables the touched rows + a sample of untouched ones (their trajectory is pure weight decay: the dense /
lazy table Adam must reproduce it over 40 steps).  IntBPRloss draws its tie-breaking noise with torch.rand from the global
CPU generator: the script seeds it per step (NOISE_SEED + step) right before the loss, the test re-draws the same numbers on
the CPU with the same seeds (same torch build in both containers) and hands them to the engine; the first three draws are stored to
pin that assumption."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G      # noqa: E402

STEPS, LR, L2, LR_DROP_AT, NOISE_SEED = 40, 1e-3, 1e-4, 20, 4242

CONFIGS = {
    # benchmarked widths (item tower 128, score tower 64, encoders 128): the one-kernel tower layers / fused encoder blocks
    'w64': (dict(context_emb_size=64, i_emb_size=64, u_emb_size=64, s_emb_size=64, im_emb_size=64, intent_emb_size=64, cross_attn_qsize=64),
            dict(B=4, L=50, lens=[50, 50, 50, 31], I=30, H=20, items=600, users=120, classes=40, ctx=60)),
    # the reference's default widths with GRU4Rec encoders, two heads, two tied layers (script/IntEL.sh:15)
    'pub': (dict(encoder='GRU4Rec', num_heads=2, num_layers=2, context_emb_size=64, intent_emb_size=32, intent_weight=0.01, diversity_alpha=1e-5),
            dict(B=4, L=50, lens=[50, 44, 50, 9], I=30, H=20, items=600, users=120, classes=40, ctx=60)),
}


def run(name, loss_name, seed, store_sd=True):
    from models.IntEL.IntEL import IntEL
    import importlib
    over, shape = CONFIGS[name]
    args = G.make_args(over)
    args.cal_diversity = 1
    corpus = G.make_corpus(shape)
    torch.manual_seed(seed)
    model = IntEL(args, corpus)
    rng = np.random.default_rng(seed)
    batches = [G.make_batch(shape, args.model_num, rng, args.history_max) for _ in range(3)]
    out = {'cfg': np.array(json.dumps(dict(args={k: v for k, v in vars(args).items() if k != 'device'}, shape=shape, seed=seed, loss=loss_name,
                                           steps=STEPS, lr=LR, l2=L2, lr_drop_at=LR_DROP_AT, noise_seed=NOISE_SEED)))}
    sd0 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    if store_sd:      # (the second w64 run starts from the same seed: its test reads the first fixture's state_dict)
        for k, v in sd0.items():
            out['sd/' + k] = v
    for i, b in enumerate(batches):
        for k, v in b.items():
            out['in%d/%s' % (i, k)] = v
    tbs = [G.to_torch(b) for b in batches]
    crit = getattr(importlib.import_module('loss.' + loss_name), loss_name)(args)
    model.train()
    opt = torch.optim.Adam(model.customize_parameters(), lr=LR, weight_decay=L2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)      # BaseRunner.py:200: StepLR(optimizer, decay_step, decay_lr)
    B, L = batches[0]['i_id_s'].shape
    losses = np.zeros((STEPS, 3))
    for step in range(STEPS):
        if step == LR_DROP_AT:
            sched.step()
        tb = tbs[step % 3]
        opt.zero_grad()
        oo = model(tb)
        torch.manual_seed(NOISE_SEED + step)
        if step < 3:
            out['noise%d' % step] = torch.rand(B, L, L).numpy()
            torch.manual_seed(NOISE_SEED + step)
        loss, ens, itl = crit(oo, tb)
        loss.backward()
        opt.step()
        losses[step] = [float(loss.detach()), float(ens.detach()), float(itl.detach())]
    out['losses'] = losses
    touched = {'iid_embeddings.weight': np.unique(np.concatenate([np.concatenate([b['i_id_s'].ravel(), b['his_item_id'].ravel()]) for b in batches])),
               'uid_embeddings.weight': np.unique(np.concatenate([b['u_id_c'] for b in batches]))}
    for pn, p in model.named_parameters():
        if pn in touched:
            extra = np.arange(0, p.shape[0], max(1, p.shape[0] // 48))
            rows = np.unique(np.concatenate([touched[pn], extra]))
            out['final_rows/' + pn] = rows
            out['final/' + pn] = p.detach()[torch.from_numpy(rows)].numpy()
        elif p.numel() > 4096:      # large weight matrices: projection and norm of the 40-step UPDATE (keeps the fixture small; r regenerated from PROJ_SEED)
            out['finalproj/' + pn] = G.grad_projection(pn, p.detach().numpy() - sd0[pn])
        else:
            out['final/' + pn] = p.detach().numpy().copy()
    path = os.path.join(HERE, 'trajectory_%s_%s.npz' % (name, loss_name))
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0), 'loss %.6f -> %.6f' % (losses[0, 0], losses[-1, 0]))


if __name__ == '__main__':
    G.install_shims()
    torch.set_num_threads(4)
    run('w64', 'IntListloss', seed=31)
    run('w64', 'IntBPRloss', seed=31, store_sd=False)      # same initial state and batches as the run above
    run('pub', 'IntBPRloss', seed=33)
