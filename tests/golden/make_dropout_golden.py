"""Generates tests/golden/intel_dropout.npz: a TRAINING-mode forward / IntMSEloss / backward of the reference model
(imported from /root/reference, build container only) with --dropout 0.5 -- the configuration of the paper's
IntEL-MSE runs -- where nn.Dropout's Bernoulli draw is pinned: the module is swapped for one that applies keep
masks stored in the fixture, in call order (item tower layers, then score tower layers, IntEL.py:182-197).

    python tests/golden/make_dropout_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG      # noqa: E402  (argument / batch builders shared with the other fixtures)


class PinnedDropout(torch.nn.Module):
    def __init__(self, p, masks):
        super().__init__()
        self.p, self.masks, self.calls = p, masks, 0

    def forward(self, x):
        m = self.masks[self.calls]
        self.calls += 1
        assert m.shape == x.shape
        return x * m / (1.0 - self.p)


def main():
    sys.path.insert(0, MG.REF_SRC)
    from models.IntEL.IntEL import IntEL
    from loss.IntMSEloss import IntMSEloss
    over = dict(dropout=0.5, num_layers=2, num_heads=2, cal_diversity=1, diversity_alpha=0.01)
    shape = dict(B=3, L=20, lens=[20, 13, 7], I=12, H=6, items=500, users=80, classes=20, ctx=30)
    args = MG.make_args(over)
    corpus = MG.make_corpus(shape)
    seed = 77
    torch.manual_seed(seed)
    model = IntEL(args, corpus)
    rng = np.random.default_rng(seed)
    batch = MG.make_batch(shape, args.model_num, rng, args.history_max)
    B, L = batch['i_id_s'].shape
    d_i, d_s = args.i_emb_size + args.im_emb_size, args.s_emb_size
    keep_i = (rng.random((args.num_layers, B, L, d_i)) >= 0.5).astype(np.float32)
    keep_s = (rng.random((args.num_layers, B, L, d_s)) >= 0.5).astype(np.float32)
    masks = [torch.from_numpy(keep_i[l]) for l in range(args.num_layers)] + [torch.from_numpy(keep_s[l]) for l in range(args.num_layers)]
    out = {'cfg': np.array(json.dumps(dict(args={k: v for k, v in vars(args).items() if k != 'device'}, shape=shape, seed=seed))),
           'detail': np.array('full'), 'keep_i': keep_i, 'keep_s': keep_s}
    for k, v in model.state_dict().items():
        out['sd/' + k] = v.detach().numpy().copy()
    for k, v in batch.items():
        out['in/' + k] = v
    model.train()
    model.dropout_layer = PinnedDropout(args.dropout, masks)
    tb = MG.to_torch(batch)
    o = model(tb)
    assert model.dropout_layer.calls == 2 * args.num_layers
    for k in ('weights', 'ens_score', 'intents'):
        out['out/' + k] = o[k].detach().numpy().copy()
    loss, el, il = IntMSEloss(args)(o, tb)
    loss.backward()
    out['loss'], out['loss_ens'], out['loss_int'] = loss.detach().numpy(), el.detach().numpy(), il.detach().numpy()
    for n, p in model.named_parameters():
        out['grad/' + n] = (p.grad.numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
    np.savez_compressed(os.path.join(HERE, 'intel_dropout.npz'), **out)
    print('wrote intel_dropout.npz', len(out), 'arrays, loss', float(loss))


if __name__ == '__main__':
    main()
