#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, which does not exist on the GPU
box).  Nothing here is imported by the product or by the tests: the tests read the
committed ``*.npz`` files only.  Re-run with ``python tests/golden/make_golden.py``.

What is captured (fixtures F1-F7, F9):
  F1  forward:  state_dict + batch dict  -> weights, ens_score, intents
  F2  BPRloss:  (+ the torch.rand noise the reference drew)  -> loss with/without diversity
  F3  Listloss: -> loss with/without diversity
  F4  intent CE / KL
  F5  grads of every parameter after IntBPRloss / IntListloss backward
  F6  BaseRunner.evaluate_method input lists -> metrics dict
  F7  parameters after 2 torch.optim.Adam steps (param groups of BaseModel.customize_parameters)
  F9  LifeData-shape and stress-shape variants of the above

The reference is imported unmodified; three shims are installed first (tensorboard stub,
numpy-2 aliases, nothing else).  Inputs are synthetic and seeded; their layout is exactly
what BaseModel.Dataset.collate_batch emits.
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

REF_SRC = '/root/reference/IntEL/src'
OUT_DIR = os.path.dirname(os.path.abspath(__file__))


def install_shims():
    sys.dont_write_bytecode = True
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    # (1) tensorboard is not installed; BaseRunner imports it unconditionally.
    tb = types.ModuleType('torch.utils.tensorboard')

    class SummaryWriter(object):
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass
    tb.SummaryWriter = SummaryWriter
    sys.modules['torch.utils.tensorboard'] = tb
    # (2) numpy 2 removed these aliases.
    for name, val in (('object', object), ('float', float), ('int', int), ('bool', bool)):
        if not hasattr(np, name):
            setattr(np, name, val)


# ----------------------------------------------------------------------------------------
# configs
# ----------------------------------------------------------------------------------------
BASE_ARGS = dict(
    # BaseModel / GeneralSeq / IntEL flags (reference defaults)
    model_path='', buffer=1, model_num=3, history_max=20, encoder='BERT4Rec',
    context_emb_size=16, i_emb_size=16, u_emb_size=32, s_emb_size=32, im_emb_size=16,
    intent_emb_size=16, cross_attn_qsize=32, num_heads=1, dropout=0, num_layers=1,
    cross_attention=1,
    # loss flags
    intent_weight=0.1, ensemble_weight=1, kl_temp=2, kl_weight=0.5, cal_diversity=0,
    diversity_alpha=0.01,
)

PROJ_SEED = 12345
DETAIL = {'default': 'full', 'gru_bpr': 'full', 'noxatt': 'full', 'tmall64': 'bpr',
          'lifedata': 'proj', 'stress': 'proj'}


def grad_projection(name, g):
    """(sum g*r, ||g||) with r ~ N(0,1) seeded by PROJ_SEED and the parameter's shape."""
    r = np.random.default_rng(PROJ_SEED).standard_normal(g.shape)
    g64 = g.astype(np.float64)
    return np.array([(g64 * r).sum(), np.sqrt((g64 * g64).sum())])


CONFIGS = {
    # name: (arg overrides, shape dict)
    'default': (dict(), dict(B=4, L=50, lens=[50, 50, 37, 12], I=30, H=20, items=3000, users=400,
                             classes=60, ctx=100)),
    'gru_bpr': (dict(encoder='GRU4Rec', num_heads=2, num_layers=2, context_emb_size=64,
                     intent_emb_size=32, intent_weight=0.01, diversity_alpha=1e-5),
                dict(B=4, L=50, lens=[50, 44, 50, 9], I=30, H=20, items=3000, users=400,
                     classes=60, ctx=100)),
    'noxatt': (dict(cross_attention=0, num_heads=2, num_layers=2),
               dict(B=4, L=50, lens=[50, 50, 37, 12], I=30, H=20, items=3000, users=400,
                    classes=60, ctx=100)),
    # detail levels: 'full' = elementwise grads for both losses + Adam; 'bpr' = elementwise
    # IntBPRloss grads only; 'proj' = per-parameter random projections of the grads (keeps the
    # fixture small; the projection vector is regenerated from PROJ_SEED by the test).
    'tmall64': (dict(context_emb_size=64, i_emb_size=64, u_emb_size=64, s_emb_size=64,
                     im_emb_size=64, intent_emb_size=64, cross_attn_qsize=64),
                dict(B=4, L=50, lens=[50, 50, 50, 31], I=30, H=20, items=2000, users=300,
                     classes=60, ctx=100)),
    'lifedata': (dict(model_num=5, context_emb_size=32, i_emb_size=32, u_emb_size=32,
                      s_emb_size=32, im_emb_size=32, intent_emb_size=32, cross_attn_qsize=32),
                 dict(B=3, L=100, lens=[100, 63, 100], I=10, H=20, items=1500, users=200,
                      classes=40, ctx=50)),
    'stress': (dict(model_num=8, history_max=200, context_emb_size=64, i_emb_size=64,
                    u_emb_size=64, s_emb_size=64, im_emb_size=64, intent_emb_size=64,
                    cross_attn_qsize=64, num_heads=2, num_layers=2),
               dict(B=2, L=200, lens=[200, 171], I=32, H=200, items=1500, users=100,
                    classes=40, ctx=50)),
}


def make_args(over):
    d = dict(BASE_ARGS)
    d.update(over)
    ns = argparse.Namespace(**d)
    ns.device = torch.device('cpu')
    return ns


def make_corpus(shape):
    # Everything IntEL.__init__ reads from the corpus (IntEL.py:38-39,99; BaseModel.py:40,154-155)
    return types.SimpleNamespace(
        itemfnum=[shape['classes']], contextfnum=[shape['ctx']],
        zero_int=np.zeros(shape['I']), max_uid=shape['users'] - 1, max_iid=shape['items'] - 1)


def make_batch(shape, K, rng, hist_max):
    """A batch dict with the dtypes/padding of collate_batch (BaseModel.py:121-142)."""
    B, L, I = shape['B'], shape['L'], shape['I']
    lens = np.asarray(shape['lens'], dtype=np.int64)
    assert lens.max() == L and len(lens) == B
    H = min(shape['H'], hist_max)
    # history lengths: first = full, one session without history (len 1, zero rows)
    hl = rng.integers(1, H + 1, size=B)
    hl[0] = H
    hil = rng.integers(1, H + 1, size=B)
    hil[min(1, B - 1)] = H
    nohist = B - 1
    hl[nohist] = 1
    hil[nohist] = 1
    Hm, Him = int(hl.max()), int(hil.max())

    i_id = np.zeros((B, L), dtype=np.int64)
    i_class = np.zeros((B, L), dtype=np.int64)
    scores = np.zeros((B, L, K), dtype=np.float64)
    ranking = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        n = int(lens[b])
        i_id[b, :n] = rng.integers(1, shape['items'], size=n)
        i_class[b, :n] = rng.integers(0, shape['classes'], size=n)
        raw = rng.normal(size=(n, K)) * 5.0
        scores[b, :n] = (raw - raw.min(0)) / (raw.max(0) - raw.min(0) + 1e-6)   # BaseModel.py:172-173
        r = np.zeros(n, dtype=np.int64)
        r[:5] = [3, 2, 1, 1, 1]
        if b == 1 and n > 8:
            r[-3:] = -1          # unlabelled tail (BaseModel.py:183)
        if b == B - 1:
            r[:] = 1             # edge: positives with no lower tier (BPRloss.py:23-28)
            r[0] = 2
        ranking[b, :n] = rng.permutation(r)

    def softmax_rows(x):
        e = np.exp(x - x.max(-1, keepdims=True))
        return e / e.sum(-1, keepdims=True)

    his_intents = np.zeros((B, Hm, I), dtype=np.float64)
    his_ctx = np.zeros((B, Hm), dtype=np.int64)
    his_item = np.zeros((B, Him), dtype=np.int64)
    his_item_int = np.zeros((B, Him, I), dtype=np.float64)
    for b in range(B):
        if b == nohist:
            continue          # zeros[1,I], context 0, item 0 (GeneralSeq.py:48-52, IntEL.py:234-237)
        his_intents[b, :hl[b]] = softmax_rows(rng.normal(size=(hl[b], I)) * 2)
        his_ctx[b, :hl[b]] = rng.integers(0, shape['ctx'], size=hl[b])
        his_item[b, :hil[b]] = rng.integers(1, shape['items'], size=hil[b])
        his_item_int[b, np.arange(hil[b]), rng.integers(0, I, size=hil[b])] = 1.0
    intents = softmax_rows(rng.normal(size=(B, I)) * 2)
    intents[0, : I // 3] = 0.0        # exact zeros exercise the negative branch of ce_loss
    intents[0] /= intents[0].sum()

    batch = {
        'u_id_c': rng.integers(0, shape['users'], size=B).astype(np.int64),
        'c_id_c': np.arange(B, dtype=np.int64),
        'context_mh': rng.integers(0, shape['ctx'], size=B).astype(np.int64),
        'session_len': lens, 'history_len': hl.astype(np.int64),
        'history_item_len': hil.astype(np.int64),
        'i_id_s': i_id, 'i_class_c': i_class, 'ranking': ranking, 'scores': scores,
        'intents': intents, 'intentloss_w': np.ones((B, I)) / I,
        'his_intents': his_intents, 'his_context_mh': his_ctx,
        'his_item_id': his_item, 'his_item_int': his_item_int,
    }
    return batch


def to_torch(batch):
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    t['batch_size'] = int(batch['u_id_c'].shape[0])
    t['phase'] = 'train'
    return t


def table_rows_touched(batch, name):
    if name == 'iid_embeddings.weight':
        return np.unique(np.concatenate([batch['i_id_s'].ravel(), batch['his_item_id'].ravel()]))
    if name == 'uid_embeddings.weight':
        return np.unique(batch['u_id_c'])
    return None


def run_config(name, seed):
    from models.IntEL.IntEL import IntEL
    from loss.IntBPRloss import IntBPRloss
    from loss.IntListloss import IntListloss
    from loss.BPRloss import BPRloss
    from loss.Listloss import Listloss

    over, shape = CONFIGS[name]
    args = make_args(over)
    corpus = make_corpus(shape)
    torch.manual_seed(seed)
    model = IntEL(args, corpus)
    model.eval()
    rng = np.random.default_rng(seed)
    batch = make_batch(shape, args.model_num, rng, args.history_max)
    out = {}
    out['cfg'] = np.array(json.dumps(dict(args={k: v for k, v in vars(args).items() if k != 'device'},
                                          shape=shape, seed=seed)))
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in sd0.items():
        out['sd/' + k] = v.numpy().copy()
    for k, v in batch.items():
        out['in/' + k] = v

    tb = to_torch(batch)
    # ---- F1 forward
    with torch.no_grad():
        o = model(tb)
    for k in ('weights', 'ens_score', 'intents'):
        out['out/' + k] = o[k].numpy()

    B, L = batch['i_id_s'].shape
    noise_seed = 1000 + seed
    torch.manual_seed(noise_seed)
    noise = torch.rand(B, L, L)
    out['bpr/noise'] = noise.numpy()

    # ---- F2/F3/F4 losses on the forward outputs
    for div in (0, 1):
        args.cal_diversity = div
        with torch.no_grad():
            torch.manual_seed(noise_seed)
            l, _, _ = BPRloss(args)(o, tb)
            out['bpr/loss%d' % div] = l.numpy()
            l, _, _ = Listloss(args)(o, tb)
            out['pl/loss%d' % div] = l.numpy()
    crit = IntBPRloss(args)
    with torch.no_grad():
        il, ce, kl = crit.get_intloss(o, tb)
    out['int/loss'], out['int/ce'], out['int/kl'] = il.numpy(), ce.numpy(), kl.numpy()
    # a prediction with an exact zero exercises the "make soft" branch (BaseIntloss.py:32-35)
    pz = o['intents'].clone()
    pz[:, 0] = 0.0
    with torch.no_grad():
        il, ce, kl = crit.get_intloss({'intents': pz}, tb)
    out['intz/pred'] = pz.numpy()
    out['intz/loss'], out['intz/ce'], out['intz/kl'] = il.numpy(), ce.numpy(), kl.numpy()

    # ---- F5 grads (cal_diversity=1)
    args.cal_diversity = 1
    detail = DETAIL[name]
    out['detail'] = np.array(detail)
    for tag, cls in (('bpr', IntBPRloss), ('pl', IntListloss)):
        model.zero_grad()
        crit = cls(args)
        torch.manual_seed(noise_seed)
        oo = model(tb)
        loss, ens, itl = crit(oo, tb)
        loss.backward()
        out['int%s/loss' % tag] = loss.detach().numpy()
        out['int%s/ens' % tag] = ens.detach().numpy()
        out['int%s/int' % tag] = itl.detach().numpy()
        if detail == 'bpr' and tag == 'pl':
            continue
        for pn, p in model.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            rows = table_rows_touched(batch, pn)
            if detail == 'proj':
                out['gradproj_%s/%s' % (tag, pn)] = grad_projection(pn, g.numpy())
            elif rows is not None:
                # dense embedding grads: keep touched rows, assert the rest is exactly zero
                mask = torch.ones(g.shape[0], dtype=torch.bool)
                mask[torch.from_numpy(rows)] = False
                assert float(g[mask].abs().max()) == 0.0
                out['grad_%s_rows/%s' % (tag, pn)] = rows
                out['grad_%s/%s' % (tag, pn)] = g[torch.from_numpy(rows)].numpy()
            else:
                out['grad_%s/%s' % (tag, pn)] = g.numpy().copy()

    if detail != 'full':
        path = os.path.join(OUT_DIR, 'intel_%s.npz' % name)
        np.savez_compressed(path, **out)
        print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0))
        return
    # ---- F7 two Adam steps, IntBPRloss, coupled L2 (BaseRunner.py:182-188, BaseModel.py:53-62)
    model.load_state_dict(sd0)
    model.train()
    lr, l2 = 1e-3, 1e-4
    opt = torch.optim.Adam(model.customize_parameters(), lr=lr, weight_decay=l2)
    crit = IntBPRloss(args)
    losses = []
    for step in range(2):
        opt.zero_grad()
        torch.manual_seed(noise_seed + step)
        out['adam/noise%d' % step] = torch.rand(B, L, L).numpy()
        torch.manual_seed(noise_seed + step)
        oo = model(tb)
        loss, _, _ = crit(oo, tb)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    out['adam/losses'] = np.array(losses)
    out['adam/lr_l2'] = np.array([lr, l2])
    for pn, p in model.named_parameters():
        rows = table_rows_touched(batch, pn)
        if rows is not None:
            extra = np.arange(0, p.shape[0], max(1, p.shape[0] // 64))
            rows = np.unique(np.concatenate([rows, extra]))
            out['adam_rows/' + pn] = rows
            out['adam/' + pn] = p.detach()[torch.from_numpy(rows)].numpy()
        else:
            out['adam/' + pn] = p.detach().numpy().copy()
    model.load_state_dict(sd0)
    path = os.path.join(OUT_DIR, 'intel_%s.npz' % name)
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0))


def run_metrics(seed):
    """F6: BaseRunner.evaluate_method (BaseRunner.py:56-131) on ragged lists."""
    from helpers.BaseRunner import BaseRunner
    rng = np.random.default_rng(seed)
    n = 37
    session_len = rng.integers(4, 60, size=n)
    session_len[0] = 2           # shorter than max(topk)
    preds, ranks = [], []
    pos = {'c_paynum_i': np.zeros(n, dtype=np.int64), 'c_favnum_i': np.zeros(n, dtype=np.int64),
           'c_clicknum_i': np.zeros(n, dtype=np.int64)}
    Lmax = int(session_len.max())
    for i in range(n):
        ln = int(session_len[i])
        pad_to = Lmax if i % 3 else ln        # some rows arrive padded to the batch length
        p = np.zeros(pad_to, dtype=np.float32)
        p[:ln] = rng.normal(size=ln).astype(np.float32)   # negative scores: pads (0) outrank them
        r = np.zeros(pad_to, dtype=np.int64)
        npay, nfav, nclk = rng.integers(0, 2), rng.integers(0, 2), rng.integers(0, 4)
        if i == 5:
            npay = nfav = 0
            nclk = 1
        lab = [3] * npay + [2] * nfav + [1] * nclk
        lab = lab[:ln]
        if len(lab) == 0:
            lab = [1]
            nclk, npay, nfav = 1, 0, 0
        rr = np.array(lab + [0] * (ln - len(lab)))
        r[:ln] = rng.permutation(rr)
        pos['c_paynum_i'][i] = (r[:ln] == 3).sum()
        pos['c_favnum_i'][i] = (r[:ln] == 2).sum()
        pos['c_clicknum_i'][i] = (r[:ln] == 1).sum()
        preds.append(p)
        ranks.append(r)
    topk = [3, 1, 5, 10]
    res = BaseRunner.evaluate_method(preds, ranks, {k: v.copy() for k, v in pos.items()}, topk,
                                     ['NDCG', 'HR'], session_len.copy())
    out = {'session_len': session_len, 'topk': np.array(topk), 'n': np.array(n)}
    for i in range(n):
        out['pred/%d' % i] = preds[i]
        out['rank/%d' % i] = ranks[i]
    for k, v in pos.items():
        out['pos/' + k] = v
    out['keys'] = np.array(json.dumps(sorted(res.keys())))
    for k, v in res.items():
        out['metric/' + k] = np.array(float(v))
    path = os.path.join(OUT_DIR, 'metrics.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, len(res), 'metrics')


if __name__ == '__main__':
    install_shims()
    torch.set_num_threads(4)
    names = sys.argv[1:] or list(CONFIGS.keys())
    for i, nm in enumerate(names):
        if nm == 'metrics':
            continue
        run_config(nm, seed=11 + list(CONFIGS.keys()).index(nm))
    if not sys.argv[1:] or 'metrics' in sys.argv[1:]:
        run_metrics(seed=5)
