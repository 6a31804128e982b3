"""Synthetic Tmall-shape / LifeData-shape / stress-shape workloads.

The reference ships only a toy sample; throughput is therefore measured on synthetic batches that
have exactly the layout ``BaseModel.Dataset.collate_batch`` (models/BaseModel.py:121-142) emits,
already narrowed to the ABI dtypes (int32 ids, fp32 scores + their float64 copy for the diversity
term, item-history intents as indices).  Generation is seeded and runs on the target device.
"""
import argparse
import types

import numpy as np
import torch

# name -> (model flags, corpus sizes, batch shape)
WORKLOADS = {
    # BASELINE.json configs[1]/[2]: Tmall-shape, list=50, K=3 base rankers, every embedding 64-d
    'tmall': dict(flags=dict(model_num=3, context_emb_size=64, i_emb_size=64, u_emb_size=64, s_emb_size=64,
                             im_emb_size=64, intent_emb_size=64, cross_attn_qsize=64, num_heads=1, num_layers=1,
                             encoder='BERT4Rec', history_max=20),
                  corpus=dict(items=1000000, users=100000, classes=357, ctx=931, I=30),
                  batch=dict(L=50, H=20)),
    # the same corpus with the PUBLISHED IntEL-BPR hyper-parameters (script/IntEL.sh:15 of the reference): GRU4Rec encoders, 2 heads,
    # 2 tied layers, 16/16/32/32-wide embeddings, context 64, intent 32, cal_diversity 1 (alpha 1e-5), batch 512, lr 1e-4, l2 1e-4
    'tmall_pub': dict(flags=dict(model_num=3, context_emb_size=64, i_emb_size=16, u_emb_size=32, s_emb_size=32, im_emb_size=16,
                                 intent_emb_size=32, cross_attn_qsize=32, num_heads=2, num_layers=2, encoder='GRU4Rec', history_max=20,
                                 cal_diversity=1, diversity_alpha=1e-5, intent_weight=0.01),
                      corpus=dict(items=1000000, users=100000, classes=357, ctx=931, I=30),
                      batch=dict(L=50, H=20), bench_batch=512, optim=(1e-4, 1e-4)),
    # ... and the PUBLISHED IntEL-MSE hyper-parameters (script/IntEL.sh:9, the paper's best Tmall model): the model's default widths
    # (16/16/32/32, context 16, intent 16, 1 head x 1 layer, BERT4Rec encoders), dropout 0.5, cal_diversity 1 (alpha 1e-5), batch 512, lr 1e-3, l2 1e-6
    'tmall_pub_mse': dict(flags=dict(model_num=3, context_emb_size=16, i_emb_size=16, u_emb_size=32, s_emb_size=32, im_emb_size=16,
                                     intent_emb_size=16, cross_attn_qsize=32, num_heads=1, num_layers=1, encoder='BERT4Rec', history_max=20,
                                     cal_diversity=1, diversity_alpha=1e-5, intent_weight=0.003, dropout=0.5),
                          corpus=dict(items=1000000, users=100000, classes=357, ctx=931, I=30),
                          batch=dict(L=50, H=20), bench_batch=512, optim=(1e-3, 1e-6)),
    # configs[3]: LifeData-shape (K=5, 10 intents, list=100)
    'lifedata': dict(flags=dict(model_num=5, context_emb_size=64, i_emb_size=64, u_emb_size=64, s_emb_size=64,
                                im_emb_size=64, intent_emb_size=64, cross_attn_qsize=64, num_heads=1, num_layers=1,
                                encoder='BERT4Rec', history_max=20),
                     corpus=dict(items=1000000, users=100000, classes=357, ctx=931, I=10),
                     batch=dict(L=100, H=20)),
    # configs[4]: stress (10M items, session len 200, list 200, K=8)
    'stress': dict(flags=dict(model_num=8, context_emb_size=64, i_emb_size=64, u_emb_size=64, s_emb_size=64,
                              im_emb_size=64, intent_emb_size=64, cross_attn_qsize=64, num_heads=1, num_layers=1,
                              encoder='BERT4Rec', history_max=200),
                   corpus=dict(items=10000000, users=100000, classes=357, ctx=931, I=32),
                   batch=dict(L=200, H=200)),
    # the bundled toy Tmall sample's shape (BASELINE.json configs[0]; SURVEY.md §8: 266 341 items, 5 147 users,
    # 357 classes, 931 contexts, I = 3 behaviours x 357 classes = 1071, lists up to 90) with the reference's default flags
    'toyshape': dict(flags=dict(model_num=3, context_emb_size=16, i_emb_size=16, u_emb_size=32, s_emb_size=32,
                                im_emb_size=16, intent_emb_size=16, cross_attn_qsize=32, num_heads=1, num_layers=1,
                                encoder='BERT4Rec', history_max=20),
                     corpus=dict(items=266341, users=5147, classes=357, ctx=931, I=1071),
                     batch=dict(L=90, H=20)),
    # tiny shape for smoke tests
    'tiny': dict(flags=dict(model_num=3, context_emb_size=16, i_emb_size=16, u_emb_size=32, s_emb_size=32,
                            im_emb_size=16, intent_emb_size=16, cross_attn_qsize=32, num_heads=1, num_layers=1,
                            encoder='BERT4Rec', history_max=20),
                 corpus=dict(items=5000, users=500, classes=60, ctx=100, I=30),
                 batch=dict(L=50, H=20)),
}

DEFAULT_FLAGS = dict(model_path='', buffer=1, dropout=0, cross_attention=1, intent_weight=0.1, ensemble_weight=1,
                     kl_temp=2, kl_weight=0.5, cal_diversity=0, diversity_alpha=0.01)


def make_args(workload, device, **over):
    w = WORKLOADS[workload]
    d = dict(DEFAULT_FLAGS)
    d.update(w['flags'])
    d.update(over)
    ns = argparse.Namespace(**d)
    ns.device = device
    return ns


def make_corpus(workload, **over):
    c = dict(WORKLOADS[workload]['corpus'])
    c.update(over)
    return types.SimpleNamespace(itemfnum=[c['classes']], contextfnum=[c['ctx']], zero_int=np.zeros(c['I']),
                                 max_uid=c['users'] - 1, max_iid=c['items'] - 1), c


def make_batch(workload, B, device, seed=0, zipf=False, ragged=False, corpus_over=None, scores64=True):
    """One synthetic batch dict (device tensors).  ``ragged`` draws session_len in [L/4, L]."""
    w = WORKLOADS[workload]
    c = dict(w['corpus'])
    c.update(corpus_over or {})
    K, I = w['flags']['model_num'], c['I']
    Lmax, H = w['batch']['L'], w['batch']['H']
    g = torch.Generator(device=device)
    g.manual_seed(1234567 + seed)

    def randint(lo, hi, shape):
        return torch.randint(lo, hi, shape, generator=g, device=device, dtype=torch.int64)
    if ragged:
        slen = randint(max(1, Lmax // 4), Lmax + 1, (B,))
        slen[0] = Lmax
    else:
        slen = torch.full((B,), Lmax, dtype=torch.int64, device=device)
    valid = torch.arange(Lmax, device=device)[None, :] < slen[:, None]
    if zipf:   # Zipf(1.05)-like popularity via inverse-CDF on a power law
        u = torch.rand(B, Lmax, generator=g, device=device, dtype=torch.float64)
        ids = (torch.pow(u, -1.0 / 0.05).clamp(max=float(c['items'] - 1))).long().clamp(1, c['items'] - 1)
    else:
        ids = randint(1, c['items'], (B, Lmax))
    ids = ids * valid
    cls = randint(0, c['classes'], (B, Lmax)) * valid
    raw = torch.rand(B, Lmax, K, generator=g, device=device, dtype=torch.float64)
    big = torch.where(valid[:, :, None], raw, torch.full_like(raw, float('inf')))
    small = torch.where(valid[:, :, None], raw, torch.full_like(raw, float('-inf')))
    mn, mx = big.min(dim=1, keepdim=True)[0], small.max(dim=1, keepdim=True)[0]
    scores = ((raw - mn) / (mx - mn + 1e-6)) * valid[:, :, None]            # BaseModel.py:172-173, pads 0
    # ranking: [3]x1, [2]x1, [1]x3, rest 0, randomly placed among the valid positions
    key = torch.rand(B, Lmax, generator=g, device=device) + (~valid) * 2.0
    order = key.argsort(dim=1)
    ranking = torch.zeros(B, Lmax, dtype=torch.int64, device=device)
    labels = torch.tensor([3, 2, 1, 1, 1], device=device)
    ranking.scatter_(1, order[:, :5], labels[None, :].expand(B, 5))
    ranking = ranking * valid
    hl = randint(1, H + 1, (B,))
    hil = randint(1, H + 1, (B,))
    hv = torch.arange(H, device=device)[None, :] < hl[:, None]
    hiv = torch.arange(H, device=device)[None, :] < hil[:, None]
    his_intents = torch.softmax(torch.rand(B, H, I, generator=g, device=device), dim=-1) * hv[:, :, None]
    intents = torch.softmax(torch.rand(B, I, generator=g, device=device, dtype=torch.float64) * 2, dim=-1)
    batch = {
        'u_id_c': randint(0, c['users'], (B,)).int(),
        'context_mh': randint(0, c['ctx'], (B,)).int(),
        'session_len': slen.int(), 'history_len': hl.int(), 'history_item_len': hil.int(),
        'i_id_s': ids.int(), 'i_class_c': cls.int(), 'ranking': ranking.int(),
        'scores': scores if scores64 else scores.float(),
        'intents': intents,
        'his_intents': his_intents.float(),
        'his_context_mh': (randint(0, c['ctx'], (B, H)) * hv).int(),
        'his_item_id': (randint(1, c['items'], (B, H)) * hiv).int(),
        'his_item_idx': torch.where(hiv, randint(0, I, (B, H)), torch.full((B, H), -1, device=device)).int(),
        'batch_size': B, 'phase': 'train',
    }
    if int(hl.min()) >= 1 and int(hil.min()) >= 1:
        # totals of the valid history rows, known to the producer on the host (model.prepare_batch: packed encoders; histories of >= 1 event)
        batch['his_rows'], batch['hisitem_rows'] = int(hl.sum()), int(hil.sum())
    return batch


def to_reference_layout(batch, I):
    """The same batch in the reference's own layout (int64 ids, dense one-hot item-history intents,
    float64 scores / his_intents) -- what the CPU oracle and the reference consume."""
    out = {}
    for k, v in batch.items():
        if not torch.is_tensor(v):
            out[k] = v
        elif v.dtype == torch.int32:
            out[k] = v.long().cpu()
        else:
            out[k] = v.cpu()
    idx = out.pop('his_item_idx')
    oh = torch.zeros(idx.shape[0], idx.shape[1], I, dtype=torch.float64)
    m = idx >= 0
    oh[m.nonzero(as_tuple=True) + (idx[m],)] = 1.0
    out['his_item_int'] = oh
    out['his_intents'] = out['his_intents'].double()
    out['scores'] = out['scores'].double()
    return out
