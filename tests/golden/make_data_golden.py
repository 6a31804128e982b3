#!/usr/bin/env python3
"""F8 fixture: input-pipeline semantics.  Writes a tiny dataset in the reference's on-disk format
(tests/golden/minidata/: our own synthetic sessions, NOT reference data) and records what the REFERENCE's
SeqReader + IntEL.Dataset + collate_batch produce from it (tests/golden/data_feed.npz).  Build container only."""
import argparse
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, BASE_ARGS  # noqa: E402

MINI = os.path.join(HERE, 'minidata')


def write_minidata(seed=3):
    rng = np.random.default_rng(seed)
    os.makedirs(MINI, exist_ok=True)
    n_items, n_class, n_users = 60, 7, 6
    items = {str(i): {'i_cat_c': int(rng.integers(1, 9)), 'i_seller_c': int(rng.integers(1, 9)), 'i_brand_c': int(rng.integers(1, 9)),
                      'i_class_c': int(rng.integers(0, n_class))} for i in range(1, n_items + 1)}
    users = {str(u): {'u_age_c': int(rng.integers(0, 5)), 'u_gender_c': int(rng.integers(0, 3))} for u in range(1, n_users + 1)}
    json.dump(items, open(os.path.join(MINI, 'item_metadata.json'), 'w'))
    json.dump(users, open(os.path.join(MINI, 'user_metadata.json'), 'w'))
    rows = {'train': [], 'dev': [], 'test': []}
    cid = 100
    intents = {}
    for u in range(1, n_users + 1):
        n_sess = int(rng.integers(3, 9)) if u != 2 else 26          # user 2: history longer than history_max
        times = np.sort(rng.choice(np.arange(1, 40), size=n_sess, replace=False))
        for k, t in enumerate(times):
            n = int(rng.integers(6, 15))
            ids = rng.choice(np.arange(1, n_items + 1), size=n, replace=False).tolist()
            pay, fav, clk = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(1, 4))
            npos = pay + fav + clk
            neg = n - npos - (2 if k % 4 == 0 else 0)                # some sessions have an unlabelled tail
            row = {'u_id_c': u, 'c_time_i': int(t)}
            for s in ('c_pCTR_s', 'c_pCVR_s', 'c_pFVR_s'):
                row[s] = str([round(float(x), 6) for x in rng.normal(size=n) * 4])
            row.update({'i_id_s': str(ids), 'c_paynum_i': pay, 'c_favnum_i': fav, 'c_clicknum_i': clk, 'c_trueneg_i': max(neg, 0),
                        'pos_num': npos, 'c_id_c': cid})
            v = rng.random(3 * n_class)
            v[rng.random(3 * n_class) < 0.5] = 0.0
            if v.sum() == 0:
                v[0] = 1.0
            intents[str(cid)] = (v / v.sum()).tolist()
            phase = 'train' if k < n_sess - 2 else ('dev' if k == n_sess - 2 else 'test')
            rows[phase].append(row)
            cid += 1
    cols = ['u_id_c', 'c_time_i', 'c_pCTR_s', 'c_pCVR_s', 'c_pFVR_s', 'i_id_s', 'c_paynum_i', 'c_favnum_i', 'c_clicknum_i', 'c_trueneg_i',
            'pos_num', 'c_id_c']
    import pandas as pd
    for p, r in rows.items():
        df = pd.DataFrame(r, columns=cols).sample(frac=1.0, random_state=1)     # files are not pre-sorted
        df.to_csv(os.path.join(MINI, p + '.csv'), sep='\t', index=False)
    json.dump(intents, open(os.path.join(MINI, 'intents_multi.json'), 'w'))


def main():
    install_shims()
    write_minidata()
    import torch
    from helpers.SeqReader import SeqReader
    from models.IntEL.IntEL import IntEL
    d = dict(BASE_ARGS)
    d.update(datapath=os.path.join(HERE), dataset='minidata', sep='\t', intent_note='_multi', max_session_len=10, model_num=3,
             history_max=20)
    args = argparse.Namespace(**d)
    args.device = torch.device('cpu')
    corpus = SeqReader(args)
    torch.manual_seed(0)
    model = IntEL(args, corpus)
    out = {'cfg': np.array(json.dumps({k: v for k, v in d.items()}))}
    out['corpus/contextfnum'] = np.array(corpus.contextfnum)
    out['corpus/itemfnum'] = np.array(corpus.itemfnum)
    out['corpus/userfnum'] = np.array(corpus.userfnum)
    out['corpus/max_ids'] = np.array([corpus.max_uid, corpus.max_iid])
    out['corpus/intent_num'] = np.array(len(corpus.zero_int))
    for phase in ('train', 'dev', 'test'):
        out['corpus/%s/position' % phase] = np.asarray(corpus.interactions[phase]['position'])
        out['corpus/%s/item_position' % phase] = np.asarray(corpus.interactions[phase]['item_position'])
        out['corpus/%s/c_id_c' % phase] = np.asarray(corpus.interactions[phase]['c_id_c'])
    np.random.seed(1234)
    for phase in ('train', 'dev', 'test'):
        ds = IntEL.Dataset(model, corpus, phase)
        ds.prepare()
        fds = [ds[i] for i in range(len(ds))]
        out['%s/n' % phase] = np.array(len(fds))
        for i, fd in enumerate(fds):
            for k, v in fd.items():
                out['%s/%d/%s' % (phase, i, k)] = np.asarray(v)
        batch = ds.collate_batch(fds[:5])
        for k, v in batch.items():
            if torch.is_tensor(v):
                out['%s/batch/%s' % (phase, k)] = v.numpy()
    path = os.path.join(HERE, 'data_feed.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
    main()
