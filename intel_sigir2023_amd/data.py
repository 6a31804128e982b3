"""Input pipeline -- host-side counterpart of the reference's readers and ``Dataset`` classes
(SURVEY.md §8-a16 / §8-f1): ``helpers/BaseReader.py``, ``helpers/SeqReader.py`` and the
``_get_feed_dict`` chain ``GeneralModel -> GeneralShuffleModel -> GeneralSeq -> IntEL``
(``models/BaseModel.py:158-197``, ``models/GeneralSeq.py:35-54``, ``models/IntEL/IntEL.py:220-239``)
plus ``collate_batch`` (``models/BaseModel.py:121-142``).

Same on-disk formats (tab-separated CSV with Python-literal list columns, JSON metadata / intents), same
corpus attributes, same per-sample semantics (per-list min-max score normalisation, ranking labels, the
per-access random permutation drawn with the identical ``np.random.choice`` call, history cuts, the
behaviour-code inconsistency of the item-history intent index) and the same collated batch layout, so
that a batch produced here is interchangeable with the reference's (pinned by tests/golden/data_feed.npz).
The organisation differs: corpus tables are built once into flat Python/numpy structures and a sample is
assembled by one function instead of a four-level class chain.
"""
import ast
import gc
import json
import logging
import os

import numpy as np
import pandas as pd
import torch
from torch.nn.utils.rnn import pad_sequence


def _literal(x):
    return ast.literal_eval(str(x))


class SeqReader(object):
    """Corpus = BaseReader.__init__ (BaseReader.py:26-44) + SeqReader._append_his_info (SeqReader.py:22-60)."""

    @staticmethod
    def parse_data_args(parser):
        parser.add_argument('--datapath', type=str, default='../data/', help='Input data dir.')
        parser.add_argument('--dataset', type=str, default='basedata', help='Choose a dataset.')
        parser.add_argument('--sep', type=str, default='\t', help='sep of csv file.')
        parser.add_argument('--intent_note', type=str, default='')
        parser.add_argument('--max_session_len', type=int, default=40)
        return parser

    cfeatures = ['c_time_i']
    ifeatures = ['i_class_c']
    ufeatures = ['u_age_c', 'u_gender_c']
    pos_types = ['c_paynum_i', 'c_favnum_i', 'c_clicknum_i']       # sorted by the pre-defined ranking
    basic_scores = ['c_pCTR_s', 'c_pCVR_s', 'c_pFVR_s']

    def __init__(self, args):
        self.sep = args.sep
        self.prefix = args.datapath
        self.dataset = args.dataset
        self.max_session_len = args.max_session_len
        root = os.path.join(self.prefix, self.dataset)
        frames = {}
        ctx_values = [set([0]) for _ in self.cfeatures]
        max_uid = max_iid = 0
        for phase in ('train', 'dev', 'test'):
            df = pd.read_csv(os.path.join(root, phase + '.csv'), sep=self.sep)
            df.sort_values(by=['u_id_c', 'c_time_i'], inplace=True)
            df.reset_index(drop=True, inplace=True)
            max_uid = max(max_uid, df['u_id_c'].max())
            for i, f in enumerate(self.cfeatures):
                ctx_values[i] |= set(df[f].unique())
            lens = []
            for ids in df['i_id_s'].tolist():
                ids = _literal(ids)
                max_iid = max(max_iid, max(ids))
                lens.append(len(ids))
            df['session_len'] = lens
            frames[phase] = df
        self.contextfnum = [max(len(c), max(c) + 1) for c in ctx_values]
        self.max_uid, self.max_iid = max_uid, max_iid
        all_df = pd.concat([frames[p] for p in ('train', 'dev', 'test')], ignore_index=True)
        self._read_meta(root)
        self._read_intent(root, args.intent_note)
        self._histories(all_df, frames)
        self.interactions = {p: self._to_dict(frames[p], self.max_session_len if p == 'train' else -1) for p in frames}

    def _read_meta(self, root):
        items = json.load(open(os.path.join(root, 'item_metadata.json')))
        self.itemmeta = {}
        seen = [set([0]) for _ in self.ifeatures]
        for key, rec in items.items():
            self.itemmeta[int(key)] = np.array([rec[f] for f in self.ifeatures]).astype(int)
            for i, f in enumerate(self.ifeatures):
                seen[i].add(rec[f] - 1)                     # sic (BaseReader.py:90)
        self.itemfnum = [max(len(f), max(f) + 1) for f in seen]
        users = json.load(open(os.path.join(root, 'user_metadata.json')))
        self.usermeta = {}
        seen = [set([0]) for _ in self.ufeatures]
        for key, rec in users.items():
            self.usermeta[int(key)] = np.array([rec[f] for f in self.ufeatures]).astype(int)
            for i, f in enumerate(self.ufeatures):
                seen[i].add(rec[f])
        self.userfnum = [max(len(f), max(f) + 1) for f in seen]

    def _read_intent(self, root, note):
        raw = json.load(open(os.path.join(root, 'intents%s.json' % note)))
        self.intents = {}
        n = 0
        for key, vec in raw.items():
            self.intents[int(_literal(key))] = np.array(vec)
            n = len(vec)
        self.zero_int = np.zeros(n)
        self.intentloss_w = np.ones(n) / n

    def _histories(self, all_df, frames):
        """Per-user chronological session / positive-item histories and each session's position in them."""
        df = all_df.sort_values(by=['c_time_i', 'u_id_c'], kind='mergesort')
        cols = {c: df[c].tolist() for c in ['c_id_c', 'u_id_c', 'c_clicknum_i', 'c_paynum_i', 'c_favnum_i'] + self.cfeatures}
        item_lists = df['i_id_s'].tolist()
        self.user_his, self.user_itemhis, self.user_itemsession, self.user_itembehave = {}, {}, {}, {}
        position, item_position = [], []
        for i in range(len(item_lists)):
            uid, cid = cols['u_id_c'][i], cols['c_id_c'][i]
            click, pay, fav = cols['c_clicknum_i'][i], cols['c_paynum_i'][i], cols['c_favnum_i'][i]
            feats = [cols[f][i] for f in self.cfeatures]
            positives = _literal(item_lists[i])[:click + pay + fav]
            if uid not in self.user_his:
                self.user_his[uid], self.user_itemhis[uid] = [], []
                self.user_itemsession[uid], self.user_itembehave[uid] = [], []
            position.append(len(self.user_his[uid]))
            item_position.append(len(self.user_itemhis[uid]))
            self.user_his[uid].append([cid] + feats)
            self.user_itemhis[uid] += positives
            self.user_itemsession[uid] += [[cid] + feats] * len(positives)
            self.user_itembehave[uid] += [0] * click + [1] * fav + [2] * pay     # behaviour codes (SeqReader.py:50)
        pos_df = df[['u_id_c', 'c_time_i', 'c_id_c']].copy()
        pos_df['position'] = position
        pos_df['item_position'] = item_position
        for p in frames:
            frames[p] = pd.merge(left=frames[p], right=pos_df, how='left', on=['u_id_c', 'c_time_i', 'c_id_c'])

    @staticmethod
    def _to_dict(df, max_len):
        """utils.df2dict (utils/utils.py:15-30)."""
        out = df.to_dict('list')
        for key in out:
            if key.endswith('_s'):
                out[key] = [_literal(x) if max_len == -1 else _literal(x)[:max_len] for x in out[key]]
            else:
                out[key] = np.array(out[key])
                if key.endswith('_c'):
                    out[key] = out[key].astype(int)
        return out


class Dataset(torch.utils.data.Dataset):
    """``IntEL.Dataset`` (models/IntEL/IntEL.py:219-239 and its three parents)."""

    def __init__(self, model, corpus, phase):
        self.model, self.corpus, self.phase = model, corpus, phase
        self.data = corpus.interactions[phase]
        self.buffer_dict = {}

    def __len__(self):
        for key in self.data:
            return len(self.data[key])
        return 0

    def __getitem__(self, index):
        if self.model.buffer and self.phase != 'train':
            return self.buffer_dict[index]
        return self._get_feed_dict(index)

    def actions_before_epoch(self):
        pass

    def prepare(self):
        """BaseModel.py:106-114: dev/test samples (and their permutation) are built once."""
        if self.model.buffer and self.phase != 'train':
            for i in range(len(self)):
                self.buffer_dict[i] = self._get_feed_dict(i)
            for key in ['i_id_s'] + self.corpus.basic_scores:
                self.data.pop(key)
            gc.collect()

    def _get_feed_dict(self, index):
        c, d = self.corpus, self.data
        uid = d['u_id_c'][index]
        fd = {}
        for key in ['u_id_c', 'c_id_c'] + c.pos_types:
            fd[key] = d[key][index]
        ctx = 0
        for i, key in enumerate(c.cfeatures):
            ctx = ctx * c.contextfnum[i] + d[key][index]
        usr = 0
        for i, key in enumerate(c.ufeatures):
            usr = usr * c.userfnum[i] + c.usermeta[uid][i]
        fd['context_mh'], fd['user_mh'] = ctx, usr
        items = d['i_id_s'][index]
        for i, key in enumerate(c.ifeatures):
            fd[key] = np.array([c.itemmeta[iid][i] for iid in items])
        fd['i_id_s'] = np.array(items)
        for key in c.basic_scores:                       # per-list min-max normalisation (BaseModel.py:172-173)
            x = np.array(d[key][index])
            fd[key] = (x - x.min()) / (x.max() - x.min() + 1e-6)
        n = len(fd['i_id_s'])
        fd['session_len'] = n
        fd['intents'] = c.intents.get(fd['c_id_c'], c.zero_int)
        top = len(c.pos_types)
        labels = []
        for t, key in enumerate(c.pos_types):
            labels += [top - t] * fd[key]
        labels += [0] * d['c_trueneg_i'][index]
        labels = np.array(labels + [-1] * (n - len(labels)))
        fd['ranking'] = labels[:n] if len(labels) > n else labels
        # GeneralShuffleModel: a fresh permutation of the candidate list on every access (BaseModel.py:194-196)
        perm = np.random.choice(np.arange(n), n, replace=False).astype(int)
        for key in ['i_id_s', 'ranking'] + c.basic_scores + c.ifeatures:
            fd[key] = fd[key][perm]
        # GeneralSeq: session history (GeneralSeq.py:38-53)
        position = d['position'][index]
        max_his = self.model.max_his
        if position:
            hist = c.user_his[uid][:position]
            if max_his > 0:
                hist = hist[-max_his:]
            fd['his_intents'] = np.array([c.intents[h[0]] for h in hist])
            mh = [0] * len(hist)
            for i, _ in enumerate(c.cfeatures):
                mh = [mh[k] * c.contextfnum[i] + hist[k][i + 1] for k in range(len(hist))]
            fd['his_context_mh'] = np.array(mh)
        else:
            fd['his_intents'] = np.zeros([1, self.model.intent_num])
            fd['his_context_mh'] = np.array([0])
        fd['position'] = position
        fd['history_len'] = len(fd['his_context_mh'])
        fd['intentloss_w'] = c.intentloss_w
        # IntEL: positive-item history with intent index behaviour*I/K + class (IntEL.py:222-237)
        item_position = d['item_position'][index]
        I, K = self.model.intent_num, self.model.model_num
        if item_position:
            h_items = c.user_itemhis[uid][:item_position]
            h_beh = c.user_itembehave[uid][:item_position]
            h_int = [h_beh[k] * I / K + c.itemmeta[h_items[k]][0] for k in range(len(h_beh))]
            if max_his > 0:
                h_items, h_int = h_items[-max_his:], h_int[-max_his:]
            fd['his_item_id'] = np.array(h_items)
            onehot = np.zeros([len(h_int), I])
            for k, v in enumerate(h_int):
                onehot[k, int(v)] = 1
            fd['his_item_int'] = onehot
        else:
            fd['his_item_id'] = np.array([0])
            fd['his_item_int'] = np.zeros([1, I])
        fd['history_item_len'] = len(fd['his_item_id'])
        return fd

    def collate_batch(self, feed_dicts):
        """BaseModel.py:121-142: ragged arrays -> pad_sequence(0); base scores stacked on the last axis."""
        out = {}
        for key in feed_dicts[0]:
            vals = [d[key] for d in feed_dicts]
            if isinstance(vals[0], np.ndarray) and any(len(v) != len(vals[0]) for v in vals):
                out[key] = pad_sequence([torch.from_numpy(v) for v in vals], batch_first=True)
            else:
                out[key] = torch.from_numpy(np.array(vals))
        out['scores'] = torch.stack([out[k] for k in self.corpus.basic_scores], dim=2)
        for k in self.corpus.basic_scores:
            out.pop(k)
        out['batch_size'] = len(feed_dicts)
        out['phase'] = self.phase
        # host-side totals of the valid history rows (not in the reference's dict): model.prepare_batch runs the BERT4Rec
        # encoders on those rows only when they are present
        if 'history_len' in out and 'history_item_len' in out and int(out['history_len'].min()) >= 1 and int(out['history_item_len'].min()) >= 1:
            out['his_rows'] = int(out['history_len'].sum())
            out['hisitem_rows'] = int(out['history_item_len'].sum())
        return out


def standin_intents(root, sep='\t', n_behaviors=3):
    """Deterministic stand-in for the missing ``intents_multi.json`` of the toy sample:
    per session, the normalised histogram over ``behaviour*C + i_class_c`` of its positive items, where the
    positives are the first pay+fav+click entries of ``i_id_s`` (in that order) and C = number of classes."""
    items = json.load(open(os.path.join(root, 'item_metadata.json')))
    C = max(rec['i_class_c'] for rec in items.values()) + 1
    out = {}
    for phase in ('train', 'dev', 'test'):
        df = pd.read_csv(os.path.join(root, phase + '.csv'), sep=sep)
        for _, row in df.iterrows():
            ids = _literal(row['i_id_s'])
            pay, fav, clk = int(row['c_paynum_i']), int(row['c_favnum_i']), int(row['c_clicknum_i'])
            beh = [2] * pay + [1] * fav + [0] * clk
            h = np.zeros(n_behaviors * C)
            for b, iid in zip(beh, ids[:len(beh)]):
                h[b * C + items[str(iid)]['i_class_c']] += 1
            if h.sum() > 0:
                h = h / h.sum()
            out[str(int(row['c_id_c']))] = h.tolist()
    return out
