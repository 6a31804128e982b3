"""Device-side input feed: a columnar, HBM-resident form of the corpus that
``data.SeqReader`` reads (helpers/BaseReader.py + helpers/SeqReader.py) and batch assembly by one HIP kernel
(``intel_feed_collate``) instead of the per-sample Python of ``Dataset._get_feed_dict`` + ``collate_batch``
(models/BaseModel.py:121-197, models/GeneralSeq.py:35-54, models/IntEL/IntEL.py:220-239).

``ColumnarStore.collate(indices, ...)`` returns a batch dict in the layout ``IntEL.prepare_batch`` takes
(int32 ids, fp32 scores / intents, ``his_item_idx`` instead of the dense one-hot) plus the labels the losses
read.  Parity with the per-sample path is bit-exact when the same permutations are supplied
(``shuffle='host'``; tests/test_feed_gpu.py); ``shuffle='device'`` draws them on the GPU.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


class ColumnarStore(object):
    """Flat arrays of one phase of the corpus.  Host copies (numpy) are kept for the shape maxima."""

    SESSION_KEYS = ('u_id', 'context_mh', 'n_pay', 'n_fav', 'n_click', 'n_trueneg', 'position', 'item_position', 'intent_row')

    def __init__(self, corpus, phase, model_num, intent_num, max_his):
        d = corpus.interactions[phase]
        self.phase = phase
        self.n_sessions = len(d['u_id_c'])
        self.n_scores = len(corpus.basic_scores)
        self.intent_num, self.model_num, self.max_his = int(intent_num), int(model_num), int(max_his)
        h = {}
        h['u_id'] = _i32(d['u_id_c'])
        ctx = np.zeros(self.n_sessions, dtype=np.int64)
        for i, key in enumerate(corpus.cfeatures):                       # BaseModel.py:163-165
            ctx = ctx * corpus.contextfnum[i] + np.asarray(d[key], dtype=np.int64)
        h['context_mh'] = _i32(ctx)
        h['n_pay'], h['n_fav'], h['n_click'] = _i32(d['c_paynum_i']), _i32(d['c_favnum_i']), _i32(d['c_clicknum_i'])
        h['n_trueneg'] = _i32(d['c_trueneg_i'])
        h['position'], h['item_position'] = _i32(d['position']), _i32(d['item_position'])
        # intent rows: row 0 = zeros (corpus.zero_int), then one row per session id with a label
        cids = sorted(corpus.intents.keys())
        row_of = {cid: r + 1 for r, cid in enumerate(cids)}
        rows = np.zeros((len(cids) + 1, self.intent_num), dtype=np.float64)
        for cid, r in row_of.items():
            rows[r] = corpus.intents[cid]
        h['intent_rows'] = np.ascontiguousarray(rows.astype(np.float32))   # the rounding the model's .float() applies
        h['intent_row'] = _i32([row_of.get(int(c), 0) for c in d['c_id_c']])
        # candidate lists (already cut at max_session_len for the training phase, utils.df2dict)
        lens = np.array([len(x) for x in d['i_id_s']], dtype=np.int64)
        h['list_off'] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        flat_ids = np.concatenate([np.asarray(x, dtype=np.int64) for x in d['i_id_s']]) if self.n_sessions else np.zeros(0, np.int64)
        h['item_id'] = _i32(flat_ids)
        h['item_class'] = _i32([corpus.itemmeta[int(i)][0] for i in flat_ids])
        sc = np.zeros((len(flat_ids), self.n_scores), dtype=np.float64)
        for k, key in enumerate(corpus.basic_scores):
            sc[:, k] = np.concatenate([np.asarray(x, dtype=np.float64) for x in d[key]]) if self.n_sessions else 0
        h['scores'] = np.ascontiguousarray(sc)
        # per-user chronological histories (SeqReader.py:22-60), CSR over u_id
        n_users = int(max(max(corpus.user_his.keys(), default=0), corpus.max_uid)) + 1
        self.n_users = n_users
        uh_off, uh_ctx, uh_row = [0], [], []
        ui_off, ui_id, ui_idx = [0], [], []
        I, K = self.intent_num, self.model_num
        for uid in range(n_users):
            for rec in corpus.user_his.get(uid, []):
                mh = 0
                for i, _ in enumerate(corpus.cfeatures):                 # GeneralSeq.py:43-46
                    mh = mh * corpus.contextfnum[i] + rec[i + 1]
                uh_ctx.append(mh)
                if rec[0] not in row_of:
                    raise KeyError('session %s of user %d has no intent label (the reference indexes corpus.intents directly, '
                                   'GeneralSeq.py:42)' % (rec[0], uid))
                uh_row.append(row_of[rec[0]])
            uh_off.append(len(uh_ctx))
            items, beh = corpus.user_itemhis.get(uid, []), corpus.user_itembehave.get(uid, [])
            ui_id += list(items)
            ui_idx += [int(beh[k] * I / K + corpus.itemmeta[items[k]][0]) for k in range(len(items))]   # IntEL.py:226, 232
            ui_off.append(len(ui_id))
        h['uhis_off'], h['uitem_off'] = np.asarray(uh_off, dtype=np.int64), np.asarray(ui_off, dtype=np.int64)
        h['uhis_context_mh'], h['uhis_intent_row'] = _i32(uh_ctx), _i32(uh_row)
        h['uitem_id'], h['uitem_intent_idx'] = _i32(ui_id), _i32(ui_idx)
        self.host = h
        self.list_len = lens
        self.dev = None
        self._struct = None

    @classmethod
    def from_arrays(cls, host, phase, model_num, intent_num, max_his):
        """Store over ready-made columnar arrays (keys = the pointer fields of IntelFeedStore + 'intent_rows')."""
        self = cls.__new__(cls)
        self.phase = phase
        self.intent_num, self.model_num, self.max_his = int(intent_num), int(model_num), int(max_his)
        self.host = {k: np.ascontiguousarray(v) for k, v in host.items()}
        self.n_sessions = len(self.host['u_id'])
        self.n_scores = self.host['scores'].shape[1]
        self.n_users = len(self.host['uhis_off']) - 1
        self.list_len = np.diff(self.host['list_off']).astype(np.int64)
        self.dev = None
        self._struct = None
        return self

    @classmethod
    def synthetic(cls, n_sessions, L, K, I, H, n_items, n_users, n_classes, n_ctx, seed=0):
        """Random corpus of fixed-length lists (the shape of bench.py's workloads) for feed throughput measurements."""
        r = np.random.RandomState(seed)
        h = {}
        h['u_id'] = r.randint(0, n_users, n_sessions).astype(np.int32)
        h['context_mh'] = r.randint(0, n_ctx, n_sessions).astype(np.int32)
        h['n_pay'] = np.ones(n_sessions, np.int32)
        h['n_fav'] = np.ones(n_sessions, np.int32)
        h['n_click'] = np.full(n_sessions, 3, np.int32)
        h['n_trueneg'] = np.full(n_sessions, L - 5, np.int32)
        h['position'] = r.randint(1, H + 1, n_sessions).astype(np.int32)
        h['item_position'] = r.randint(1, H + 1, n_sessions).astype(np.int32)
        rows = r.rand(4097, I).astype(np.float32)
        rows[0] = 0
        h['intent_rows'] = rows / np.maximum(rows.sum(1, keepdims=True), 1e-9)
        h['intent_row'] = r.randint(1, 4097, n_sessions).astype(np.int32)
        h['list_off'] = (np.arange(n_sessions + 1, dtype=np.int64) * L)
        h['item_id'] = r.randint(0, n_items, n_sessions * L).astype(np.int32)
        h['item_class'] = r.randint(0, n_classes, n_sessions * L).astype(np.int32)
        h['scores'] = r.rand(n_sessions * L, K)
        h['uhis_off'] = (np.arange(n_users + 1, dtype=np.int64) * H)
        h['uhis_context_mh'] = r.randint(0, n_ctx, n_users * H).astype(np.int32)
        h['uhis_intent_row'] = r.randint(1, 4097, n_users * H).astype(np.int32)
        h['uitem_off'] = (np.arange(n_users + 1, dtype=np.int64) * H)
        h['uitem_id'] = r.randint(0, n_items, n_users * H).astype(np.int32)
        h['uitem_intent_idx'] = r.randint(0, I, n_users * H).astype(np.int32)
        return cls.from_arrays(h, 'train', K, I, H)

    # ---- host-side shape logic (what pad_sequence would produce) -------------------------------------
    def history_lens(self, idx):
        p, ip = self.host['position'][idx].astype(np.int64), self.host['item_position'][idx].astype(np.int64)
        if self.max_his > 0:
            p, ip = np.minimum(p, self.max_his), np.minimum(ip, self.max_his)
        return np.maximum(p, 1), np.maximum(ip, 1)

    def batch_shape(self, idx):
        hl, hil = self.history_lens(idx)
        return int(self.list_len[idx].max()), int(hl.max()), int(hil.max())

    # ---- device side ------------------------------------------------------------------------------------
    def to(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise L.IntelHipError('ColumnarStore lives in HBM: device must be a GPU (got %s)' % device)
        self.dev = {k: torch.from_numpy(v).to(device) for k, v in self.host.items()}
        st = L.IntelFeedStore(n_sessions=self.n_sessions, n_users=self.n_users, n_scores=self.n_scores, intent_num=self.intent_num,
                              max_his=self.max_his, n_intent_rows=self.host['intent_rows'].shape[0])
        for k, _ in L.IntelFeedStore._fields_[6:]:
            setattr(st, k, self.dev[k].data_ptr())
        self._struct = st
        self.device = device
        return self

    def nbytes(self):
        return int(sum(v.nbytes for v in self.host.values()))

    def max_shape(self):
        """(L, H, Hi) large enough for any batch of this store."""
        return self.batch_shape(np.arange(self.n_sessions))

    def collate(self, idx, shuffle='device', perm=None, seed=0, shape=None):
        """idx: session indices (array-like, or an int32 device tensor).  shuffle: 'none' | 'device' | 'host' (perm: per-session
        permutations, slot i <- stored position perm[b][i]).  shape: (L, H, Hi) to pad to -- at least the batch maxima;
        default = the batch maxima (what pad_sequence gives), which needs the indices on the host."""
        if self._struct is None:
            raise L.IntelHipError('ColumnarStore.to(device) first')
        dev = self.device
        if torch.is_tensor(idx):
            idx_dev = idx.to(device=dev, dtype=torch.int32).contiguous()
            idx_host = None if shape is not None else idx.detach().cpu().numpy().astype(np.int64)
            B = int(idx.numel())
        else:
            idx_host = np.asarray(idx, dtype=np.int64)
            idx_dev = torch.from_numpy(idx_host.astype(np.int32)).to(dev)
            B = len(idx_host)
        Lm, H, Hi = shape if shape is not None else self.batch_shape(idx_host)
        I, K = self.intent_num, self.n_scores
        mode = {'none': 0, 'device': 1, 'host': 2}[shuffle]
        perm_dev = None
        if mode == 2:
            pm = np.zeros((B, Lm), dtype=np.int32)
            for b in range(B):
                pb = np.asarray(perm[b], dtype=np.int32)
                pm[b, :len(pb)] = pb
            perm_dev = torch.from_numpy(pm).to(dev)
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        out = {
            'i_id_s': torch.empty(B, Lm, **i32), 'i_class_c': torch.empty(B, Lm, **i32), 'scores': torch.empty(B, Lm, K, **f32),
            'ranking': torch.empty(B, Lm, **i32), 'session_len': torch.empty(B, **i32), 'u_id_c': torch.empty(B, **i32),
            'context_mh': torch.empty(B, **i32), 'intents': torch.empty(B, I, **f32), 'his_context_mh': torch.empty(B, H, **i32),
            'his_intents': torch.empty(B, H, I, **f32), 'history_len': torch.empty(B, **i32), 'his_item_id': torch.empty(B, Hi, **i32),
            'his_item_idx': torch.empty(B, Hi, **i32), 'history_item_len': torch.empty(B, **i32),
        }
        fo = L.IntelFeedOut(B=B, L=Lm, H=H, Hi=Hi)
        for k, v in out.items():
            setattr(fo, k, v.data_ptr())
        L.check(L.lib().intel_feed_collate(C.byref(self._struct), L.ptr(idx_dev), mode, L.ptr(perm_dev), C.c_ulonglong(int(seed) & (2 ** 64 - 1)),
                                           C.byref(fo), L.stream_ptr(dev)), 'intel_feed_collate')
        out['batch_size'] = B
        out['phase'] = self.phase
        if idx_host is not None:          # host-known totals of the valid history rows (model.prepare_batch: packed encoders)
            hl, hil = self.history_lens(idx_host)
            if int(hl.min()) >= 1 and int(hil.min()) >= 1:      # packed rows need histories of >= 1 event (include/intel_hip.h); the reference's Dataset guarantees it
                out['his_rows'], out['hisitem_rows'] = int(hl.sum()), int(hil.sum())
        out['_keep'] = (idx_dev, perm_dev)          # inputs of the asynchronous launch
        return out


def epoch_batches(store, batch_size, epoch=0, seed=0, shuffle_sessions=True, shuffle_lists='device', drop_last=False, rank=0, world=1,
                  keep_all=False):
    """Batches of one pass over the store: the DataLoader(shuffle=True) + per-access list shuffle of the reference's
    training loop (helpers/BaseRunner.py:275-277, models/BaseModel.py:194-196), assembled on the device.
    Data parallel (world > 1): ``batch_size`` is the GLOBAL batch; rank r assembles sessions [r*B/world, (r+1)*B/world) of
    every global batch, padded to the GLOBAL batch's shape (pad rows are keys) -- the session order, the list
    permutations (keyed by the session's corpus index) and the padding are those of the single-process run.  A global batch
    must split evenly (each rank's loss is a mean over its shard): a ragged last batch is trimmed to a multiple of world.
    keep_all (evaluation sets): NO session is dropped -- a global batch that does not split evenly gives the first ranks one
    session more, and a rank left without a session gets a placeholder (the batch's first session) with weight 0.  Every batch
    then carries 'eval_weight' ([B] float64: 1 per real session) for the weighted reductions of runner.evaluate."""
    n = store.n_sessions
    order = np.random.RandomState(seed * 1000003 + epoch).permutation(n) if shuffle_sessions else np.arange(n)
    for lo in range(0, n, batch_size):
        idx = order[lo:lo + batch_size]
        if drop_last and len(idx) < batch_size:
            break
        shape = None
        weight = None
        if world > 1 and keep_all:
            shape = store.batch_shape(idx)
            cuts = [(len(idx) * r) // world for r in range(world + 1)]
            mine = idx[cuts[rank]:cuts[rank + 1]]
            weight = np.ones(max(len(mine), 1), dtype=np.float64)
            if len(mine) == 0:
                mine, weight[0] = idx[:1], 0.0
            idx = mine
        elif world > 1:
            keep = len(idx) - len(idx) % world
            if keep == 0:
                break
            idx = idx[:keep]
            shape = store.batch_shape(idx)
            per = keep // world
            idx = idx[rank * per:(rank + 1) * per]
        out = store.collate(idx, shuffle=shuffle_lists, seed=(seed << 20) + epoch * 65537 + lo, shape=shape)
        if keep_all:
            import torch
            w = weight if weight is not None else np.ones(len(idx), dtype=np.float64)
            out['eval_weight'] = torch.from_numpy(w).to(out['session_len'].device)
        yield out
