"""Train / evaluate loop -- host-side counterpart of the reference's ``helpers/BaseRunner.py``.

Same CLI flags (``parse_runner_args``), same loop semantics (epoch loop, dev-metric model selection
with ``stop_tol = 1e-4``, early stop, best-model reload, NaN check) and a bit-for-bit restatement of
``evaluate_method`` (the source of the NDCG@3 numbers).  Two ways to drive the HIP path:
  * ``use_engine=False``: exactly the reference's hot loop -- ``model(batch)``, ``criterion(...)``,
    ``loss.backward()``, ``torch.optim`` step (helpers/BaseRunner.py:279-290);
  * ``use_engine=True`` (default): ``IntELEngine.train_step`` (flat buffers + fused Adam + DP all-reduce).
"""
import gc
import logging
import os
from time import time

import numpy as np
import torch

from . import parallel
from .engine import IntELEngine


def format_metric(result_dict):
    """utils/utils.py:64-89."""
    keys = sorted(result_dict.keys(), key=lambda k: (int(k.split('@')[1]) if '@' in k else 0, k.split('@')[0]))
    out = []
    for k in keys:
        m = result_dict[k]
        if isinstance(m, (float, np.floating)):
            out.append('{}:{:<.4f}'.format(k, m))
        elif isinstance(m, (int, np.integer)):
            out.append('{}:{}'.format(k, m))
    return ','.join(out)


class BaseRunner(object):
    @staticmethod
    def parse_runner_args(parser):
        """helpers/BaseRunner.py:22-54."""
        parser.add_argument('--epoch', type=int, default=200, help='Number of epochs.')
        parser.add_argument('--test_epoch', type=int, default=-1, help='Print test results every test_epoch (-1 means no print).')
        parser.add_argument('--early_stop', type=int, default=10, help='The number of epochs when dev results drop continuously.')
        parser.add_argument('--lr', type=float, default=1e-3, help='Learning rate.')
        parser.add_argument('--l2', type=float, default=0, help='Weight decay in optimizer.')
        parser.add_argument('--intent_l2', type=float, default=1e-6, help='Parsed and ignored, like the reference (BaseRunner.py:160).')
        parser.add_argument('--batch_size', type=int, default=256, help='Batch size during training.')
        parser.add_argument('--eval_batch_size', type=int, default=100, help='Batch size during testing.')
        parser.add_argument('--optimizer', type=str, default='Adam', help='optimizer: SGD, Adam, Adagrad, Adadelta')
        parser.add_argument('--num_workers', type=int, default=4, help='Number of processors when prepare batches in DataLoader')
        parser.add_argument('--pin_memory', type=int, default=0, help='pin_memory in DataLoader')
        parser.add_argument('--topk', type=str, default='1,3,5', help='The number of items recommended to each user.')
        parser.add_argument('--metrics', type=str, default='NDCG,HR', help='metrics: NDCG, HR')
        parser.add_argument('--main_metric', type=str, default='NDCG@1', help='main metric')
        parser.add_argument('--test_ensemble', type=int, default=1)
        parser.add_argument('--decay_lr', type=float, default=0)
        parser.add_argument('--decay_step', type=int, default=1)
        parser.add_argument('--device_metrics', type=int, default=1,
                            help='1: HR@k / NDCG@k of evaluate_method on the device; 0: predictions to the host, numpy (reference flow)')
        return parser

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def evaluate_method(prediction_scores, ranking_lists, pos_nums, topk, metrics, session_len, show_num=False):
        """helpers/BaseRunner.py:56-131.  Returns the same 25-key dict on the same inputs:
        per-behaviour HR@k / NDCG@k with binary relevance over the label-sorted prefix, and the overall
        ``NDCG@k`` with LINEAR gains; predictions are padded with 0 and labels with -2 (-> 0) up to
        ``max(max(session_len), max(topk))`` (a padded slot therefore outranks a negative score)."""
        count = min(len(session_len), len(prediction_scores))
        session_len = np.asarray(session_len)[:count]
        pos_nums = {k: np.asarray(v)[:count] for k, v in pos_nums.items()}
        width = int(max(int(session_len.max()), max(topk)))
        preds = np.zeros((count, width), dtype=np.float64)
        labels = np.full((count, width), -2, dtype=np.int64)
        for i in range(count):
            n = min(int(session_len[i]), len(prediction_scores[i]))
            preds[i, :n] = np.asarray(prediction_scores[i])[:n]
            n2 = min(int(session_len[i]), len(ranking_lists[i]))
            labels[i, :n2] = np.asarray(ranking_lists[i])[:n2]
        rows = np.arange(count).reshape(-1, 1)
        by_label = np.argsort(labels, axis=1)[:, ::-1]
        labels, preds = labels[rows, by_label], preds[rows, by_label]
        labels[labels < 0] = 0
        rank_pos = preds.argsort(axis=1)                      # ascending: best prediction is the last column
        discounts = 1.0 / np.log2(np.arange(width) + 2.0)
        evaluations = dict()
        total = np.sum(np.array(list(pos_nums.values())), axis=0).reshape(-1, 1)
        for btype, pos_num in pos_nums.items():
            behavior = btype.split('_')[1].split('num')[0]
            all_pos = total if 'click' in btype else pos_num.reshape(-1, 1)
            hits = rank_pos < all_pos
            keep = np.flatnonzero(all_pos[:, 0] > 0)
            hits, all_pos_k = hits[keep], all_pos[keep]
            if show_num:
                logging.info('# %s session: %d' % (behavior, len(keep)))
            for k in topk:
                kk = min(k, width)
                for metric in metrics:
                    key = '{}_{}@{}'.format(behavior, metric, k)
                    if metric == 'HR':
                        evaluations[key] = (hits[:, -kk:].sum(axis=1) > 0).mean()
                    elif metric == 'NDCG':
                        if k == 1:
                            continue                          # NDCG@1 == HR@1
                        dcg = (hits[:, -kk:] * discounts[:kk][::-1]).sum(axis=1)
                        idcg = ((np.arange(kk).reshape(1, -1) < all_pos_k) * discounts[:kk]).sum(axis=1)
                        evaluations[key] = (dcg / idcg).mean()
                    else:
                        raise ValueError('Undefined evaluation metric: {}.'.format(metric))
        best_first = np.argsort(preds, axis=1)[:, ::-1]
        gains = labels[rows, best_first]
        ideal = np.sort(labels, axis=1)[:, ::-1]
        for k in topk:
            dcg = (gains[:, :k] * discounts[:k]).sum(axis=1)
            idcg = (ideal[:, :k] * discounts[:k]).sum(axis=1)
            evaluations['NDCG@%d' % k] = (dcg / idcg).mean()
        return evaluations

    # ---- the same 25 keys on the device ------------------------------------------------------------
    BEHAVIOURS = ('pay', 'fav', 'click')

    @staticmethod
    def label_positions(ranking_lists, session_len, width):
        """Slot of every item in the reference's label-descending pre-sort (helpers/BaseRunner.py:66-81): the SAME numpy
        call on the same padded label matrix, so the order among equal labels is the reference's by construction.  It
        depends on the labels only -- computed once per evaluation set, then every evaluation runs on the device
        (intel_eval_metrics).  Returns int32 [n, width]; row i holds the slots of items 0..session_len[i]-1."""
        n = len(session_len)
        labels = np.full((n, width), -2, dtype=np.int64)
        for i in range(n):
            m = min(int(session_len[i]), len(ranking_lists[i]))
            labels[i, :m] = np.asarray(ranking_lists[i])[:m]
        order = np.argsort(labels, axis=1)[:, ::-1]              # order[i, slot] = item
        pos = np.empty((n, width), dtype=np.int32)
        pos[np.arange(n).reshape(-1, 1), order] = np.arange(width, dtype=np.int32)
        return pos

    @staticmethod
    def evaluate_method_device(ens_score, ranking, session_len, topk, metrics, width=0, pos_nums=None, label_pos=None):
        """Per-session values of every evaluate_method key for one padded batch, on the device.
        ens_score [B,L] f32, ranking [B,L] i32, session_len [B] i32, pos_nums [B,3] i32 or None, label_pos [B,L] i32 or None.
        Returns (values [B, 7*len(topk)] f64, valid [B,3] u8): see BaseRunner.reduce_device_metrics."""
        import ctypes as C
        from . import _lib as L
        B, Lm = ens_score.shape
        nk = len(topk)
        out = torch.empty(B, 7 * nk, dtype=torch.float64, device=ens_score.device)
        valid = torch.empty(B, 3, dtype=torch.uint8, device=ens_score.device)
        tk = (C.c_int * nk)(*[int(k) for k in topk])
        L.check(L.lib().intel_eval_metrics(B, Lm, int(width), nk, tk, L.ptr(ens_score.contiguous()), L.ptr(ranking.contiguous()),
                                           L.ptr(session_len.contiguous()), L.ptr(pos_nums), L.ptr(label_pos), L.ptr(out), L.ptr(valid),
                                           L.stream_ptr(ens_score.device)), 'intel_eval_metrics')
        return out, valid

    @staticmethod
    def reduce_device_metrics(sums, counts, n_sessions, topk, metrics):
        """sums [7*nk] = column sums of the per-session values (behaviour keys over their valid sessions only), counts [3] =
        valid sessions per behaviour -> the evaluate_method dict (same keys; NDCG@1 of a behaviour is omitted like the
        reference does, helpers/BaseRunner.py:107-108)."""
        nk = len(topk)
        res = dict()
        for t, beh in enumerate(BaseRunner.BEHAVIOURS):
            for ki, k in enumerate(topk):
                for metric in metrics:
                    if metric == 'HR':
                        res['%s_HR@%d' % (beh, k)] = float(sums[(t * nk + ki) * 2]) / float(counts[t]) if counts[t] else float('nan')
                    elif metric == 'NDCG':
                        if k == 1:
                            continue
                        res['%s_NDCG@%d' % (beh, k)] = float(sums[(t * nk + ki) * 2 + 1]) / float(counts[t]) if counts[t] else float('nan')
                    else:
                        raise ValueError('Undefined evaluation metric: {}.'.format(metric))
        for ki, k in enumerate(topk):
            res['NDCG@%d' % k] = float(sums[6 * nk + ki]) / float(n_sessions)
        return res

    @staticmethod
    def evaluate_intents(true_intents, predict_intents, topk=[1, 5, 10, 30]):
        """helpers/BaseRunner.py:133-150."""
        true_intents, predict_intents = np.asarray(true_intents), np.asarray(predict_intents)
        evaluations = dict()
        true_labels = np.argmax(true_intents, axis=1).reshape(-1, 1)
        asc = np.argsort(predict_intents, axis=1)
        desc = asc[:, ::-1]
        rows = np.arange(len(predict_intents)).reshape(-1, 1)
        true_sort = true_intents[rows, desc]
        true_perfect = np.sort(true_intents, axis=1)[:, ::-1]
        discounts = 1 / np.log2(np.arange(40) + 2.0)
        for k in topk:
            dcg = (true_sort[:, :k] * discounts[:k]).sum(axis=1)
            idcg = (true_perfect[:, :k] * discounts[:k]).sum(axis=1)
            evaluations['Int-NDCG@%d' % k] = (dcg / idcg).mean()
            evaluations['Int-HR@%d' % k] = ((asc == true_labels)[:, -k:].sum(axis=-1) > 0).mean()
        return evaluations

    # ------------------------------------------------------------------------------------------
    def __init__(self, args, use_engine=True):
        self.epoch = args.epoch
        self.test_epoch = args.test_epoch
        self.early_stop = args.early_stop
        self.learning_rate = args.lr
        self.batch_size = args.batch_size
        self.eval_batch_size = args.eval_batch_size
        self.l2 = args.l2
        self.intent_l2 = args.l2           # sic: BaseRunner.py:160
        self.optimizer_name = args.optimizer
        self.topk = [int(x) for x in args.topk.split(',')]
        self.metrics = [m.strip().upper() for m in args.metrics.split(',')]
        self.main_metric = args.main_metric
        self.test_ensemble = args.test_ensemble
        self.decay_lr = args.decay_lr
        self.decay_step = args.decay_step
        self.stop_tol = 1e-4
        self.use_engine = use_engine and self.optimizer_name == 'Adam'
        self.args = args
        self.engine = None
        self.time = None
        self.device_metrics = bool(getattr(args, 'device_metrics', 1))      # evaluate_method on the device (intel_eval_metrics)
        self._eval_sets = {}

    def _check_time(self, start=False):
        if self.time is None or start:
            self.time = [time()] * 2
            return self.time[0]
        tmp = self.time[1]
        self.time[1] = time()
        return self.time[1] - tmp

    def _build_optimizer(self, model):
        """helpers/BaseRunner.py:182-188."""
        logging.info('Optimizer: ' + self.optimizer_name)
        opt_cls = getattr(torch.optim, self.optimizer_name)
        optimizer = opt_cls(model.customize_parameters({'intent_l2': self.intent_l2, 'ens_l2': self.l2}),
                            lr=self.learning_rate, weight_decay=self.l2)
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=self.decay_step, gamma=self.decay_lr)
        return optimizer, scheduler

    def step_lr(self, model, epochs_done):
        """StepLR(step_size=decay_step, gamma=decay_lr) stepped once per epoch (helpers/BaseRunner.py:187,238-241): the
        torch scheduler on the autograd path, the same schedule in closed form on the engine path."""
        if self.use_engine and self.engine is not None:
            lr = self.learning_rate * self.decay_lr ** (epochs_done // self.decay_step)
            self.engine.set_lr(lr)
            return lr
        model.scheduler.step()
        return model.scheduler.get_last_lr()[0]

    def eval_termination(self, criterion):
        return len(criterion) - criterion.index(max(criterion)) > self.early_stop

    # ---- one epoch over an iterable of batch dicts (helpers/BaseRunner.py:268-291) ------------------
    def fit(self, model, batches, criterion, loss_name=None):
        model.train()
        losses = []
        if self.use_engine:
            if self.engine is None:
                # lazy table Adam (engine.py): evaluation catches its batches' rows up inside the forward pass, state_dict()
                # (save_model) and load_state_dict() settle the whole table through the engine's hooks
                self.engine = IntELEngine(model, loss_name or type(criterion).__name__, self.args, lr=self.learning_rate, l2=self.l2,
                                          lazy_table={'0': False, '1': True}.get(os.environ.get('INTEL_ADAM_LAZY', 'auto'), 'auto'))
            self.engine.defer_table_wait = True      # back-to-back steps of one epoch: the next forward starts under the table's sweep (engine.py);
            try:                                     # everything after the epoch reads the model through the engine's hooks / flush()
                for batch in batches:
                    loss, _, _ = self.engine.train_step(batch)
                    losses.append(loss.detach())
            finally:
                self.engine.defer_table_wait = False
                self.engine.flush()
        else:
            if model.optimizer is None:
                model.optimizer, model.scheduler = self._build_optimizer(model)
            for batch in batches:
                model.optimizer.zero_grad()
                out = model(batch)
                loss, _, _ = criterion(out, batch)
                loss.backward()
                model.optimizer.step()
                losses.append(loss.detach())
        vals = torch.stack([l.double().reshape(()) for l in losses])
        if parallel.world_size() > 1:      # each rank holds the mean loss of its (equal) shard of every global batch
            vals = parallel.allgather(vals).mean(0)
        return float(np.mean(vals.cpu().numpy()))

    # ---- predict / evaluate (helpers/BaseRunner.py:293-355) -----------------------------------------
    @torch.no_grad()
    def predict(self, model, batches, criterion):
        model.eval()
        preds, ranks, losses, true_int, pred_int, slens = [], [], [], [], [], []
        counts = []
        for batch in batches:
            out = model(batch)
            loss, _, _ = criterion(out, batch)
            # 'eval_weight' (feed.epoch_batches(keep_all=True)): 0 marks a placeholder session of a rank whose shard of a ragged
            # global batch is empty -- it is evaluated (the kernels need a session) and dropped here
            w = batch.get('eval_weight')
            keep = np.ones(int(batch['session_len'].shape[0]), dtype=bool) if w is None else (w.cpu().numpy() > 0)
            losses.append(float(loss))
            counts.append(int(keep.sum()))
            sel = lambda a: [x for x, k in zip(a, keep) if k]
            preds.extend(sel(out['ens_score'].cpu().numpy()))
            ranks.extend(sel(batch['ranking'].cpu().numpy()))
            slens.extend(sel(batch['session_len'].cpu().numpy().tolist()))
            true_int.extend(sel(batch['intents'].cpu().numpy()))
            pred_int.extend(sel(out['intents'].cpu().numpy()))
        if parallel.world_size() > 1:
            # data parallel: every rank evaluated its shard of each batch; the metrics are means over ALL sessions, so the
            # per-session records are gathered (rank order = session order) and every rank computes the same numbers
            parts = parallel.allgather_object((preds, losses, ranks, true_int, pred_int, slens, counts))
            preds, ranks, true_int, pred_int, slens = ([x for p in parts for x in p[i]] for i in (0, 2, 3, 4, 5))
            # a batch's loss = mean over its sessions: the shard means weighted by the shard sizes (uneven for a ragged batch)
            losses = [float(sum(p[1][j] * p[6][j] for p in parts) / max(1, sum(p[6][j] for p in parts))) for j in range(len(losses))]
        return preds, float(np.mean(losses)), ranks, true_int, pred_int, slens

    def _eval_set(self, batches, topk):
        """Per-evaluation-set constants of the device metrics, computed once per (set, largest cutoff) -- dev / test batches are
        fixed lists: the reference's max_len = max(longest list of the WHOLE set, largest cutoff PASSED, helpers/BaseRunner.py:66)
        and every batch's label-sort slots at that width."""
        key = (id(batches), max(topk))
        hit = self._eval_sets.get(key)
        if hit is not None and hit[0] is batches:
            return hit[1], hit[2]
        slens = [b['session_len'].cpu().numpy() for b in batches]
        local_max = max(int(s.max()) for s in slens) if slens else 0
        width = max(parallel.global_max_([local_max], batches[0]['session_len'].device)[0], max(topk))
        lps = []
        for b, sl in zip(batches, slens):
            r = b['ranking'].cpu().numpy()
            Lb = r.shape[1]
            lp = self.label_positions([r[i] for i in range(len(sl))], np.minimum(sl, Lb), width)[:, :Lb]
            lps.append(torch.from_numpy(np.ascontiguousarray(lp)).to(b['ranking'].device))
        self._eval_sets[key] = (batches, width, lps)
        return width, lps

    @torch.no_grad()
    def evaluate_on_device(self, model, batches, topk, metrics, criterion, topk_intent=[1, 5, 10, 30]):
        """evaluate() without the per-batch .cpu().numpy() round trip of the predictions (helpers/BaseRunner.py:338-343):
        forward, criterion and every evaluate_method key per batch on the device (intel_eval_metrics); only the column sums
        come back.  Data parallel: sums and counts are all-reduced, so every rank reports the global numbers."""
        model.eval()
        width, lps = self._eval_set(batches, topk)
        dev = batches[0]['session_len'].device
        nk = len(topk)
        sums = torch.zeros(7 * nk, dtype=torch.float64, device=dev)
        counts = torch.zeros(4, dtype=torch.float64, device=dev)      # valid sessions per behaviour, all sessions
        losses, true_int, pred_int, nsess = [], [], [], []
        for batch, lp in zip(batches, lps):
            out = model(batch)
            loss, _, _ = criterion(out, batch)
            losses.append(loss.detach().double().reshape(()))
            rk = batch['ranking'] if batch['ranking'].dtype == torch.int32 else batch['ranking'].to(torch.int32)
            sl = batch['session_len'] if batch['session_len'].dtype == torch.int32 else batch['session_len'].to(torch.int32)
            vals, valid = self.evaluate_method_device(out['ens_score'], rk, sl, topk, metrics, width=width, label_pos=lp)
            ew = batch.get('eval_weight')                 # 0: placeholder session of an empty shard (feed.epoch_batches(keep_all=True))
            ew = torch.ones(vals.shape[0], dtype=torch.float64, device=dev) if ew is None else ew.to(dev).double()
            w = ew[:, None].expand_as(vals).clone()
            for t in range(3):
                w[:, t * nk * 2:(t + 1) * nk * 2] *= valid[:, t:t + 1].double()
            sums += torch.where(w > 0, vals, torch.zeros_like(vals)).sum(0)
            counts[:3] += (valid.double() * ew[:, None]).sum(0)
            counts[3] += ew.sum()
            nsess.append(ew.sum().reshape(()))
            keep = ew > 0
            true_int.append(batch['intents'][keep])
            pred_int.append(out['intents'][keep])
        lossv = torch.stack(losses)
        if parallel.world_size() > 1:
            parallel.allreduce_sum_([sums, counts])
            # a batch's loss = mean over its sessions: the shard means weighted by the shard sizes
            nv = torch.stack(nsess)
            num, den = lossv * nv, nv.clone()
            parallel.allreduce_sum_([num, den])
            lossv = num / den.clamp_min(1.0)
        res = dict()
        if self.test_ensemble:
            res.update(self.reduce_device_metrics(sums.cpu().numpy(), counts.cpu().numpy()[:3], float(counts[3]), topk, metrics))
        ti = torch.cat(true_int).cpu().numpy()
        pi = torch.cat(pred_int).cpu().numpy()
        if parallel.world_size() > 1:
            parts = parallel.allgather_object((ti, pi))
            ti, pi = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        if len(ti):
            res.update(self.evaluate_intents(ti, pi, topk=[k for k in topk_intent if k <= ti.shape[1]]))
        return float(lossv.mean().cpu()), res

    def evaluate(self, model, batches, topk, metrics, criterion, pos_nums=None, topk_intent=[1, 5, 10, 30]):
        on_device = bool(self.device_metrics and pos_nums is None and isinstance(batches, list) and len(batches) and batches[0]['session_len'].is_cuda
                         and len(topk) <= 8 and max(topk) <= 64 and max(b['i_id_s'].shape[1] for b in batches) <= 512)
        if parallel.world_size() > 1:
            # the two paths use different collectives: every rank must take the same one.  EVERY rank enters this vote -- whatever its own
            # batches look like (a generator, an empty list, host batches: all of that is folded into its `on_device`) -- on a device that
            # does not depend on the batches
            dev = getattr(model, 'device', None)
            if dev is None or torch.device(dev).type != 'cuda':
                dev = torch.device('cuda', torch.cuda.current_device()) if (torch.cuda.is_available() and torch.distributed.get_backend() == 'nccl') else torch.device('cpu')
            on_device = parallel.global_max_([0 if on_device else 1], dev)[0] == 0
        if on_device:
            return self.evaluate_on_device(model, batches, topk, metrics, criterion, topk_intent)
        preds, loss, ranks, true_int, pred_int, slens = self.predict(model, batches, criterion)
        res = dict()
        if self.test_ensemble:
            if pos_nums is None:          # derive the per-behaviour positive counts from the labels
                R = [np.asarray(r)[:n] for r, n in zip(ranks, slens)]
                pos_nums = {'c_paynum_i': np.array([(r == 3).sum() for r in R]),
                            'c_favnum_i': np.array([(r == 2).sum() for r in R]),
                            'c_clicknum_i': np.array([(r == 1).sum() for r in R])}
            res.update(self.evaluate_method(preds, ranks, pos_nums, topk, metrics, np.asarray(slens)))
        if len(true_int):
            res.update(self.evaluate_intents(true_int, pred_int, topk=[k for k in topk_intent if k <= len(true_int[0])]))
        gc.collect()
        return loss, res

    def train(self, model, data, criterion, loss_name=None):
        """helpers/BaseRunner.py:190-266.  ``data`` = {'train': callable -> iterable of batches per epoch,
        'dev': list of batches, 'test': list of batches}."""
        main_results, dev_results = [], []
        self.train_losses = []
        self._check_time(start=True)
        for epoch in range(self.epoch):
            self._check_time()
            loss = self.fit(model, data['train'](epoch), criterion, loss_name)
            self.train_losses.append(loss)
            if np.isnan(loss):
                raise ValueError('Loss is nan!')
            train_t = self._check_time()
            dev_loss, dev_res = self.evaluate(model, data['dev'], self.topk[:1], self.metrics, criterion, topk_intent=[3, 5])
            dev_results.append(dev_res)
            main_results.append(dev_res[self.main_metric])
            msg = 'Epoch {:<5} loss={:<.4f} [{:<3.1f} s]\tdev loss={:<.4f}, ({})'.format(epoch + 1, loss, train_t, dev_loss, format_metric(dev_res))
            if self.test_epoch > 0 and epoch % self.test_epoch == 0 and data.get('test') is not None:      # BaseRunner.py:225-233
                test_loss, test_res = self.evaluate(model, data['test'], self.topk[:1], self.metrics, criterion, topk_intent=[5])
                msg += ' test loss={:<.4f}, ({})'.format(test_loss, format_metric(test_res))
            if self.decay_lr > 0:                                                                           # BaseRunner.py:238-241
                c_lr = self.step_lr(model, epoch + 1)
                logging.info('LR: %.4f' % c_lr)
            if len(main_results) == 1 or max(main_results[:-1]) < main_results[-1] - self.stop_tol:
                if model.model_path and parallel.rank() == 0:
                    model.save_model()
                msg += ' *'
            logging.info(msg)
            if self.early_stop > 0 and self.eval_termination(main_results):
                logging.info('Early stop at %d based on dev result.' % (epoch + 1))
                break
        best = main_results.index(max(main_results))
        logging.info(os.linesep + 'Best Iter(dev)={:>5}\t dev=({}) [{:<.1f} s] '.format(best + 1, format_metric(dev_results[best]), self.time[1] - self.time[0]))
        parallel.barrier()                 # rank 0 has written the best model
        if model.model_path and os.path.exists(model.model_path):
            model.load_model()
        return main_results
