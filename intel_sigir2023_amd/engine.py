"""Training / evaluation engine: the fast path around BaseRunner.fit's hot loop
(helpers/BaseRunner.py:279-290: forward -> criterion -> backward -> optimizer.step).

Compared with driving the module through autograd + torch.optim (which also works and is what the
parity tests exercise), the engine keeps every parameter, gradient and Adam moment in flat buffers:
  * gradients are persistent (no per-step allocation; embedding-table gradients stay dense like
    nn.Embedding(sparse=False) but are re-zeroed inside the Adam sweep that consumes them),
  * data parallelism is one all-reduce per flat bucket (parallel.py),
  * Adam is two launches (decayed bucket, bias bucket) with the reference's param-group semantics
    (models/BaseModel.py:53-62, helpers/BaseRunner.py:182-188).
"""
import ctypes as C

import os

import torch

from . import _lib as L
from . import parallel


class IntELEngine(object):
    def __init__(self, model, loss_name='IntBPRloss', args=None, lr=1e-3, l2=0.0, betas=(0.9, 0.999), eps=1e-8, lazy_table=None):
        self.model = model
        self.loss_name = loss_name
        if loss_name not in ('IntBPRloss', 'IntListloss', 'IntMSEloss', 'BPRloss', 'Listloss', 'MSEloss'):
            raise ValueError('unsupported loss ' + loss_name)
        self.kind = 'bpr' if 'BPR' in loss_name else ('mse' if 'MSE' in loss_name else 'list')
        self.with_intent = loss_name.startswith('Int')
        g = lambda k, d: getattr(args, k, d) if args is not None else d
        self.intent_weight = float(g('intent_weight', 0.1))
        self.ensemble_weight = float(g('ensemble_weight', 1.0))
        self.kl_weight, self.kl_temp = float(g('kl_weight', 0.5)), float(g('kl_temp', 2.0))
        self.cal_diversity, self.alpha = int(g('cal_diversity', 0)), float(g('diversity_alpha', 0.01))
        self.lr, self.l2, self.betas, self.eps = float(lr), float(l2), betas, float(eps)
        self._table_ev = None          # event behind the item-id table's pending Adam sweep (train_step, one-GPU wide schedule)
        # Opt-in for loops that call train_step back to back and touch the model only THROUGH the engine / the model's own methods (bench.py, runner.fit):
        # a step then returns with the item-id table's sweep still running on its side stream and the next forward starts under it.  Code that reads
        # model.iid_embeddings.weight (or eng.m / eng.v) directly on another stream must call eng.flush() first -- hence off by default.
        self.defer_table_wait = os.environ.get('INTEL_DEFER_TABLE', '0') == '1'
        self.step_count = 0
        # BPR tie-breaking noise (BPRloss.py:26) drawn inside the loss kernel: one generator per engine, seeded from torch's
        # CPU generator (reproducible under torch.manual_seed) and -- data parallel -- made COMMON to all ranks once, here;
        # every step then draws the same 64-bit seed on every rank without communication, and the kernel keys its counter by
        # the GLOBAL session index (session0 = rank * B), so N shards draw what one process draws for the whole batch
        seed0 = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
        if parallel.active():
            t = seed0.to(next(model.parameters()).device)
            parallel.broadcast_([t])
            seed0 = t.cpu()
        self._noise_gen = torch.Generator()
        self._noise_gen.manual_seed(int(seed0.item()))
        self._dp_shape = None
        # two-phase backward + table all-reduce / Adam on a side stream (default); INTEL_OVERLAP_TABLE=0 runs the plain order
        self.overlap_table_update = os.environ.get('INTEL_OVERLAP_TABLE', '1') != '0'
        # the one-call backward with its four branches on four streams, the table exchange (data parallel) and sweep underneath
        # its tail; INTEL_BWD_SCHEDULE=phased: the two-call order (table gradient first, its exchange under the second call)
        self.wide_backward = os.environ.get('INTEL_BWD_SCHEDULE', 'wide') != 'phased'
        self._side = None
        self._sort_side = None
        self._sorted_scatter = os.environ.get('INTEL_SCATTER_SORTED', 'auto')      # auto | 1 (always sort) | 0 (unsorted atomics)
        self._dup_share = None
        self._noise_tensor = os.environ.get('INTEL_BPR_NOISE', 'kernel') == 'tensor'     # A/B switch: draw the BPR noise with torch.rand
        self.device = next(model.parameters()).device
        L.require_gpu(next(model.parameters()))
        self._flatten()
        self._bufs = {}
        # readers outside train_step (state_dict / load_state_dict / a plain model(batch) forward) are ordered behind a pending table sweep
        self._sync_hooks = [self.model.register_state_dict_pre_hook(lambda *a, **k: self._table_sync()),
                            self.model.register_load_state_dict_pre_hook(lambda *a, **k: self._table_sync()),
                            self.model.register_forward_pre_hook(lambda *a, **k: self._table_sync())]
        # one byte per item-id gradient row: the backward marks the rows it adds into, the table's Adam sweep reads and
        # clears the gradient only there (the other ~95 % of the rows have g = 0): 6 instead of 8 streams over the table
        self._iid_flags = None
        w = model.iid_embeddings.weight
        if os.environ.get('INTEL_ADAM_ROWS', '1') != '0' and w.shape[1] in (16, 32, 64, 128, 256):
            self._iid_flags = torch.zeros(w.shape[0], dtype=torch.uint8, device=self.device)
            L.check(L.lib().intel_set_iid_grad_row_flags(model._context(), L.ptr(self._iid_flags)), 'intel_set_iid_grad_row_flags')
        # lazy form of the item-id table's dense Adam (include/intel_hip.h: IntelLazyTable): a step updates the rows that carry a
        # gradient, every forward pass first brings the rows it gathers up to date, flush() the rest -- bit-identical to the dense
        # sweep wherever the table is observed THROUGH the engine / the model (forward, state_dict, load_state_dict); code that
        # reads model.iid_embeddings.weight or eng.m / eng.v directly calls eng.flush() first.  Opt-in: lazy_table=True or
        # INTEL_ADAM_LAZY=1 (bench.py and the runner's engine path switch it on)
        # 'auto' (what bench.py and the runner pass): decided at the first step from the batch shape -- lazy when a step touches at
        # most 1/4 of the rows of a table of at least 128 MB (the 10 M-item stress table at batch 256: 1 %, the dense sweep is 2.75 of its 5.7 ms step,
        # 43 k -> 69 k sessions/s; Tmall shape, lazy against dense, round 4: 1536 sessions (11 % of the rows) +7 %, 2048 +5 %, 2560 +3.5 %, 3072 (22 %) +1 %,
        # 3584 +-0, 4096 (29 %) -0.6 %: the sweep hides under the backward's tail there and the lazy form only saves its traffic), dense otherwise
        if lazy_table is None:
            lazy_table = {'0': False, '1': True}.get(os.environ.get('INTEL_ADAM_LAZY', '0'), 'auto')
        self._lazy = None
        self._lazy_auto = lazy_table == 'auto' and self._iid_flags is not None and w.shape[0] > 0
        if lazy_table is True and self._iid_flags is not None and w.shape[0] > 0:
            self._lazy_init()

    # ---- lazy table Adam ---------------------------------------------------------------------------------------------
    LAZY_CAP = 1 << 16          # steps the schedule window holds before the table is flushed and the window moved

    def _lazy_init(self):
        w = self.model.iid_embeddings.weight
        rows, d = w.shape
        self._lazy_last = torch.zeros(rows, dtype=torch.int32, device=self.device)
        self._lazy_sched = torch.zeros(self.LAZY_CAP, 2, dtype=torch.float32, device=self.device)
        b1, b2 = self.betas
        self._lazy = L.IntelLazyTable(p=self.flat['iid'].data_ptr(), m=self.m['iid'].data_ptr(), v=self.v['iid'].data_ptr(),
                                      last=self._lazy_last.data_ptr(), sched=self._lazy_sched.data_ptr(), rows=rows, d=d,
                                      base=self.step_count, cap=self.LAZY_CAP, beta1=b1, beta2=b2, eps=self.eps, weight_decay=self.l2)
        self._lazy_last.fill_(self.step_count)
        self._lazy_publish()
        self._lazy_settled = self.step_count
        # readers outside the engine: state_dict() / load_state_dict() see (and replace) an up-to-date table; the model's own
        # forward (evaluation through model(batch)) settles the table once instead of replaying stale rows in every gather
        self._lazy_hooks = [self.model.register_state_dict_pre_hook(lambda *a, **k: self.flush()),
                            self.model.register_load_state_dict_pre_hook(lambda *a, **k: self.flush()),
                            self.model.register_forward_pre_hook(lambda *a, **k: self.flush())]

    def _lazy_publish(self, settled=False):
        """Tell the context which step the gathers of a forward pass must deliver the table's rows at (settled: every row is
        up to date -- the plain gather will do until the next step)."""
        upto = self._lazy.base if settled else self.step_count
        L.check(L.lib().intel_set_lazy_table(self.model._context(), C.byref(self._lazy), upto), 'intel_set_lazy_table')

    def _table_sync(self):
        """Order the current stream behind the pending sweep of the item-id table (a step's tail left running: train_step).  Everything that reads the
        table or its optimizer state outside the next training forward comes through here."""
        if self._table_ev is not None:
            torch.cuda.current_stream(self.device).wait_event(self._table_ev)
            self._table_ev = None

    def flush(self):
        """Lazy table Adam: every row of the item-id table (and its moments) brought up to the current step.  No-op otherwise."""
        self._table_sync()
        if self._lazy is None or self._lazy_settled == self.step_count:
            return
        L.check(L.lib().intel_adam_lazy_flush(C.byref(self._lazy), self.step_count, L.stream_ptr(self.device)), 'intel_adam_lazy_flush')
        self._lazy_settled = self.step_count
        self._lazy_publish(settled=True)

    def _lazy_step(self, stream_ptr):
        b1, b2 = self.betas
        hyper = (C.c_float(b1).value, C.c_float(b2).value, C.c_float(self.eps).value, C.c_float(self.l2).value)
        changed = hyper != (self._lazy.beta1, self._lazy.beta2, self._lazy.eps, self._lazy.weight_decay)
        if changed or self.step_count - self._lazy.base > self._lazy.cap:
            # window full, or a hyper-parameter other than lr changed (the replay uses ONE set): settle everything, move the window
            self.step_count -= 1
            L.check(L.lib().intel_adam_lazy_flush(C.byref(self._lazy), self.step_count, stream_ptr), 'intel_adam_lazy_flush')
            self.step_count += 1
            self._lazy.base = self.step_count - 1
            self._lazy.beta1, self._lazy.beta2, self._lazy.eps, self._lazy.weight_decay = hyper
        L.check(L.lib().intel_adam_lazy_step(C.byref(self._lazy), L.ptr(self.gflat['iid']), L.ptr(self._iid_flags), self.lr,
                                             self.step_count, stream_ptr), 'intel_adam_lazy_step')
        self._lazy_publish()

    # ---- data-parallel exchange of the item-id table gradient ------------------------------------------------------
    def _sparse_exchange(self, keep, world):
        """Touched-rows all-gather instead of the dense all-reduce?  INTEL_DP_EXCHANGE = auto | dense |
        sparse.  auto: compare the bytes a rank receives: (world-1) * rows * (4 d + 4) against the ring all-reduce's
        2 (world-1)/world * table bytes -- the table wins at Tmall shape on 8 GPUs (75 MB of rows per rank against
        64 MB), the rows win on fewer GPUs and by far for the 10 M-item stress table."""
        import os
        mode = os.environ.get('INTEL_DP_EXCHANGE', 'auto')
        if mode != 'auto':
            return mode == 'sparse'           # 'dense' and 'sharded' (see _sharded_table_update) reduce the whole table
        rows = keep['i_id_s'].numel() + keep['his_item_id'].numel()     # same on every rank: _check_global_shape
        d = self.model.iid_embeddings.weight.shape[1]
        return rows * (4 * d + 4) < 2.0 / world * self.gflat['iid'].numel() * 4

    def _exchange_touched_rows(self, keep, stream_ptr):
        """gflat['iid'] <- sum over ranks, exchanged as (row index, row) pairs of the rows this step touched.  Every rank
        takes its rows out of its table (leaving it all zero) and then adds the buffers of ALL ranks, its own included,
        in rank order: the same summation order everywhere, so the replicas stay bit-identical."""
        lib = L.lib()
        table = self.model.iid_embeddings.weight.grad
        d = table.shape[1]
        idx = self._touched_idx_dev(keep, stream_ptr)
        cap = idx.numel()
        rows = self._buf('xch_rows', (cap, d), torch.float32)
        L.check(lib.intel_rows_take(L.ptr(table), d, L.ptr(idx), cap, L.ptr(rows), 1, stream_ptr), 'intel_rows_take')
        all_idx = parallel.allgather(idx)
        all_rows = parallel.allgather(rows)
        for r in range(all_idx.shape[0]):
            L.check(lib.intel_rows_add(L.ptr(table), d, L.ptr(all_idx[r]), cap, L.ptr(all_rows[r]), stream_ptr), 'intel_rows_add')
        if self._iid_flags is not None:                   # the rows of the other ranks must be visited by the table's Adam sweep too
            L.check(lib.intel_rows_mark(L.ptr(self._iid_flags), L.ptr(all_idx), all_idx.numel(), stream_ptr), 'intel_rows_mark')
        self._bufs['xch_keep'] = (all_idx, all_rows)      # alive until the kernels have run

    def _touched_idx_dev(self, keep, stream_ptr):
        """The rows of the item-id gradient table this rank's backward added into, as a static-shape index list (cap = the batch's id count, the
        same on every rank; -1 padding): compacted ON THE DEVICE from the row marks the embedding scatter left (intel_rows_compact: two small
        launches over the marks, ascending order) -- no sort, no concatenation, nothing sized on the host.  Without row marks (INTEL_ADAM_ROWS=0)
        the list is derived from the batch's ids with torch (sort + first-of-run: _touched_idx)."""
        if self._iid_flags is None:
            return self._touched_idx(keep)
        lib = L.lib()
        cap = keep['i_id_s'].numel() + keep['his_item_id'].numel()
        nrows = self.model.iid_embeddings.weight.shape[0]
        idx = self._buf('xch_idx', (cap,), torch.int32)
        scratch = self._buf('xch_scratch', (int(lib.intel_rows_compact_scratch_ints(nrows)),), torch.int32)
        L.check(lib.intel_rows_compact(L.ptr(self._iid_flags), nrows, L.ptr(idx), cap, L.ptr(scratch), stream_ptr), 'intel_rows_compact')
        return idx

    def _sharded(self):
        """INTEL_DP_EXCHANGE=sharded and no lazy table: the table's optimizer state is partitioned over the ranks."""
        import os
        return os.environ.get('INTEL_DP_EXCHANGE', 'auto') == 'sharded' and self._lazy is None

    def _sharded_table_update(self, stream_ptr):
        """The item-id table's step with its Adam state SHARDED over the ranks (ZeRO-1 for this one bucket): reduce-scatter of the
        table gradient -> every rank runs torch.optim.Adam's dense update on ITS 1/world row range only (parameter, both moments,
        28 B per parameter of HBM traffic instead of world x that) -> all-gather of the updated rows.  Over xGMI the two collectives
        move what the ring all-reduce of the dense exchange moves ((world-1)/world of the table each way); the sweep's HBM traffic
        and the moments' memory are divided by world.  Rows are padded to a multiple of world inside the flat buckets themselves (_flatten: no staging copy); every
        replica ends the step with the same table bits (the owner's arithmetic is the dense sweep's, adam_kernel)."""
        lib = L.lib()
        w, r = parallel.world_size(), parallel.rank()
        rows, d = self.model.iid_embeddings.weight.shape
        per = -(-rows // w)                                   # rows per rank
        n = per * d
        # the table's flat buckets (parameter, gradient, both moments) were allocated with room for per * w rows (_flatten): the collectives work on
        # views of them -- no concatenation, no staging copy.  The padding rows stay zero under Adam (p = g = m = v = 0)
        gfull, pfull = self.gflat['iid'][:w * n], self.flat['iid'][:w * n]
        gs = self._buf('shard_g', (n,), torch.float32)
        parallel.reduce_scatter_sum(gfull, gs)
        b1, b2 = self.betas
        lo, hi = r * n, (r + 1) * n
        L.check(lib.intel_adam_step(L.ptr(self.flat['iid'][lo:hi]), L.ptr(gs), L.ptr(self.m['iid'][lo:hi]), L.ptr(self.v['iid'][lo:hi]), n,
                                    self.lr, b1, b2, self.eps, self.l2, self.step_count, 1.0, 0, stream_ptr), 'intel_adam_step')
        parallel.allgather_inplace(pfull, r, n)               # every rank's updated rows into every replica's table, in place
        gfull.zero_()                                         # the local gradient (every row) and the row marks are consumed
        if self._iid_flags is not None:
            self._iid_flags.zero_()

    @staticmethod
    def _touched_idx(keep):
        """Unique touched item-id rows of the batch at a static shape (torch.unique would synchronise the host to size its
        result, which stalls the enqueueing of the backward): sort, keep the first of every run, -1 elsewhere.  Depends on the
        batch only: computed once per prepared batch, ahead of the forward pass."""
        idx = keep.get('xch_idx')
        if idx is None:
            ids = torch.cat([keep['i_id_s'].reshape(-1), keep['his_item_id'].reshape(-1)])
            srt = ids.sort().values
            prev = torch.empty_like(srt)
            prev[0] = -1
            prev[1:] = srt[:-1]
            idx = keep['xch_idx'] = torch.where(srt != prev, srt, torch.full_like(srt, -1)).to(torch.int32).contiguous()
        return idx

    def _sort_scatter_ids(self, ib, keep):
        """The batch's item / class / history-item ids sorted with their row indices, on a side stream under the forward pass
        (the backward's embedding scatter then sums runs of equal ids in registers before its float atomics: popular items --
        Zipf -- no longer serialise on one address).  Returns the event the backward must wait for (None: switched off)."""
        mode = self._sorted_scatter
        if mode == 'auto':
            # sorting costs ~0.16 ms of side-stream work per step and pays only when ids repeat (Zipf popularity: 6.55 -> 4.68 ms
            # per step; uniform ids: 4.02 -> 4.18 ms): decide from the share of repeated ids in a 16 384-id sample of the batch,
            # measured on the first step and every 256th (one host synchronisation each time)
            if self._dup_share is None or self.step_count % 256 == 0:
                sample = keep['i_id_s'].reshape(-1)[:16384]
                self._dup_share = 1.0 - float(torch.unique(sample).numel()) / float(sample.numel())
            mode = '1' if self._dup_share > 0.2 else '0'
        if mode != '1':
            for key in ('iid', 'hisitem'):
                setattr(ib, key + '_sort_ids', None)
                setattr(ib, key + '_sort_rows', None)
            return None
        dev = self.device
        if self._sort_side is None:
            self._sort_side = torch.cuda.Stream(device=dev)
        side, cur = self._sort_side, torch.cuda.current_stream(dev)
        side.wait_stream(cur)               # also orders the reuse of last step's index buffers after that step's backward
        with torch.cuda.stream(side):
            for key, src in (('iid', keep['i_id_s']), ('hisitem', keep['his_item_id'])):
                v, i = torch.sort(src.reshape(-1))
                keep[key + '_sort_ids'], keep[key + '_sort_rows'] = v, i.to(torch.int32)
                setattr(ib, key + '_sort_ids', v.data_ptr())
                setattr(ib, key + '_sort_rows', keep[key + '_sort_rows'].data_ptr())
            ev = torch.cuda.Event()
            ev.record(side)
        return ev

    def _table_stream(self):
        """The stream of the item-id table's Adam sweep next to the one-call backward's tail: one of the context's own side
        streams (idle by then) -- a stream of our own would be a fifth active one and share a hardware queue."""
        if getattr(self, '_table', None) is None:
            ptr = L.lib().intel_side_stream(self.model._context(), 1)
            self._table = torch.cuda.ExternalStream(ptr, device=self.device) if ptr else self._side_stream()
        return self._table

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _param_cache(self):
        """(slot items, detached parameters, parameter-pointer array, gradient-pointer array), built once: the engine owns the flat buckets
        the parameters and gradients are views of (_flatten), so the pointers are the same every step.  Invalidated explicitly where the
        buckets are rebuilt (_flatten); on the hot path only a cheap key is checked (first / last parameter address, the bucket addresses,
        counts), the full key -- the address of EVERY parameter and gradient tensor, ~400 data_ptr() calls -- every 256th step and whenever the
        cheap one moved."""
        c = getattr(self, '_pcache', None)

        def cheap_key(items):
            return (len(items), items[0][2].data_ptr(), items[-1][2].data_ptr(), len(self.grad_by_slot)) + tuple(t.data_ptr() for t in self.flat.values()) + \
                   tuple(t.data_ptr() for t in self.gflat.values())

        def full_key(items):
            return (tuple(p.data_ptr() for _, _, p in items), tuple((s, g.data_ptr()) for s, g in sorted(self.grad_by_slot.items())))
        if c is not None:
            self._pcache_age = getattr(self, '_pcache_age', 0) + 1
            if c[5] == cheap_key(c[0]) and (self._pcache_age % 256 or c[4] == full_key(c[0])):
                return c[0], c[1], c[2], c[3]
        items = self.model.slot_items()
        params = [p.detach() for _, _, p in items]
        for t in params:
            L.require_gpu(t)
        parr = self.model._param_array({s: t.contiguous() for (s, _, _), t in zip(items, params)})
        garr = self.model._param_array(self.grad_by_slot)
        c = self._pcache = (items, params, parr, garr, full_key(items), cheap_key(items))
        return c[0], c[1], c[2], c[3]

    # ---- flat parameter / gradient / moment buckets -------------------------------------------------
    def _flatten(self):
        self._pcache = None
        items = self.model.slot_items()
        # three flat buckets: the item-id table (its all-reduce is the big one and is overlapped with the second
        # half of the backward pass), every other decayed parameter, the biases (weight_decay 0)
        groups = {'iid': [], 'decay': [], 'nodecay': []}
        for s, name, p in items:
            if name == 'iid_embeddings.weight':
                groups['iid'].append((s, name, p))
            else:
                groups['nodecay' if 'bias' in name else 'decay'].append((s, name, p))
        self.flat, self.gflat, self.m, self.v = {}, {}, {}, {}
        self.grad_by_slot = {}
        for gname, lst in groups.items():
            sizes = [((p.numel() + 63) // 64) * 64 for _, _, p in lst]      # 256-byte aligned slices
            total = sum(sizes)
            if gname == 'iid' and lst and parallel.world_size() > 1:      # room for ceil(rows / world) * world rows: the sharded table update's collectives
                rows_, d_ = lst[0][2].shape                                 # (reduce-scatter, in-place all-gather) then work on views of the buckets
                total = max(total, ((-(-rows_ // parallel.world_size()) * parallel.world_size() * d_ + 63) // 64) * 64)
            flat = torch.zeros(total, dtype=torch.float32, device=self.device)
            gflat = torch.zeros(total, dtype=torch.float32, device=self.device)
            off = 0
            for (s, name, p), sz in zip(lst, sizes):
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat[off:off + n].view_as(p)
                gview = gflat[off:off + n].view_as(p)
                p.grad = gview
                self.grad_by_slot[s] = gview
                off += sz
            self.flat[gname], self.gflat[gname] = flat, gflat
            self.m[gname] = torch.zeros_like(flat)
            self.v[gname] = torch.zeros_like(flat)

    def buckets(self):
        self._table_sync()
        return [self.gflat['iid'], self.gflat['decay'], self.gflat['nodecay']]

    def param_buckets(self):
        self._table_sync()
        return [self.flat['iid'], self.flat['decay'], self.flat['nodecay']]

    def _buf(self, name, shape, dtype):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t

    # ---- one training step ---------------------------------------------------------------------------
    def _check_global_shape(self, ib):
        """Data parallel: every rank must run the step at the SAME (B, L, H, Hi) -- pad rows are keys, the
        losses are means over equal shards, and the touched-rows exchange sizes its all-gather from the shape.  Checked with
        one tiny all-gather when the local shape changes (INTEL_DP_SHAPE_CHECK=always: every step; off: never)."""
        mode = os.environ.get('INTEL_DP_SHAPE_CHECK', 'change')
        shape = (ib.B, ib.L, ib.H, ib.Hi)
        if mode == 'off' or (mode != 'always' and shape == self._dp_shape):
            return
        t = torch.tensor(shape, dtype=torch.int32, device=self.device)
        allt = parallel.allgather(t).cpu().tolist()
        if any(tuple(s) != shape for s in allt):
            raise L.IntelHipError('data-parallel ranks disagree on the padded batch shape (B, L, H, Hi): %s -- pad every '
                                  'batch to the GLOBAL maximum (feed: shape=store.max_shape(); runner: parallel.global_max_)' % allt)
        self._dp_shape = shape

    def set_lr(self, lr):
        """StepLR counterpart of the engine path (helpers/BaseRunner.py:238-241)."""
        self.lr = float(lr)

    def train_step(self, batch, noise=None, noise_seed=None):
        """forward + loss + backward + gradient all-reduce + Adam on one (local) batch.
        Returns (loss, ensemble_loss, intent_loss) as device tensors (no host sync).
        noise: optional [B,L,L] tie-breaking noise of the BPR loss (parity tests pass the reference's draw);
        noise_seed: optional explicit 64-bit seed of the in-kernel draw (default: the engine's generator)."""
        self.model._opt_steps = getattr(self.model, '_opt_steps', 0) + 1      # parameters change behind torch's version counters
        model = self.model
        lib = L.lib()
        dev = self.device
        world = parallel.world_size()
        dp = parallel.active()              # exchange steps on (world > 1, or a forced one-rank group: INTEL_DP_FORCE=1)
        ib, keep = model.prepare_batch(batch)
        if dp:
            self._check_global_shape(ib)
            if self._sparse_exchange(keep, world) and self._iid_flags is None:
                self._touched_idx(keep)      # (torch fallback without row marks: derived from the batch's ids ahead of the forward pass)
        if self._lazy_auto:             # same decision on every rank: the shape is global (_check_global_shape)
            self._lazy_auto = False
            w = model.iid_embeddings.weight       # ... and the table is large enough for its sweep to matter (>= 128 MB: >= 0.15 ms per step)
            if (ib.B * world) * (ib.L + ib.Hi) * 4 <= w.shape[0] and w.numel() * 4 >= (128 << 20):
                self._lazy_init()
        B, Lmax, K, I = ib.B, ib.L, model.model_num, model.intent_num
        items, params, parr, garr = self._param_cache()
        sort_ev = self._sort_scatter_ids(ib, keep)
        if self._table_ev is not None:      # the previous step's table sweep may still run: only the forward's item-id gathers wait for it
            lib.intel_set_table_wait_event(model._context(), C.c_void_p(self._table_ev.cuda_event))
            self._table_ev_keep = self._table_ev
            try:
                weights, ens, intents = model.run_forward(ib, keep, params, train=True, items=items, parr=parr)
            except BaseException:
                # the forward may not have enqueued its waits: take the one-shot event back and order EVERY later reader behind the sweep
                lib.intel_set_table_wait_event(model._context(), None)
                self._table_sync()
                raise
            self._table_ev = None           # handed over: the forward's gathers are ordered behind the sweep
        else:
            weights, ens, intents = model.run_forward(ib, keep, params, train=True, items=items, parr=parr)
        model._generation = getattr(model, '_generation', 0) + 1
        st = L.stream_ptr(dev)
        nb = lib.intel_loss_workspace_bytes(B, Lmax, K)
        ws = self._buf('loss_ws', (int(nb),), torch.uint8)
        loss_e = self._buf('loss_e', (1,), torch.float32)
        d_ens = self._buf('d_ens', (B, Lmax), torch.float32)
        d_w = self._buf('d_w', (B, Lmax, K), torch.float32)
        ranking, slen = keep.get('ranking_i32'), keep['session_len']
        if ranking is None:
            ranking = batch['ranking'] if batch['ranking'].dtype == torch.int32 else batch['ranking'].to(torch.int32)
            ranking = ranking.contiguous()
        sc = batch['scores']
        sc64 = sc.contiguous() if sc.dtype == torch.float64 else None
        sc32 = keep['scores']
        gs_e = self.ensemble_weight / world
        # the intent loss (three small launches) does not depend on the ensemble loss: it runs next to it on the context stream
        # the table sweep will use later (idle now), with its own scratch
        d_int, out3, iside = None, None, None
        if self.with_intent:
            out3 = self._buf('out3', (3,), torch.float64)
            d_int = self._buf('d_int', (B, I), torch.float32)
            ws_i = self._buf('loss_ws_intent', (int(nb),), torch.uint8)
            label = batch['intents']
            label = label if label.dtype == torch.float64 else label.double()
            label = label.contiguous()
            cur0 = torch.cuda.current_stream(dev)
            iside = self._table_stream() if (self.overlap_table_update and self.wide_backward) else None
            if iside is not None:
                iside.wait_stream(cur0)
            with torch.cuda.stream(iside if iside is not None else cur0):
                L.check(lib.intel_intent_loss(B, I, L.ptr(intents), L.ptr(label), self.kl_weight, self.kl_temp,
                                              self.intent_weight / world, L.ptr(out3), L.ptr(d_int), L.ptr(ws_i), nb, L.stream_ptr(dev)),
                        'intel_intent_loss')
        if self.kind == 'bpr':
            select = self._buf('select', (B, Lmax), torch.int32)
            if noise is None and self._noise_tensor:
                noise = torch.rand(B, Lmax, Lmax, dtype=torch.float32, device=dev)      # BPRloss.py:26, as a tensor
            if noise is None:
                # the tie-breaking noise of BPRloss.py:26 is drawn inside the kernel (no [B,L,L] tensor)
                seed = int(noise_seed) if noise_seed is not None else int(torch.randint(0, 2 ** 62, (1,), generator=self._noise_gen).item())
                session0 = parallel.rank() * B
                L.check(lib.intel_bpr_loss_seeded(B, Lmax, K, L.ptr(ens), L.ptr(ranking), L.ptr(slen), C.c_ulonglong(seed),
                                                  C.c_ulonglong(session0), L.ptr(sc64),
                                                  L.ptr(sc32), L.ptr(weights), self.cal_diversity, self.alpha, gs_e, L.ptr(loss_e),
                                                  L.ptr(select), L.ptr(d_ens), L.ptr(d_w), L.ptr(ws), nb, st), 'intel_bpr_loss_seeded')
            else:
                L.check(lib.intel_bpr_loss(B, Lmax, K, L.ptr(ens), L.ptr(ranking), L.ptr(slen), L.ptr(noise), L.ptr(sc64),
                                           L.ptr(sc32), L.ptr(weights), self.cal_diversity, self.alpha, gs_e, L.ptr(loss_e),
                                           L.ptr(select), L.ptr(d_ens), L.ptr(d_w), L.ptr(ws), nb, st), 'intel_bpr_loss')
        elif self.kind == 'mse':
            L.check(lib.intel_mse_loss(B, Lmax, K, L.ptr(ens), L.ptr(ranking), L.ptr(slen), L.ptr(sc64), L.ptr(sc32),
                                       L.ptr(weights), self.cal_diversity, self.alpha, gs_e, L.ptr(loss_e), L.ptr(d_ens),
                                       L.ptr(d_w), L.ptr(ws), nb, st), 'intel_mse_loss')
        else:
            L.check(lib.intel_list_loss(B, Lmax, K, L.ptr(ens), L.ptr(ranking), L.ptr(slen), L.ptr(sc64), L.ptr(sc32),
                                        L.ptr(weights), self.cal_diversity, self.alpha, gs_e, L.ptr(loss_e), L.ptr(d_ens),
                                        L.ptr(d_w), L.ptr(ws), nb, st), 'intel_list_loss')
        if iside is not None:
            torch.cuda.current_stream(dev).wait_stream(iside)
        # (loss, ensemble_loss, intent_loss) like the reference's criterion: one tiny launch into a fresh 3-vector (the
        # caller may keep every step's values: runner.fit averages them at the end of the epoch).  One-call backward: it rides on
        # the table stream behind the sweep (off both the head and the tail of the step); otherwise here, before the backward
        tot = torch.empty(3, dtype=torch.float64, device=dev)

        def loss_total(stream_ptr):
            L.check(lib.intel_loss_total(L.ptr(loss_e), L.ptr(out3), self.ensemble_weight, self.intent_weight, L.ptr(tot), stream_ptr), 'intel_loss_total')
        wide = self.overlap_table_update and self.wide_backward
        if not wide:
            loss_total(st)
        self.step_count += 1
        b1, b2 = self.betas
        if sort_ev is not None:
            torch.cuda.current_stream(dev).wait_event(sort_ev)

        def adam(gname, wd, stream_ptr, dense_reduced=False):
            n = self.flat[gname].numel()
            if gname == 'iid' and self._iid_flags is not None and n:
                if dense_reduced:       # the gradient was summed over ranks as a dense table: so are the row marks
                    parallel.allreduce_max_(self._iid_flags)
                if self._lazy is not None:
                    return self._lazy_step(stream_ptr)
                rows, d = self.model.iid_embeddings.weight.shape
                L.check(lib.intel_adam_step_rows(L.ptr(self.flat[gname]), L.ptr(self.gflat[gname]), L.ptr(self.m[gname]),
                                                 L.ptr(self.v[gname]), rows, d, L.ptr(self._iid_flags), self.lr, b1, b2, self.eps,
                                                 wd, self.step_count, 1.0, stream_ptr), 'intel_adam_step_rows')
            elif n:
                L.check(lib.intel_adam_step(L.ptr(self.flat[gname]), L.ptr(self.gflat[gname]), L.ptr(self.m[gname]),
                                            L.ptr(self.v[gname]), n, self.lr, b1, b2, self.eps, wd, self.step_count, 1.0,
                                            1, stream_ptr), 'intel_adam_step')
        if wide:
            # the whole backward in one call, its four branches (both towers, both encoders) on four streams.  The side stream is
            # made to wait (inside intel_backward) for the item-id table gradient only, so the table's exchange (data parallel:
            # touched rows all-gathered, or the dense all-reduce) and its dense Adam sweep -- HBM-bound, 28 B per parameter -- run
            # underneath the backward's tail of small launches (shared intent-embedding gradients, deferred reductions), the
            # dense buckets' all-reduce and the dense groups' Adam
            cur = torch.cuda.current_stream(dev)
            side = self._table_stream()
            lib.intel_set_table_stream(model._context(), C.c_void_p(side.cuda_stream))
            try:
                model.run_backward(ib, keep, params, d_w, d_ens, d_int, grad_tensors=self.grad_by_slot, items=items, parr=parr, garr=garr)
            finally:
                lib.intel_set_table_stream(model._context(), None)
            # exchange observability (bench.py --gpus N: `exchange` object): HIP events around the table exchange (+ its Adam sweep) on the
            # side stream and around the dense buckets' all-reduce on the main stream, and at the point where the main stream has nothing
            # left but to wait for the side stream -- exposed = how long the table branch outlasts everything else of the step
            tx = self._exchange_events if (dp and getattr(self, 'time_exchange', False)) else None
            with torch.cuda.stream(side):
                sparse = dp and self._sparse_exchange(keep, world)
                if tx is not None:
                    tx['form'] = 'sharded' if self._sharded() else ('touched_rows' if sparse else 'dense')
                    tx['t0'].record(side)
                if dp and self._sharded():
                    self._sharded_table_update(L.stream_ptr(dev))
                    if tx is not None:
                        tx['t1'].record(side)
                else:
                    if sparse:
                        self._exchange_touched_rows(keep, L.stream_ptr(dev))
                    elif dp:
                        parallel.allreduce_sum_([self.gflat['iid']])
                    if tx is not None:
                        tx['t1'].record(side)
                    adam('iid', self.l2, L.stream_ptr(dev), dense_reduced=dp and not sparse)
                if dp or not self.defer_table_wait:
                    loss_total(L.stream_ptr(dev))
                if tx is not None:
                    tx['t2'].record(side)
            if tx is not None:
                tx['b0'].record(cur)
            if dp:
                parallel.allreduce_sum_([self.gflat['decay'], self.gflat['nodecay']])
            if tx is not None:
                tx['b1'].record(cur)
            self._adam_dense_groups(st)
            if tx is not None:
                tx['m'].record(cur)
            if dp or not self.defer_table_wait:
                cur.wait_stream(side)
            else:
                # The table's sweep (HBM-bound, ~0.3 ms at the 1 M-row table) is NOT waited for here: the next step's forward starts under it -- its
                # session-history encoder and score tower never touch the table, the item tower's branch runs on this very side stream (stream order),
                # and the item-history encoder's gather waits for the event below inside intel_forward.  Every other reader goes through _table_sync.
                loss_total(st)
                self._table_ev = torch.cuda.Event()
                self._table_ev.record(side)
            if tx is not None:
                self._exchange_pending = True
        elif self.overlap_table_update:
            # phase 1 completes the item-id table gradient (the 256 MB bucket).  Its all-reduce (data parallel) and its
            # dense Adam sweep -- HBM-bound, 28 B per parameter -- then run on a side stream underneath phase 2 (score-tower
            # layers, session-history encoder: matrix work that touches neither the table nor its gradient)
            cur = torch.cuda.current_stream(dev)
            side = self._table_stream()         # a context stream (idle in phase 2), not a fifth stream of our own
            model.run_backward(ib, keep, params, d_w, d_ens, d_int, grad_tensors=self.grad_by_slot, items=items, parr=parr, garr=garr, phase=1)
            sparse = dp and self._sparse_exchange(keep, world)
            sharded = dp and self._sharded()
            work = parallel.allreduce_sum_async(self.gflat['iid']) if (dp and not sparse and not sharded) else None
            ev = torch.cuda.Event()
            ev.record(cur)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                if work is not None:
                    work.wait()
                if sharded:
                    self._sharded_table_update(L.stream_ptr(dev))
                else:
                    if sparse:
                        self._exchange_touched_rows(keep, L.stream_ptr(dev))
                    adam('iid', self.l2, L.stream_ptr(dev), dense_reduced=work is not None)
            model.run_backward(ib, keep, params, d_w, d_ens, d_int, grad_tensors=self.grad_by_slot, items=items, parr=parr, garr=garr, phase=2)
            if dp:
                parallel.allreduce_sum_([self.gflat['decay'], self.gflat['nodecay']])
            adam('decay', self.l2, st)
            adam('nodecay', 0.0, st)
            cur.wait_stream(side)
        else:
            model.run_backward(ib, keep, params, d_w, d_ens, d_int, grad_tensors=self.grad_by_slot, items=items, parr=parr, garr=garr)
            sharded = dp and self._sharded()
            if dp:
                parallel.allreduce_sum_(([] if sharded else [self.gflat['iid']]) + [self.gflat['decay'], self.gflat['nodecay']])
            if sharded:
                self._sharded_table_update(st)
            for gname, wd in (('iid', self.l2), ('decay', self.l2), ('nodecay', 0.0)):
                if not (sharded and gname == 'iid'):
                    adam(gname, wd, st, dense_reduced=dp)
        return tot[0], tot[1], tot[2]

    def _adam_dense_groups(self, stream_ptr):
        """torch's two dense parameter groups ('decay': weight_decay = l2, 'nodecay': biases, 0) in ONE launch (intel_adam_step_pair)."""
        names = ('decay', 'nodecay')
        arr = lambda d: (C.c_void_p * 2)(*[d[n].data_ptr() if d[n].numel() else None for n in names])
        pa = (arr(self.flat), arr(self.gflat), arr(self.m), arr(self.v), (C.c_longlong * 2)(*[self.flat[n].numel() for n in names]))
        b1, b2 = self.betas
        wd = (C.c_float * 2)(self.l2, 0.0)
        L.check(L.lib().intel_adam_step_pair(pa[0], pa[1], pa[2], pa[3], pa[4], wd, self.lr, b1, b2, self.eps, self.step_count, 1.0, 1, stream_ptr),
                'intel_adam_step_pair')

    # ---- data-parallel exchange timing (observability only) ---------------------------------------------
    @property
    def _exchange_events(self):
        ev = getattr(self, '_tx_events', None)
        if ev is None:
            ev = {k: torch.cuda.Event(enable_timing=True) for k in ('t0', 't1', 't2', 'b0', 'b1', 'm')}
            ev['form'] = None
            self._tx_events = ev
            self._tx_acc = {'steps': 0, 'table_exchange_ms': 0.0, 'table_branch_ms': 0.0, 'buckets_ms': 0.0, 'exposed_ms': 0.0}
        return ev

    def exchange_collect(self):
        """After a step run with ``time_exchange = True`` (synchronises): add its timings to the running totals."""
        if not getattr(self, '_exchange_pending', False):
            return
        torch.cuda.synchronize(self.device)
        ev, acc = self._tx_events, self._tx_acc
        acc['steps'] += 1
        acc['table_exchange_ms'] += ev['t0'].elapsed_time(ev['t1'])       # the collective(s) of the item-id table gradient (sharded: + Adam on the slice + all-gather)
        acc['table_branch_ms'] += ev['t0'].elapsed_time(ev['t2'])         # ... + the table's Adam sweep
        acc['buckets_ms'] += ev['b0'].elapsed_time(ev['b1'])              # all-reduce of the two dense buckets on the main stream
        acc['exposed_ms'] += max(0.0, ev['m'].elapsed_time(ev['t2']))     # the side stream still busy when the main stream has finished its own work
        self._exchange_pending = False

    def exchange_report(self):
        acc = getattr(self, '_tx_acc', None)
        if not acc or not acc['steps']:
            return None
        n = acc['steps']
        return {'form': self._tx_events['form'], 'steps': n, 'table_exchange_ms': round(acc['table_exchange_ms'] / n, 4),
                'table_branch_ms': round(acc['table_branch_ms'] / n, 4), 'dense_buckets_allreduce_ms': round(acc['buckets_ms'] / n, 4),
                'exposed_ms': round(acc['exposed_ms'] / n, 4), 'schedule': 'one-call backward, table exchange + sweep on a side stream'}

    # ---- evaluation -----------------------------------------------------------------------------------
    @torch.no_grad()
    def eval_step(self, batch, k=3):
        """forward + on-device NDCG@k (helpers/BaseRunner.py:328-343 + :117-126).  Returns
        (out_dict, ndcg[B] device tensor)."""
        model = self.model
        self.flush()                # lazy table Adam: settle the table once per evaluation phase (no-op when nothing is pending)
        ib, keep = model.prepare_batch(batch)
        items, params, parr, _ = self._param_cache()
        weights, ens, intents = model.run_forward(ib, keep, params, train=False, items=items, parr=parr)
        model._generation = getattr(model, '_generation', 0) + 1
        ndcg = self._buf('ndcg', (ib.B,), torch.float32)
        ranking = batch['ranking'] if batch['ranking'].dtype == torch.int32 else batch['ranking'].to(torch.int32)
        ranking = ranking.contiguous()
        L.check(L.lib().intel_ndcg(ib.B, ib.L, k, L.ptr(ens), L.ptr(ranking), L.ptr(keep['session_len']), L.ptr(ndcg),
                                   L.stream_ptr(self.device)), 'intel_ndcg')
        return {'weights': weights, 'ens_score': ens, 'intents': intents}, ndcg
