"""IntEL model -- host-side mirror of the reference's ``models/IntEL/IntEL.py`` (class ``IntEL``).

Same constructor contract (``IntEL(args, corpus)``), same CLI flags (``parse_model_args``), same
``state_dict`` keys and shapes, same ``forward(data) -> {"weights","ens_score","intents"}``; the
arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI (include/intel_hip.h).  The
sub-modules below are parameter containers only (their ``forward`` is never called): there is no
PyTorch / CPU fallback -- calling the model without the HIP library or off-GPU raises.
"""
import ctypes as C
import logging
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


def _list_product(xs):
    p = 1
    for x in xs:
        p *= int(x)
    return p


class _AttnHead(nn.Module):
    """Parameters of modules/layers.py:11-29 (MultiHeadAttention; no output projection)."""

    def __init__(self, d_model, n_heads, bias=True):
        super().__init__()
        self.d_model, self.h = d_model, n_heads
        self.q_linear = nn.Linear(d_model, d_model, bias=bias)
        self.k_linear = nn.Linear(d_model, d_model, bias=bias)
        self.v_linear = nn.Linear(d_model, d_model, bias=bias)


class _TransformerLayer(nn.Module):
    """Parameters of modules/layers.py:62-80 (TransformerLayer)."""

    def __init__(self, d_model, d_ff, n_heads):
        super().__init__()
        self.masked_attn_head = _AttnHead(d_model, n_heads)
        self.layer_norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.layer_norm2 = nn.LayerNorm(d_model)


class _BERT4RecEncoder(nn.Module):
    """Parameters of models/GeneralSeq.py:80-87."""

    def __init__(self, emb_size, max_his, num_layers=2, num_heads=2):
        super().__init__()
        self.p_embeddings = nn.Embedding(max_his + 1, emb_size)
        self.transformer_block = nn.ModuleList(
            [_TransformerLayer(emb_size, emb_size, num_heads) for _ in range(num_layers)])


class _GRU4RecEncoder(nn.Module):
    """Parameters of models/GeneralSeq.py:58-62."""

    def __init__(self, emb_size, hidden_size=128):
        super().__init__()
        self.rnn = nn.GRU(input_size=emb_size, hidden_size=hidden_size, batch_first=True)
        self.out = nn.Linear(hidden_size, emb_size, bias=False)


class _CrossAtt(nn.Module):
    """Parameters of modules/attention.py:121-147 (CrossAtt, bias-free q/k/v)."""

    def __init__(self, input_qsize, input_vsize, size):
        super().__init__()
        self.query_layer = nn.Linear(input_qsize, size, bias=False)
        self.key_layer = nn.Linear(input_vsize, size, bias=False)
        self.value_layer = nn.Linear(input_vsize, size, bias=False)


class IntEL(nn.Module):
    reader, runner = 'SeqReader', 'BaseRunner'
    extra_log_args = ['cross_attn_qsize', 'num_heads', 'num_layers', 'encoder', 'intent_emb_size']

    # ---- CLI contract (IntEL.py:17-34, GeneralSeq.py:15-17, BaseModel.py:20-27) ----------------
    @staticmethod
    def parse_model_args(parser):
        parser.add_argument('--encoder', type=str, default='BERT4Rec', help='A sequence encoder for intent prediction.')
        parser.add_argument('--context_emb_size', type=int, default=16, help='Embedding size for context.')
        parser.add_argument('--i_emb_size', type=int, default=16, help='Embedding size for item id.')
        parser.add_argument('--u_emb_size', type=int, default=32, help='Embedding size for user.')
        parser.add_argument('--s_emb_size', type=int, default=32, help='Embedding size for score.')
        parser.add_argument('--im_emb_size', type=int, default=16, help='Embedding size for item metadata.')
        parser.add_argument('--intent_emb_size', type=int, default=16, help='Embedding size for intent.')
        parser.add_argument('--cross_attn_qsize', type=int, default=32, help='Embedding size for cross-attention query.')
        parser.add_argument('--num_heads', type=int, default=1, help='Number of attention heads.')
        parser.add_argument('--dropout', type=float, default=0, help='Dropout probability for each deep layer')
        parser.add_argument('--num_layers', type=int, default=1, help='Number of self-attention layers.')
        parser.add_argument('--cross_attention', type=int, default=1,
                            help='Using cross-attention structure or direct attention.')
        parser.add_argument('--history_max', type=int, default=20)
        parser.add_argument('--model_path', type=str, default='', help='Model save path.')
        parser.add_argument('--buffer', type=int, default=1, help='Whether to buffer feed dicts for dev/test')
        parser.add_argument('--model_num', type=int, default=2, help='Number of base models.')
        parser.add_argument('--dtype', type=str, default='f32',
                            help='f32: the parity mode; bf16: every linear / weight gradient is ONE bf16 MFMA product (operands rounded '
                                 'to bf16, fp32 accumulate; fp32 master weights, Adam moments, residuals, LayerNorm, softmax)')
        parser.add_argument('--weight_norm', type=str, default='none',
                            help='none: raw fusion weights (IntEL.py:214, the reference); softmax: K-way softmax over them')
        return parser

    # variant switches (class attributes; aWELv_IntEL overrides them)
    POOL_MEAN = 0             # 1: mean-pooled h * g(intent) feature, one weight vector per session (aWELv_IntEL.py:188-201)
    FORCE_WEIGHT_NORM = None  # number of softmax applications fixed by the class (aWELv_IntEL: 2)

    def __init__(self, args, corpus):
        super().__init__()
        # BaseModel / GeneralModel / GeneralSeq attributes (BaseModel.py:38-45,151-155; GeneralSeq.py:19-21)
        self.intent_num = len(corpus.zero_int)
        self.device = getattr(args, 'device', torch.device('cpu'))
        self.model_path = getattr(args, 'model_path', '')
        self.buffer = getattr(args, 'buffer', 1)
        self.optimizer = None
        self.check_list = list()
        self.user_num = int(corpus.max_uid + 1)
        self.item_num = int(corpus.max_iid + 1)
        self.max_his = args.history_max
        self.itemfnum = _list_product(corpus.itemfnum)
        self.model_num = args.model_num
        self.dropout = float(getattr(args, 'dropout', 0))

        # parameters, created in the reference's order (IntEL.py:43-115) so that a given
        # torch.manual_seed yields the same default initialisation
        self.iid_embeddings = nn.Embedding(self.item_num, args.i_emb_size)
        self.im_emb_size = 0
        if self.itemfnum > 0:
            self.item_embeddings = nn.Embedding(self.itemfnum, args.im_emb_size)
            self.im_emb_size = args.im_emb_size
        self.uid_embeddings = nn.Embedding(self.user_num, args.u_emb_size)
        self.intent_embeddings = nn.Linear(self.intent_num, args.intent_emb_size)
        self.score_embeddings = nn.Linear(args.model_num, args.s_emb_size)
        self.head_num, self.layer_num = args.num_heads, args.num_layers
        self.item_emb_size = args.i_emb_size + self.im_emb_size
        self.i_attn_head = _AttnHead(self.item_emb_size, self.head_num, bias=False)
        self.i_W1 = nn.Linear(self.item_emb_size, self.item_emb_size)
        self.i_W2 = nn.Linear(self.item_emb_size, self.item_emb_size)
        self.i_layer_norm = nn.LayerNorm(self.item_emb_size)
        self.score_emb_size = args.s_emb_size
        self.s_attn_head = _AttnHead(self.score_emb_size, self.head_num, bias=False)
        self.s_W1 = nn.Linear(self.score_emb_size, self.score_emb_size)
        self.s_W2 = nn.Linear(self.score_emb_size, self.score_emb_size)
        self.s_layer_norm = nn.LayerNorm(self.score_emb_size)
        self.cross_attn_qsize = args.cross_attn_qsize
        self.cross_attention = 0 if self.POOL_MEAN else getattr(args, 'cross_attention', 1)
        wn = getattr(args, 'weight_norm', 'none')
        if wn not in ('none', 'softmax', 0, 1):
            raise ValueError('weight_norm must be none or softmax')
        self.weight_norm = self.FORCE_WEIGHT_NORM if self.FORCE_WEIGHT_NORM is not None else int(wn in ('softmax', 1))
        if self.cross_attention:
            self.intent_score_attention = _CrossAtt(self.intent_num, self.score_emb_size, self.score_emb_size)
            self.intent_item_attention = _CrossAtt(self.intent_num, self.item_emb_size, self.item_emb_size)
        else:
            self.intent_score_embeddings = nn.Sequential(
                nn.Linear(self.intent_num, self.cross_attn_qsize), nn.ReLU(),
                nn.Linear(self.cross_attn_qsize, self.score_emb_size, bias=False))
            self.intent_item_embeddings = nn.Sequential(
                nn.Linear(self.intent_num, self.cross_attn_qsize), nn.ReLU(),
                nn.Linear(self.cross_attn_qsize, self.item_emb_size, bias=False))
        self.weight_embeddings = nn.Linear(
            self.item_emb_size + args.s_emb_size + args.intent_emb_size + args.u_emb_size, args.model_num)
        self.context_embeddings = nn.Embedding(_list_product(corpus.contextfnum), args.context_emb_size)
        self.encoder_name = args.encoder
        self.intent_pred_size = args.intent_emb_size + args.context_emb_size
        self.his_item_dim = args.intent_emb_size + args.i_emb_size
        if self.encoder_name == 'GRU4Rec':
            self.encoder = _GRU4RecEncoder(self.intent_pred_size, hidden_size=128)
            self.item_encoder = _GRU4RecEncoder(self.his_item_dim, hidden_size=128)
        elif self.encoder_name == 'BERT4Rec':
            self.encoder = _BERT4RecEncoder(self.intent_pred_size, self.max_his, num_layers=2, num_heads=2)
            self.item_encoder = _BERT4RecEncoder(self.his_item_dim, self.max_his, num_layers=2, num_heads=2)
        else:
            raise ValueError('Invalid sequence encoder.')
        self.pred_layer = nn.Linear(
            self.intent_pred_size + self.his_item_dim + args.context_emb_size + args.u_emb_size, self.intent_num)

        self._desc = L.IntelDesc(
            model_num=args.model_num, intent_num=self.intent_num, item_num=self.item_num,
            class_num=max(self.itemfnum, 1), user_num=self.user_num,
            ctx_num=_list_product(corpus.contextfnum), d_id=args.i_emb_size, d_im=self.im_emb_size,
            d_u=args.u_emb_size, d_s=args.s_emb_size, d_c=args.context_emb_size, d_int=args.intent_emb_size,
            q_size=args.cross_attn_qsize, heads=args.num_heads, layers=args.num_layers,
            cross_attention=int(bool(self.cross_attention)),
            encoder=0 if self.encoder_name == 'BERT4Rec' else 1, history_max=self.max_his,
            enc_layers=2, enc_heads=2, gru_hidden=128, weight_norm=self.weight_norm, pool_mean=int(self.POOL_MEAN),
            dtype={'f32': 0, 'fp32': 0, 'bf16': 1}[str(getattr(args, 'dtype', 'f32'))])
        self._ctx = None
        self._ws = None
        self._slot_names = self._build_slot_map()

    # ---- reference auxiliary API (BaseModel.py:53-78) -------------------------------------------
    def customize_parameters(self, define_dict={}):
        weight_p, bias_p = [], []
        for name, p in filter(lambda x: x[1].requires_grad, self.named_parameters()):
            (bias_p if 'bias' in name else weight_p).append(p)
        return [{'params': weight_p}, {'params': bias_p, 'weight_decay': 0}]

    def save_model(self, model_path=None):
        model_path = model_path or self.model_path
        d = os.path.dirname(model_path)
        if d and not os.path.exists(d):
            os.makedirs(d)
        torch.save(self.state_dict(), model_path)

    def load_model(self, model_path=None):
        model_path = model_path or self.model_path
        self.load_state_dict(torch.load(model_path, map_location=self.device))
        logging.info('Load model from ' + model_path)

    def count_variables(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def to(self, *a, **k):
        m = super().to(*a, **k)
        try:
            self.device = next(self.parameters()).device
        except StopIteration:
            pass
        return m

    # ---- parameter slots of the C ABI -------------------------------------------------------------
    def _build_slot_map(self):
        P = L.P
        m = {
            P['IID_EMB']: 'iid_embeddings.weight', P['UID_EMB']: 'uid_embeddings.weight',
            P['CTX_EMB']: 'context_embeddings.weight',
            P['INTENT_W']: 'intent_embeddings.weight', P['INTENT_B']: 'intent_embeddings.bias',
            P['SCORE_W']: 'score_embeddings.weight', P['SCORE_B']: 'score_embeddings.bias',
            P['WE_W']: 'weight_embeddings.weight', P['WE_B']: 'weight_embeddings.bias',
            P['PRED_W']: 'pred_layer.weight', P['PRED_B']: 'pred_layer.bias',
        }
        if self.itemfnum > 0:
            m[P['ITEM_EMB']] = 'item_embeddings.weight'
        for t, pre in (('I', 'i'), ('S', 's')):
            m[P[t + '_WQ']] = pre + '_attn_head.q_linear.weight'
            m[P[t + '_WK']] = pre + '_attn_head.k_linear.weight'
            m[P[t + '_WV']] = pre + '_attn_head.v_linear.weight'
            m[P[t + '_W1']] = pre + '_W1.weight'
            m[P[t + '_B1']] = pre + '_W1.bias'
            m[P[t + '_W2']] = pre + '_W2.weight'
            m[P[t + '_B2']] = pre + '_W2.bias'
            m[P[t + '_LNG']] = pre + '_layer_norm.weight'
            m[P[t + '_LNB']] = pre + '_layer_norm.bias'
        if self.cross_attention:
            for t, pre in (('XI', 'intent_item_attention'), ('XS', 'intent_score_attention')):
                m[P[t + '_WQ']] = pre + '.query_layer.weight'
                m[P[t + '_WK']] = pre + '.key_layer.weight'
                m[P[t + '_WV']] = pre + '.value_layer.weight'
        else:
            for t, pre in (('MI', 'intent_item_embeddings'), ('MS', 'intent_score_embeddings')):
                m[P[t + '_W0']] = pre + '.0.weight'
                m[P[t + '_B0']] = pre + '.0.bias'
                m[P[t + '_W2']] = pre + '.2.weight'
        for e, pre in ((0, 'encoder'), (1, 'item_encoder')):
            base = L.P_ENC0 + e * L.ENC_STRIDE
            if self.encoder_name == 'BERT4Rec':
                m[base + L.ENC_POS] = pre + '.p_embeddings.weight'
                for l in range(2):
                    b = base + L.ENC_BLOCK0 + l * L.ENC_BLOCK_STRIDE
                    tb = '%s.transformer_block.%d.' % (pre, l)
                    names = {'WQ': 'masked_attn_head.q_linear.weight', 'BQ': 'masked_attn_head.q_linear.bias',
                             'WK': 'masked_attn_head.k_linear.weight', 'BK': 'masked_attn_head.k_linear.bias',
                             'WV': 'masked_attn_head.v_linear.weight', 'BV': 'masked_attn_head.v_linear.bias',
                             'LN1G': 'layer_norm1.weight', 'LN1B': 'layer_norm1.bias',
                             'W1': 'linear1.weight', 'B1': 'linear1.bias', 'W2': 'linear2.weight', 'B2': 'linear2.bias',
                             'LN2G': 'layer_norm2.weight', 'LN2B': 'layer_norm2.bias'}
                    for i, nm in enumerate(L.ENC_BLOCK_NAMES):
                        m[b + i] = tb + names[nm]
            else:
                m[base + L.ENC_GRU_WIH] = pre + '.rnn.weight_ih_l0'
                m[base + L.ENC_GRU_WHH] = pre + '.rnn.weight_hh_l0'
                m[base + L.ENC_GRU_BIH] = pre + '.rnn.bias_ih_l0'
                m[base + L.ENC_GRU_BHH] = pre + '.rnn.bias_hh_l0'
                m[base + L.ENC_GRU_OUT] = pre + '.out.weight'
        return m

    def slot_items(self):
        """[(slot, name, parameter)] for every parameter the kernels read."""
        # resolved by attribute path each call (parameters may be re-assigned, e.g. by .to() or the engine's flat
        # buckets) but without walking the whole module tree: named_parameters() costs ~0.2 ms per call
        out = []
        for s, n in self._slot_order():
            obj = self
            for part in n.split('.'):
                obj = obj._modules[part] if part in obj._modules else obj._parameters[part]
            out.append((s, n, obj))
        return out

    def _slot_order(self):
        so = getattr(self, '_slot_order_cache', None)
        if so is None:
            so = sorted(self._slot_names.items())
            self._slot_order_cache = so
        return so

    def _param_array(self, tensors_by_slot):
        arr = (C.c_void_p * L.P_COUNT)()
        for s, t in tensors_by_slot.items():
            arr[s] = t.data_ptr() if t is not None else None
        return arr

    def _context(self):
        if self._ctx is None:
            ctx = L.lib().intel_create(C.byref(self._desc))
            if not ctx:
                raise L.IntelHipError('intel_create failed: ' + L.lib().intel_last_error().decode())
            self._ctx = ctx
        return self._ctx

    def __del__(self):
        try:
            if getattr(self, '_ctx', None):
                L.lib().intel_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    # ---- batch conversion ---------------------------------------------------------------------------
    @staticmethod
    def _i32(t):
        return t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous()

    @staticmethod
    def _f32(t):
        return t if t.dtype == torch.float32 and t.is_contiguous() else t.to(torch.float32).contiguous()

    def prepare_batch(self, data):
        """Narrow a reference-layout batch dict (BaseModel.py:121-142) to the ABI layout.  Returns
        (IntelBatch struct, dict of tensors kept alive).  Already-converted batches pass through."""
        if '_intel' in data:
            return data['_intel']
        dev = data['i_id_s'].device
        L.require_gpu(data['i_id_s'])
        keep = {}
        keep['i_id_s'] = self._i32(data['i_id_s'])
        Bsz, Lmax = keep['i_id_s'].shape
        if 'i_class_c' in data and data['i_class_c'] is not None:
            keep['i_class_c'] = self._i32(data['i_class_c'])
        else:
            keep['i_class_c'] = torch.zeros(Bsz, Lmax, dtype=torch.int32, device=dev)
        keep['scores'] = self._f32(data['scores'])
        for k in ('session_len', 'u_id_c', 'context_mh', 'his_context_mh', 'history_len', 'his_item_id',
                  'history_item_len'):
            keep[k] = self._i32(data[k])
        keep['his_intents'] = self._f32(data['his_intents'])
        if 'his_item_idx' in data:
            keep['his_item_idx'] = self._i32(data['his_item_idx'])
        else:
            keep['his_item_int'] = self._f32(data['his_item_int'])
        H, Hi = keep['his_context_mh'].shape[1], keep['his_item_id'].shape[1]
        if os.environ.get('INTEL_CHECK_IDS') == '1':
            # nn.Embedding raises on an out-of-range id; the gather / scatter kernels do not look: opt-in check (synchronises)
            for key, hi in (('i_id_s', self.item_num), ('his_item_id', self.item_num), ('i_class_c', max(self.itemfnum, 1)),
                            ('u_id_c', self.user_num), ('context_mh', self._desc.ctx_num), ('his_context_mh', self._desc.ctx_num)):
                t = keep[key]
                if t.numel() and (int(t.max()) >= hi or int(t.min()) < 0):
                    raise L.IntelHipError('batch[%r] holds ids outside [0, %d)' % (key, hi))
        b = L.IntelBatch(B=Bsz, L=Lmax, H=H, Hi=Hi)
        # packed histories: when the producer of the batch knows the total number of valid history rows on the HOST (the device
        # feed, the synthetic generator and data.collate_batch do: 'his_rows' / 'hisitem_rows'), the sequence encoders run on those
        # rows only.  The offsets are two small prefix sums; nothing here synchronises with the device
        if 'his_rows' in data and 'hisitem_rows' in data:
            for key, lens, total in (('his_off', keep['history_len'], data['his_rows']), ('hisitem_off', keep['history_item_len'], data['hisitem_rows'])):
                c = torch.cumsum(lens, 0, dtype=torch.int32)
                keep[key] = (c - lens).contiguous()
            b.n_his_rows, b.n_hisitem_rows = int(data['his_rows']), int(data['hisitem_rows'])
        if self.encoder_name == 'GRU4Rec' and os.environ.get('INTEL_GRU_ORDER', '1') != '0' and Bsz >= int(os.environ.get('INTEL_GRU_ORDER_MIN_B', '1024')):      # fewer workgroups than CUs: nothing to free
            # sessions ordered by history length for the one-kernel recurrence (a workgroup's time loop runs to the longest of its 16
            # sessions): depends on the batch only, like the row offsets above
            desc = os.environ.get('INTEL_GRU_ORDER', '1') == '2'          # 2: longest first (training equal, evaluation 4.1 M against 4.3 M sessions/s)
            keep['his_order'] = torch.argsort(keep['history_len'], descending=desc, stable=True).to(torch.int32).contiguous()
            keep['hisitem_order'] = torch.argsort(keep['history_item_len'], descending=desc, stable=True).to(torch.int32).contiguous()
        for k, v in keep.items():
            setattr(b, k, v.data_ptr())
        prepared = (b, keep)
        return prepared

    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        off = (-self._ws.data_ptr()) % 256
        return self._ws[off:]

    # ---- forward ---------------------------------------------------------------------------------
    def forward(self, data):
        if os.environ.get('INTEL_MODEL_OP') == '1' and '_intel' not in data:      # the dispatcher-visible whole-model op (ops.py)
            from . import ops
            return ops.model_forward(self, data)
        batch, keep = self.prepare_batch(data)
        items = self.slot_items()
        params = [p for _, _, p in items]
        # two independent decisions, as in the reference: the activation stash follows autograd (is a backward possible?),
        # nn.Dropout follows module.training only (IntEL.py:63,187,196) -- run_forward reads self.training itself
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        if need_grad:
            w, e, i = _IntELFunction.apply(self, batch, keep, *params)
        else:
            w, e, i = self.run_forward(batch, keep, [p.detach() for p in params], train=False)
            self._generation = getattr(self, '_generation', 0) + 1
        return {'weights': w, 'ens_score': e, 'intents': i}

    def invalidate_packed(self):
        """Forget the packed weight images of the last evaluation forward.  For writers that change parameters behind torch's
        version counters (raw ``p.data`` writes: parallel.broadcast_ of the initial replicas, custom loaders)."""
        self._opt_steps = getattr(self, '_opt_steps', 0) + 1
        self._packed_key = None

    def _params_key(self, param_tensors):
        """Changes whenever a parameter may have changed: torch's version counters (in-place ops, load_state_dict), the
        storage addresses, and the engine's own step counter (its fused optimizer writes through raw pointers)."""
        return (getattr(self, '_opt_steps', 0), tuple(t._version for t in param_tensors), tuple(t.data_ptr() for t in param_tensors))

    def run_forward(self, batch, keep, param_tensors, train, items=None, parr=None):
        # items / parr: the caller's cached slot list and parameter-pointer array (IntELEngine: its flat buckets never move; the
        # ~100-parameter walks below are a quarter of the host time of a 0.7 ms step)
        if parr is None:
            items = items if items is not None else self.slot_items()
            for t in param_tensors:
                L.require_gpu(t)
        dev = keep['i_id_s'].device
        lib = L.lib()
        ctx = self._context()
        # nn.Dropout of the tower layers (IntEL.py:63,187,196): training only; a fresh seed per forward from torch's CPU
        # generator (reproducible under torch.manual_seed); tests may pin the keep flags through self._dropout_keep
        p_drop = float(self.dropout) if self.training else 0.0     # module.training, not grad mode
        if p_drop > 0:
            train = True        # the dropout path keeps its mask in the stash area (written even when no backward follows)
        keep_flags = getattr(self, '_dropout_keep', None) if p_drop > 0 else None
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (p_drop > 0 and keep_flags is None) else 0
        L.check(lib.intel_set_dropout(ctx, p_drop, C.c_ulonglong(seed), L.ptr(keep_flags)), 'intel_set_dropout')
        nb = lib.intel_workspace_bytes(ctx, batch.B, batch.L, batch.H, batch.Hi, int(train))
        ws = self._workspace(nb, dev)
        K, I = self.model_num, self.intent_num
        weights = torch.empty(batch.B, batch.L, K, dtype=torch.float32, device=dev)
        ens = torch.empty(batch.B, batch.L, dtype=torch.float32, device=dev)
        intents = torch.empty(batch.B, I, dtype=torch.float32, device=dev)
        out = L.IntelOut(weights=weights.data_ptr(), ens_score=ens.data_ptr(), intents=intents.data_ptr())
        if parr is None:
            parr = self._param_array({s: t.contiguous() for (s, _, _), t in zip(items, param_tensors)})
        # evaluation over a frozen model: the packed weight images of the previous forward are still valid
        key = None if train else self._params_key(param_tensors)
        lib.intel_set_params_unchanged(ctx, int(key is not None and key == getattr(self, '_packed_key', None)))
        self._packed_key = key
        L.check(lib.intel_forward(ctx, parr, C.byref(batch), L.ptr(ws), ws.numel(), C.byref(out), int(train),
                                  L.stream_ptr(dev)), 'intel_forward')
        return weights, ens, intents

    def run_backward(self, batch, keep, param_tensors, d_weights, d_ens, d_intents, grad_tensors=None, phase=0, items=None, parr=None, garr=None):
        """d(out) -> d(param).  grad_tensors: optional {slot: tensor} of persistent buffers (embedding
        tables must arrive zeroed); otherwise fresh ones are allocated.  items / parr / garr: the caller's cached slot list and
        pointer arrays (see run_forward)."""
        if parr is None or (garr is None and grad_tensors is None):
            items = items if items is not None else self.slot_items()
        dev = keep['i_id_s'].device
        lib = L.lib()
        ctx = self._context()
        nb = lib.intel_workspace_bytes(ctx, batch.B, batch.L, batch.H, batch.Hi, 1)
        ws = self._workspace(nb, dev)
        table_slots = (L.P['IID_EMB'], L.P['ITEM_EMB'], L.P['UID_EMB'], L.P['CTX_EMB'])
        if grad_tensors is None:
            grad_tensors = {}
            for (s, _, p), t in zip(items, param_tensors):
                if not p.requires_grad:
                    continue
                grad_tensors[s] = torch.zeros_like(t) if s in table_slots else torch.empty_like(t)
        if parr is None:
            parr = self._param_array({s: t.contiguous() for (s, _, _), t in zip(items, param_tensors)})
        if garr is None:
            garr = self._param_array(grad_tensors)
        L.check(lib.intel_backward_phase(ctx, parr, C.byref(batch), L.ptr(ws), ws.numel(), L.ptr(d_weights), L.ptr(d_ens),
                                         L.ptr(d_intents), garr, int(phase), L.stream_ptr(dev)), 'intel_backward')
        return grad_tensors


class aWELv_IntEL(IntEL):
    """The reference's ``models/supervise/aWELv_IntEL.py`` (class ``aWELv_IntEL``): IntEL's towers and intent predictor with
    aWELv-style fusion weights -- ``h * MLP(intent)`` mean-pooled over the whole list (pads included, nothing is masked,
    :188-198), ONE weight vector per session, softmax applied twice (:199-200) and repeated over the list.  Same parameter
    names as IntEL with ``--cross_attention 0`` (``intent_{item,score}_embeddings``), same flags minus ``--cross_attention``
    (:16-34), same kernels for everything up to the pooling."""
    POOL_MEAN = 1
    FORCE_WEIGHT_NORM = 2

    @staticmethod
    def parse_model_args(parser):
        IntEL.parse_model_args(parser)
        for a in list(parser._actions):          # the reference class has neither flag
            if a.dest in ('cross_attention', 'weight_norm'):
                parser._remove_action(a)
                for o in a.option_strings:
                    parser._option_string_actions.pop(o, None)
        return parser


class _IntELFunction(torch.autograd.Function):
    """Makes ``loss.backward()`` (helpers/BaseRunner.py:288) drive the hand-written backward."""

    @staticmethod
    def forward(ctx, model, batch, keep, *params):
        ctx.model, ctx.batch, ctx.keep = model, batch, keep
        ctx.save_for_backward(*params)
        out = model.run_forward(batch, keep, [p.detach() for p in params], train=True)
        model._generation = getattr(model, '_generation', 0) + 1    # the stash lives in ONE workspace
        ctx.generation = model._generation
        return out

    @staticmethod
    def backward(ctx, d_weights, d_ens, d_intents):
        params = ctx.saved_tensors
        model = ctx.model
        if ctx.generation != model._generation:
            raise L.IntelHipError('backward of a stale forward: the activation stash was overwritten by a later '
                                  'training forward of the same model (one forward/backward pair at a time)')

        def c(t):
            return None if t is None else t.contiguous().float()
        grads = model.run_backward(ctx.batch, ctx.keep, [p.detach() for p in params], c(d_weights), c(d_ens), c(d_intents))
        items = model.slot_items()
        out = [grads.get(s) if p.requires_grad else None for (s, _, p) in items]
        return (None, None, None) + tuple(out)
