"""CPU oracle for the IntEL hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional PyTorch (CPU, fp32/fp64) restatement of the reference's per-session forward,
its BPR / Plackett-Luce / intent losses and its NDCG evaluation.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file; the
product (``intel_sigir2023_amd``) never does and fails loudly when its HIP library is missing.

Parity is PINNED: ``tests/test_oracle_golden.py`` checks every function here against the
fixtures in ``tests/golden/*.npz`` that ``tests/golden/make_golden.py`` produced by running
the reference itself (/root/reference, imported unmodified) in the build container.

Each function cites the reference lines it restates (paths relative to
/root/reference/IntEL/src).  Parameters are taken from a plain ``state_dict``-style mapping
``sd`` whose keys are the reference's (``models/IntEL/IntEL.py:36-115``).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------
class Config(object):
    """The subset of CLI flags the hot path depends on (models/IntEL/IntEL.py:17-34)."""

    def __init__(self, **kw):
        self.model_num = 3
        self.history_max = 20
        self.encoder = 'BERT4Rec'
        self.context_emb_size = 16
        self.i_emb_size = 16
        self.u_emb_size = 32
        self.s_emb_size = 32
        self.im_emb_size = 16
        self.intent_emb_size = 16
        self.cross_attn_qsize = 32
        self.num_heads = 1
        self.num_layers = 1
        self.cross_attention = 1
        self.dropout = 0.0
        self.weight_norm = 'none'      # 'softmax': one K-way softmax over the fusion weights (no reference counterpart)
        self.model_name = 'IntEL'      # 'aWELv_IntEL': models/supervise/aWELv_IntEL.py (mean-pooled gates, double softmax)
        # loss flags (loss/Baseloss.py:9-12, loss/BaseIntloss.py:13-20)
        self.intent_weight = 0.1
        self.ensemble_weight = 1.0
        self.kl_temp = 2.0
        self.kl_weight = 0.5
        self.cal_diversity = 0
        self.diversity_alpha = 0.01
        for k, v in kw.items():
            setattr(self, k, v)


# bf16-mode emulation (forward_bf16 below): the HIP build's `--dtype bf16` rounds BOTH operands of a product to bf16 exactly where the
# product runs on the bf16 matrix pipe -- every linear with 64 or 128 input features and an output width that is a multiple of 4
# (csrc/gemm.hip: launch_gemm_rows), the attention products of the whole-sequence kernels -- and nowhere else (odd widths, the pooling
# scores, the pruned last encoder block's one-row attention, LayerNorm, softmax, losses stay fp32).  fp32 accumulation in both.
_EMU = {'on': False}


def _bf(t):
    return t.bfloat16().float()


def _on_bf16_pipe(w):
    return _EMU['on'] and w.shape[1] in (64, 128) and w.shape[0] % 4 == 0


class _BfLinear(torch.autograd.Function):
    """y = bf16(x) bf16(w)^T + b with the BACKWARD the HIP build's bf16 mode runs (csrc/gemm.hip, NP = 1): the incoming gradient is
    rounded to bf16 as an operand of both products -- dx = bf16(dy) bf16(w), dw = bf16(dy)^T bf16(x) --, fp32 accumulation, the bias
    gradient summed in fp32.  (Plain autograd through the rounded forward would leave dy unrounded.)"""

    @staticmethod
    def forward(ctx, x, w, b):
        xb, wb = _bf(x), _bf(w)
        ctx.save_for_backward(xb, wb)
        ctx.has_b = b is not None
        return F.linear(xb, wb, b)

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        dyb = _bf(dy)
        dx = torch.matmul(dyb, wb)
        dw = torch.matmul(dyb.reshape(-1, dyb.shape[-1]).t(), xb.reshape(-1, xb.shape[-1]))
        db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_b else None
        return dx, dw, db


def _bf_linear(x, w, b=None):
    return _BfLinear.apply(x, w, b)


class _BfMatmul(torch.autograd.Function):
    """bf16(a) @ bf16(b) with bf16-rounded gradient operands (the B-row products of the pooling: csrc/gemm.hip)."""

    @staticmethod
    def forward(ctx, a, b):
        ab, bb = _bf(a), _bf(b)
        ctx.save_for_backward(ab, bb)
        return torch.matmul(ab, bb)

    @staticmethod
    def backward(ctx, dy):
        ab, bb = ctx.saved_tensors
        dyb = _bf(dy)
        return torch.matmul(dyb, bb.t()), torch.matmul(ab.t(), dyb)


class _BfAttention(torch.autograd.Function):
    """softmax(q k^T / sqrt(dk)) v with every matrix-pipe operand rounded to bf16, forward AND backward, as csrc/attn_seq.hip /
    tower.hip run it in bf16 mode: forward S = bf(q) bf(k)^T, O = bf(e) bf(v) / sum(e) (e = the unnormalised probabilities, row sums
    fp32); backward P = e / sum(e) in fp32, dP = bf(dO) bf(v)^T, delta = rowsum(P dP) (fp32), dS = P (dP - delta) / sqrt(dk),
    dV = bf(P)^T bf(dO), dQ = bf(dS) bf(k), dK = bf(dS)^T bf(q).
    delta_from_output: the flash-style general kernels (csrc/attn.hip: lists / histories longer than 64) take delta = rowsum(dO * O) from the
    stored fp32 output instead (their key-stationary dK/dV sweep cannot sum over all keys); everything else is the same."""

    @staticmethod
    def forward(ctx, q, k, v, key_mask, scale, delta_from_output=False):
        qb, kb, vb = _bf(q), _bf(k), _bf(v)
        s = torch.matmul(qb, kb.transpose(-1, -2)) * scale
        if key_mask is not None:
            s = s.masked_fill(~key_mask[:, None, None, :], float('-inf'))
        m = s.max(dim=-1, keepdim=True)[0]
        e = torch.exp(s - torch.where(torch.isinf(m), torch.zeros_like(m), m))
        e = torch.where(torch.isnan(e), torch.zeros_like(e), e)
        den = e.sum(-1, keepdim=True)
        den = torch.where(den > 0, den, torch.ones_like(den))
        o = torch.matmul(_bf(e), vb) / den
        ctx.save_for_backward(qb, kb, vb, e / den, o)
        ctx.scale = scale
        ctx.delta_from_output = delta_from_output
        return o

    @staticmethod
    def backward(ctx, do):
        qb, kb, vb, p, o = ctx.saved_tensors
        dob = _bf(do)
        dp = torch.matmul(dob, vb.transpose(-1, -2))
        delta = (do * o).sum(-1, keepdim=True) if ctx.delta_from_output else (p * dp).sum(-1, keepdim=True)
        ds = _bf(p * (dp - delta) * ctx.scale)
        dv = torch.matmul(_bf(p).transpose(-1, -2), dob)
        dq = torch.matmul(ds, kb)
        dk = torch.matmul(ds.transpose(-1, -2), qb)
        return dq, dk, dv, None, None, None


def _lin(x, sd, name, bias=True):
    b = sd.get(name + '.bias') if bias else None
    w = sd[name + '.weight']
    if _on_bf16_pipe(w):
        return _bf_linear(x, w, b)
    return F.linear(x, w, b)


# ----------------------------------------------------------------------------------------
# attention blocks
# ----------------------------------------------------------------------------------------
def mha(x, sd, prefix, heads, key_mask=None, bf16_products=True):
    """modules/layers.py:31-60.  No output projection; softmax over keys after subtracting the
    tensor-global max (a no-op unless a row sits ~88 below it); all-masked rows -> 0.
    bf16_products: under forward_bf16 only -- whether THIS attention runs on the bf16 pipe (q, k, v and the unnormalised
    probabilities rounded to bf16, row sums and the normalisation in fp32, as csrc/attn_seq.hip / tower.hip do)."""
    B, T, D = x.shape
    dk = D // heads

    def split(t):
        return t.view(B, T, heads, dk).transpose(1, 2)
    q = split(_lin(x, sd, prefix + '.q_linear'))
    k = split(_lin(x, sd, prefix + '.k_linear'))
    v = split(_lin(x, sd, prefix + '.v_linear'))
    if _EMU['on'] and bf16_products and dk in (64, 128):
        # head widths 64 / 128 only: the whole-sequence kernels (lists of <= 64 rows, csrc/attn_seq.hip: attn_seq_supported) and the general
        # kernels' bf16 form (longer lists, csrc/attn.hip: attn_bf16_products) exist for those; any other head width -- 2 heads on a 64-wide
        # tower -- runs the exact-fp32 general kernels in bf16 mode too (q, k, v arrive rounded only in so far as their linears were)
        o = _BfAttention.apply(q, k, v, key_mask, 1.0 / dk ** 0.5, T > 64)
        return o.transpose(1, 2).reshape(B, T, D)
    s = torch.matmul(q, k.transpose(-1, -2)) / dk ** 0.5
    if key_mask is not None:
        s = s.masked_fill(~key_mask[:, None, None, :], float('-inf'))
    p = torch.softmax(s - s.max(), dim=-1)
    p = torch.where(torch.isnan(p), torch.zeros_like(p), p)
    o = torch.matmul(p, v)
    return o.transpose(1, 2).reshape(B, T, D)


def _nudged(pre, taps, key, layer):
    """Test hook (tools/fuzz_parity.py): taps['__nudge__'][key][layer] is a CONSTANT tensor added to the pre-ReLU activations -- it picks a side
    for entries that sit on the kink of the relu (|pre| at rounding-noise level), where both one-sided derivatives are legitimate."""
    n = taps.get('__nudge__')
    if n is not None and key in n and n[key].get(layer) is not None:
        return pre + n[key][layer]
    return pre


def tied_tower(h, sd, attn, w1, w2, ln, heads, layers, keep=None, p=0.0, taps=None):
    """models/IntEL/IntEL.py:182-188 / 191-197: the SAME weights are applied ``layers`` times,
    attention is unmasked (padded rows act as keys and queries).  keep: per-layer 0/1 tensors of the
    training-mode nn.Dropout(p) applied before the residual add (:187, :196); None = evaluation.
    taps: optional dict; receives the pre-ReLU activations of every layer under the key ``w1`` (tests only)."""
    for l in range(layers):
        res = h
        # (bf16 emulation, evaluation: the one-kernel tower layer takes lists of up to 64 rows, widths 64 / 128, head dims 32 / 64 / 128)
        D_, dk_ = h.shape[-1], h.shape[-1] // heads
        # ... and the general attention kernels run longer lists on the bf16 pipe at head dims 64 / 128 (csrc/attn.hip: attn_bf16_products)
        h = mha(h, sd, attn, heads, bf16_products=(h.shape[1] <= 64 and D_ in (64, 128) and dk_ in (32, 64, 128)) or (h.shape[1] > 64 and dk_ in (64, 128)))
        h = _lin(h, sd, w1)
        if taps is not None:
            h = _nudged(h, taps, w1, l)
            taps.setdefault(w1, []).append(h.detach())
        h = _lin(torch.relu(h), sd, w2)
        if keep is not None:
            h = h * keep[l] / (1.0 - p)
        h = F.layer_norm(h + res, (h.shape[-1],), sd[ln + '.weight'], sd[ln + '.bias'], 1e-5)
    return h


def bert4rec(seq, lengths, sd, prefix, heads=2, layers=2, taps=None):
    """models/GeneralSeq.py:89-106 with modules/layers.py:82-88 blocks (bias=True, d_ff=d_model)."""
    B, T, D = seq.shape
    ar = torch.arange(T)
    valid = ar[None, :] < lengths[:, None]
    pos = ar[None, :] * valid.long()                      # pads -> position 0
    x = seq + sd[prefix + '.p_embeddings.weight'][pos]
    for l in range(layers):
        p = '%s.transformer_block.%d' % (prefix, l)
        # (bf16 emulation: the LAST block is run pruned by the HIP build -- one query row per session in an fp32 kernel)
        # and the whole-sequence attention kernels take histories of up to 64 rows with head dims 64 / 128)
        ctx = mha(x, sd, p + '.masked_attn_head', heads, key_mask=valid, bf16_products=l + 1 < layers and D // heads in (64, 128))      # any length: whole-sequence kernels up to 64 rows, the general ones beyond
        ctx = F.layer_norm(ctx + x, (D,), sd[p + '.layer_norm1.weight'], sd[p + '.layer_norm1.bias'], 1e-5)
        pre = _lin(ctx, sd, p + '.linear1')
        if taps is not None:
            pre = _nudged(pre, taps, p + '.linear1', 0)
            taps.setdefault(p + '.linear1', []).append(pre.detach())
        y = _lin(torch.relu(pre), sd, p + '.linear2')
        x = F.layer_norm(y + ctx, (D,), sd[p + '.layer_norm2.weight'], sd[p + '.layer_norm2.bias'], 1e-5)
    x = x * valid[:, :, None].float()
    return x[torch.arange(B), lengths - 1]


def gru4rec(seq, lengths, sd, prefix):
    """models/GeneralSeq.py:64-78: one-layer GRU (gate order r,z,n as torch.nn.GRU), hidden state
    after each session's own last step, then a bias-free projection."""
    B, T, D = seq.shape
    w_ih, w_hh = sd[prefix + '.rnn.weight_ih_l0'], sd[prefix + '.rnn.weight_hh_l0']
    b_ih, b_hh = sd[prefix + '.rnn.bias_ih_l0'], sd[prefix + '.rnn.bias_hh_l0']
    Hd = w_hh.shape[1]
    h = seq.new_zeros(B, Hd)
    for t in range(T):
        gi = F.linear(seq[:, t], w_ih, b_ih)
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, :Hd] + gh[:, :Hd])
        z = torch.sigmoid(gi[:, Hd:2 * Hd] + gh[:, Hd:2 * Hd])
        n = torch.tanh(gi[:, 2 * Hd:] + r * gh[:, 2 * Hd:])
        hn = (1 - z) * n + z * h
        live = (t < lengths)[:, None]
        h = torch.where(live, hn, h)
    return F.linear(h, sd[prefix + '.out.weight'])


def single_query_pool(intent, h, valid, sd, prefix, scale):
    """modules/attention.py:149-161 + 48-63 as called from IntEL.py:201-204.  The query has one
    row, the mask is [B,L,L]; every VALID row therefore receives the same pooled vector and every
    padded row receives 0.  The row max is taken over ALL L positions before
    masking (attention.py:57)."""
    q = F.linear(intent, sd[prefix + '.query_layer.weight'])            # [B,a]
    if _EMU['on']:
        # the HIP build's algebra (csrc/session.hip: xatt_pool): att_l = scale * <Wk^T q, h_l>, pooled = Wv (sum_l w_l h_l); the two
        # d x d products run on the bf16 pipe, the scores and the weighted sum in fp32
        wk, wv = sd[prefix + '.key_layer.weight'], sd[prefix + '.value_layer.weight']
        qk = _BfMatmul.apply(q, wk) if _on_bf16_pipe(wk.t()) else torch.matmul(q, wk)                          # [B,d]
        att = torch.einsum('bd,bld->bl', qk, h) * scale
        att = att - att.max(dim=-1, keepdim=True)[0]
        att = att.masked_fill(~valid, float('-inf'))
        w = torch.softmax(att, dim=-1)
        w = torch.where(torch.isnan(w), torch.zeros_like(w), w)
        xbar = torch.einsum('bl,bld->bd', w, h)
        pooled = _bf_linear(xbar, wv) if _on_bf16_pipe(wv) else F.linear(xbar, wv)
        return pooled[:, None, :] * valid[:, :, None].float()
    k = F.linear(h, sd[prefix + '.key_layer.weight'])                    # [B,L,a]
    v = F.linear(h, sd[prefix + '.value_layer.weight'])                  # [B,L,v]
    att = torch.einsum('ba,bla->bl', q, k) * scale
    att = att - att.max(dim=-1, keepdim=True)[0]
    att = att.masked_fill(~valid, float('-inf'))
    w = torch.softmax(att, dim=-1)
    w = torch.where(torch.isnan(w), torch.zeros_like(w), w)
    pooled = torch.einsum('bl,blv->bv', w, v)                            # [B,v]
    return pooled[:, None, :] * valid[:, :, None].float()


# ----------------------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------------------
def predict_intent(sd, data, cfg, taps=None):
    """models/IntEL/IntEL.py:126-155."""
    his = torch.cat([sd['context_embeddings.weight'][data['his_context_mh']],
                     _lin(data['his_intents'].float(), sd, 'intent_embeddings')], dim=-1)
    his_item = torch.cat([sd['iid_embeddings.weight'][data['his_item_id']],
                          _lin(data['his_item_int'].float(), sd, 'intent_embeddings')], dim=-1)
    if cfg.encoder == 'BERT4Rec':
        hv = bert4rec(his, data['history_len'], sd, 'encoder', taps=taps)
        hiv = bert4rec(his_item, data['history_item_len'], sd, 'item_encoder', taps=taps)
    elif cfg.encoder == 'GRU4Rec':
        hv = gru4rec(his, data['history_len'], sd, 'encoder')
        hiv = gru4rec(his_item, data['history_item_len'], sd, 'item_encoder')
    else:
        raise ValueError('Invalid sequence encoder.')
    cur = torch.cat([sd['context_embeddings.weight'][data['context_mh']],
                     sd['uid_embeddings.weight'][data['u_id_c']]], dim=-1)
    logits = _lin(torch.cat([cur, hiv, hv], dim=-1), sd, 'pred_layer')
    return torch.softmax(logits, dim=-1)


def predict_ensemble(sd, data, intent, cfg, dropout_keep=None, taps=None):
    """models/IntEL/IntEL.py:158-217."""
    scores = data['scores'].float()
    B, L, K = scores.shape
    valid = torch.arange(L)[None, :] < data['session_len'][:, None]
    h_i = torch.cat([sd['iid_embeddings.weight'][data['i_id_s']],
                     sd['item_embeddings.weight'][data['i_class_c']]], dim=-1)
    h_u = torch.relu(sd['uid_embeddings.weight'][data['u_id_c']])[:, None, :].expand(B, L, -1)
    ki, ks = (dropout_keep if dropout_keep is not None else (None, None))
    pd = float(getattr(cfg, 'dropout', 0.0))
    h_i = tied_tower(h_i, sd, 'i_attn_head', 'i_W1', 'i_W2', 'i_layer_norm', cfg.num_heads, cfg.num_layers, ki, pd, taps)
    h_s = _lin(scores, sd, 'score_embeddings')
    h_s = tied_tower(h_s, sd, 's_attn_head', 's_W1', 's_W2', 's_layer_norm', cfg.num_heads, cfg.num_layers, ks, pd, taps)
    if getattr(cfg, 'model_name', 'IntEL') == 'aWELv_IntEL':
        # models/supervise/aWELv_IntEL.py:188-203: gates from the intent MLPs, mean over ALL rows, softmax twice, no mask
        def mlp(p):
            t = torch.relu(_lin(intent, sd, p + '.0'))
            return F.linear(t, sd[p + '.2.weight'])
        item_x = (h_i * mlp('intent_item_embeddings')[:, None, :]).mean(dim=1)
        score_x = (h_s * mlp('intent_score_embeddings')[:, None, :]).mean(dim=1)
        h_int = torch.relu(_lin(intent, sd, 'intent_embeddings'))
        feat = torch.cat([item_x, score_x, h_u[:, 0, :], h_int], dim=-1)
        w_item = torch.softmax(_lin(feat, sd, 'weight_embeddings'), dim=-1)
        weights = torch.softmax(w_item[:, None, :].repeat(1, L, 1), dim=-1)
        return weights, (weights * scores).sum(-1)
    if cfg.cross_attention:
        scale = 1.0 / math.sqrt(cfg.cross_attn_qsize)
        item_x = single_query_pool(intent, h_i, valid, sd, 'intent_item_attention', scale)
        score_x = single_query_pool(intent, h_s, valid, sd, 'intent_score_attention', scale)
    else:
        def mlp(p):
            t = torch.relu(_lin(intent, sd, p + '.0'))
            return F.linear(t, sd[p + '.2.weight'])
        item_x = h_i * mlp('intent_item_embeddings')[:, None, :]
        score_x = h_s * mlp('intent_score_embeddings')[:, None, :]
    h_int = torch.relu(_lin(intent, sd, 'intent_embeddings'))[:, None, :].expand(B, L, -1)
    feat = torch.cat([item_x, score_x, h_u, h_int], dim=-1)
    weights = _lin(feat, sd, 'weight_embeddings')            # NO softmax (IntEL.py:214)
    if _EMU['on'] and cfg.cross_attention:
        # the HIP build computes the PAD rows' weights as their own product over [h_u | h_intent] (the pooled columns are zero
        # there): d_u + d_int input features -- on the bf16 pipe when that is 64 or 128 and K is a multiple of 4
        w_all, npad = sd['weight_embeddings.weight'], h_u.shape[-1] + h_int.shape[-1]
        w_pad = w_all[:, w_all.shape[1] - npad:]
        if _on_bf16_pipe(w_pad):
            pad = _bf_linear(torch.cat([h_u, h_int], dim=-1), w_pad, sd['weight_embeddings.bias'])
            weights = torch.where(valid[:, :, None], weights, pad)
    if getattr(cfg, 'weight_norm', 'none') == 'softmax':      # the build's optional K-way normalisation
        weights = torch.softmax(weights, dim=-1)
    ens = (weights * scores).sum(-1)
    return weights, ens


def forward(sd, data, cfg, dropout_keep=None, taps=None):
    """models/IntEL/IntEL.py:117-124.  dropout_keep = (item-tower keep masks, score-tower keep masks), one 0/1
    tensor per layer, reproduces a training-mode forward with that nn.Dropout draw.  taps: optional dict that
    receives the pre-ReLU activations of every feed-forward block, keyed by the first linear's name (tests only)."""
    intent = predict_intent(sd, data, cfg, taps)
    weights, ens = predict_ensemble(sd, data, intent, cfg, dropout_keep, taps)
    return {'weights': weights, 'ens_score': ens, 'intents': intent}


def forward_bf16(sd, data, cfg):
    """`forward` with the rounding points of the HIP build's `--dtype bf16` mode (see _EMU above): an independent CPU
    restatement of WHAT that mode computes, for tests/test_bf16_gpu.py.  BERT4Rec encoders.  The products are autograd
    Functions (_BfLinear, _BfMatmul, _BfAttention) whose backward rounds the gradient operands exactly where the HIP backward
    does, so `loss(forward_bf16(...)).backward()` is the gradient oracle of the mode."""
    _EMU['on'] = True
    try:
        return forward(sd, data, cfg)
    finally:
        _EMU['on'] = False


# ----------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------
def _pair_setup(ens, ranking, session_len):
    L = ens.shape[1]
    valid = torch.arange(L)[None, :] < session_len[:, None]
    vv = valid[:, :, None] & valid[:, None, :]
    r = ranking.clamp(min=0)
    z = ens[:, :, None] - ens[:, None, :]
    return vv, r, z


def bpr_select(ranking, session_len, noise):
    """loss/BPRloss.py:20-30: index of the sampled negative for every row; candidates are the
    valid items of the closest lower rank tier, ties broken by ``noise/10``; a row without any
    lower-ranked valid item picks argmax(noise) over ALL L columns."""
    L = ranking.shape[1]
    valid = torch.arange(L)[None, :] < session_len[:, None]
    vv = (valid[:, :, None] & valid[:, None, :]).long()
    r = ranking.clamp(min=0)
    D = (r[:, :, None] - r[:, None, :]) * vv
    sim = (D.max() + 1 - D) * (D > 0)
    best = sim.max(dim=-1, keepdim=True)[0]
    cand = ((sim == best) & (D > 0)).int()
    return (cand + noise / 10).argmax(dim=-1)


def bpr_loss(ens, ranking, session_len, noise, scores=None, weights=None, cal_diversity=0, alpha=0.01):
    """loss/BPRloss.py:37-56 (+ diversity :12-18, float64 base-score differences)."""
    vv, r, z = _pair_setup(ens, ranking, session_len)
    sel = bpr_select(ranking, session_len, noise)
    pos = (r > 0)
    npos = pos.sum(-1)
    zs = torch.gather(z, 2, sel[:, :, None]).squeeze(2)
    loss = ((-torch.log(torch.sigmoid(zs)) * pos).sum(-1) / npos).mean()
    if cal_diversity:
        sg = torch.sigmoid(zs)
        sig = sg * (1 - sg)
        bsel = torch.gather(scores, 1, sel[:, :, None].expand(-1, -1, scores.shape[2]))
        bd = scores - bsel                                              # float64 [B,L,K]
        zd = sig[:, :, None] * (bd - zs[:, :, None]) ** 2
        A = (zd * weights).sum(-1) * pos
        div = -(A.sum(-1) / npos).mean()
        loss = (loss.double() + div * alpha).float()     # in-place += keeps float32 (BPRloss.py:54)
    return loss


def list_loss(ens, ranking, session_len, scores=None, weights=None, cal_diversity=0, alpha=0.01):
    """loss/Listloss.py:25-43 (+ diversity :17-23).  No max-subtraction in the exponent."""
    vv, r, z = _pair_setup(ens, ranking, session_len)
    M = (r[:, :, None] > r[:, None, :]) & vv
    pos = (r > 0)
    npos = pos.sum(-1)
    e = torch.exp(-z) * M
    loss = (((e.sum(2) + 1) * pos).clamp(min=1).log().sum(1) / npos).mean()
    if cal_diversity:
        bd = scores[:, :, None, :] - scores[:, None, :, :]             # float64 [B,L,L,K]
        ez = torch.exp(-z)
        up = ((ez[..., None] * (bd - z[..., None]) * M[..., None]).sum(2)) ** 2
        Aw = (weights * up).sum(-1)
        bo = 2 * (1 + (ez * M).sum(2)) ** 2
        div = -((Aw / bo * pos).sum(-1) / npos).mean()
        loss = (loss.double() + div * alpha).float()     # in-place += keeps float32 (Listloss.py:41)
    return loss


def mse_loss(ens, ranking, session_len, scores=None, weights=None, cal_diversity=0, alpha=0.01):
    """loss/MSEloss.py:22-30 (+ diversity :14-20): per-list mean squared error against the clamped labels."""
    L = ens.shape[1]
    valid = torch.arange(L, device=ens.device)[None, :] < session_len[:, None]
    r = ranking.clamp(min=0)
    n = valid.sum(-1)
    loss = ((((ens - r) ** 2) * valid).sum(-1) / n).mean()
    if cal_diversity:
        d = weights * ((scores - ens.unsqueeze(2)) ** 2)                  # float64 through the float64 base scores
        div = -((d * valid.unsqueeze(2)).sum(-1).sum(-1) / n).mean()
        loss = (loss.double() + div * alpha).float()                      # in-place += keeps float32 (MSEloss.py:29)
    return loss


def int_mse_loss(out, batch, cfg):
    """loss/IntMSEloss.py:16-21."""
    il, _, _ = intent_loss(out['intents'], batch['intents'], cfg.kl_weight, cfg.kl_temp)
    el = mse_loss(out['ens_score'], batch['ranking'], batch['session_len'], batch['scores'], out['weights'],
                  cfg.cal_diversity, cfg.diversity_alpha)
    return el * cfg.ensemble_weight + il * cfg.intent_weight, el, il


def intent_loss(pred, label, kl_weight=0.5, kl_temp=2.0):
    """loss/BaseIntloss.py:30-67.  ``label`` float64; CE uses the float64 label, KL the float32
    cast; per-class weights are all ones."""
    if float(pred.detach().min()) == 0.0:
        soft = pred + 1e-6
        soft = soft / soft.sum(-1, keepdim=True)
    else:
        soft = pred
    ce = -(((label > 0) * label * soft.log()) + ((label == 0) * (1 - soft).log())).sum(-1).mean()
    lab32 = label.float()
    kl_el = torch.where(lab32 > 0, lab32 * (lab32.log() - soft.log()), torch.zeros_like(lab32))
    kl = kl_el.double().sum(-1).mean() * kl_temp * kl_temp
    return ce * (1 - kl_weight) + kl * kl_weight, ce, kl


def int_bpr_loss(out, batch, cfg, noise):
    """loss/IntBPRloss.py:15-20."""
    il, _, _ = intent_loss(out['intents'], batch['intents'], cfg.kl_weight, cfg.kl_temp)
    el = bpr_loss(out['ens_score'], batch['ranking'], batch['session_len'], noise,
                  batch['scores'], out['weights'], cfg.cal_diversity, cfg.diversity_alpha)
    return el * cfg.ensemble_weight + il * cfg.intent_weight, el, il


def int_list_loss(out, batch, cfg):
    """loss/IntListloss.py:14-19."""
    il, _, _ = intent_loss(out['intents'], batch['intents'], cfg.kl_weight, cfg.kl_temp)
    el = list_loss(out['ens_score'], batch['ranking'], batch['session_len'],
                   batch['scores'], out['weights'], cfg.cal_diversity, cfg.diversity_alpha)
    return el * cfg.ensemble_weight + il * cfg.intent_weight, el, il


# ----------------------------------------------------------------------------------------
# optimizer semantics
# ----------------------------------------------------------------------------------------
def adam_groups(named_params, l2):
    """models/BaseModel.py:53-62 + helpers/BaseRunner.py:182-188: names containing 'bias' get no
    weight decay, everything else (LayerNorm weights and embedding tables included) gets ``l2``."""
    decay, no_decay = [], []
    for name, p in named_params:
        (no_decay if 'bias' in name else decay).append(p)
    return [{'params': decay, 'weight_decay': l2}, {'params': no_decay, 'weight_decay': 0.0}]


# ----------------------------------------------------------------------------------------
# evaluation
# ----------------------------------------------------------------------------------------
def evaluate_method(prediction_scores, ranking_lists, pos_nums, topk, metrics, session_len):
    """helpers/BaseRunner.py:56-131 in numpy: pad predictions with 0 and labels with -2 (->0) up to
    max(max(session_len), max(topk)); stable pre-sort by label; per-behaviour HR/NDCG with binary
    gains over 'the first all_pos columns'; overall NDCG@k with LINEAR gains (3/2/1/0)."""
    n = min(len(session_len), len(prediction_scores))
    session_len = np.asarray(session_len[:n])
    pos_nums = {k: np.asarray(v[:n]) for k, v in pos_nums.items()}
    width = int(max(session_len.max(), max(topk)))
    P = np.zeros((n, width), dtype=np.float64)
    R = np.full((n, width), -2, dtype=np.int64)
    for i in range(n):
        m = min(int(session_len[i]), len(prediction_scores[i]))
        P[i, :m] = np.asarray(prediction_scores[i][:m], dtype=np.float64)
        m2 = min(int(session_len[i]), len(ranking_lists[i]))
        R[i, :m2] = np.asarray(ranking_lists[i][:m2])
    order = np.argsort(R, axis=1)[:, ::-1]
    rows = np.arange(n)[:, None]
    R = R[rows, order]
    P = P[rows, order]
    R[R < 0] = 0
    asc = P.argsort(axis=1)
    disc = 1.0 / np.log2(np.arange(width) + 2.0)
    res = {}
    total_pos = np.sum(np.array(list(pos_nums.values())), axis=0).reshape(-1, 1)
    for btype, cnt in pos_nums.items():
        beh = btype.split('_')[1].split('num')[0]
        allp = total_pos if 'click' in btype else cnt.reshape(-1, 1)
        hitmat = asc < allp
        keep = np.nonzero(allp[:, 0] > 0)[0]
        hitmat, allp_k = hitmat[keep], allp[keep]
        for k in topk:
            kk = min(k, width)
            for metric in metrics:
                key = '%s_%s@%d' % (beh, metric, k)
                if metric == 'HR':
                    res[key] = (hitmat[:, -kk:].sum(1) > 0).mean()
                elif metric == 'NDCG':
                    if k == 1:
                        continue
                    dcg = (hitmat[:, -kk:] * disc[:kk][::-1]).sum(1)
                    ideal = (np.arange(kk).reshape(1, -1) < allp_k)
                    idcg = (ideal * disc[:kk]).sum(1)
                    res[key] = (dcg / idcg).mean()
                else:
                    raise ValueError('Undefined evaluation metric: {}.'.format(metric))
    desc = np.argsort(P, axis=1)[:, ::-1]
    Rs = R[rows, desc]
    Rp = np.sort(R, axis=1)[:, ::-1]
    for k in topk:
        dcg = (Rs[:, :k] * disc[:k]).sum(1)
        idcg = (Rp[:, :k] * disc[:k]).sum(1)
        res['NDCG@%d' % k] = (dcg / idcg).mean()
    return res


def ndcg_at_k(ens_score, ranking, session_len, k=3):
    """The overall ``NDCG@k`` key of evaluate_method for a padded [B,L] batch (numpy in/out)."""
    B = ens_score.shape[0]
    preds = [np.asarray(ens_score[i]) for i in range(B)]
    ranks = [np.asarray(ranking[i]) for i in range(B)]
    pos = {'c_x_i': np.ones(B, dtype=np.int64)}
    # evaluate_method's overall NDCG does not depend on pos_nums; reuse its padding/sorting path
    out = evaluate_method(preds, ranks, {'c_clicknum_i': np.ones(B, dtype=np.int64)}, [k], [],
                          np.asarray(session_len))
    del pos
    return out['NDCG@%d' % k]
