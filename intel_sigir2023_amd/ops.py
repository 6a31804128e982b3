"""Thin tensor-level wrappers over the building-block entry points (used by tests and the model)."""
import torch

from . import _lib as L


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def linear(x, w, bias=None, relu=False):
    L.require_gpu(x)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    nb = L.lib().intel_op_workspace_bytes(M, N, K)
    ws = _ws(nb, x.device)
    L.check(L.lib().intel_op_linear(L.ptr(x), M, K, L.ptr(w), N, L.ptr(bias), int(relu), L.ptr(y), L.ptr(ws), nb,
                                    L.stream_ptr(x.device)), 'intel_op_linear')
    return y


def linear_dgrad(dy, w):
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
    nb = L.lib().intel_op_workspace_bytes(M, N, K)
    ws = _ws(nb, dy.device)
    L.check(L.lib().intel_op_linear_dgrad(L.ptr(dy), M, N, L.ptr(w), K, L.ptr(dx), L.ptr(ws), nb, L.stream_ptr(dy.device)),
            'intel_op_linear_dgrad')
    return dx


def linear_wgrad(dy, x, want_bias=True):
    M, N = dy.shape
    K = x.shape[1]
    dw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if want_bias else None
    nb = L.lib().intel_op_workspace_bytes(M, N, K)
    ws = _ws(nb, dy.device)
    L.check(L.lib().intel_op_linear_wgrad(L.ptr(dy), L.ptr(x), M, N, K, L.ptr(dw), L.ptr(db), L.ptr(ws), nb,
                                          L.stream_ptr(dy.device)), 'intel_op_linear_wgrad')
    return dw, db


def linear_bwd(dy, x, w, relu_mask=False, want_bias=True):
    """Both products of a [d -> d] nn.Linear's backward in one pass over the rows (csrc/pair.hip): dx = (dy @ w) [* (x > 0)], dw = dy^T x,
    db = colsum(dy)."""
    M, d = dy.shape
    dx = torch.empty(M, d, dtype=torch.float32, device=dy.device)
    dw = torch.empty(d, d, dtype=torch.float32, device=dy.device)
    db = torch.empty(d, dtype=torch.float32, device=dy.device) if want_bias else None
    nb = L.lib().intel_op_linear_bwd_workspace_bytes(M, d)
    ws = _ws(nb, dy.device)
    L.check(L.lib().intel_op_linear_bwd(L.ptr(dy), L.ptr(x), M, d, L.ptr(w), 1 if relu_mask else 0, L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(ws), nb,
                                        L.stream_ptr(dy.device)), 'intel_op_linear_bwd')
    return dx, dw, db


def linear_bwd_qkv(dy, x, w, res=None, want_bias=False):
    """The backward of a fused q/k/v projection in one pass over the rows (csrc/pair.hip): dy [M, nb*d], w [nb*d, d] stacked, x [M, d]:
    dx = dy @ w (+ res), dw = dy^T x, db = colsum(dy)."""
    M, n = dy.shape
    d = x.shape[1]
    nb = n // d
    dx = torch.empty(M, d, dtype=torch.float32, device=dy.device)
    dw = torch.empty(n, d, dtype=torch.float32, device=dy.device)
    db = torch.empty(n, dtype=torch.float32, device=dy.device) if want_bias else None
    nbytes = L.lib().intel_op_linear_bwd_qkv_workspace_bytes(M, d, nb)
    ws = _ws(nbytes, dy.device)
    L.check(L.lib().intel_op_linear_bwd_qkv(L.ptr(dy), L.ptr(x), L.ptr(res), M, d, nb, L.ptr(w), L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(ws), nbytes,
                                            L.stream_ptr(dy.device)), 'intel_op_linear_bwd_qkv')
    return dx, dw, db


def attention(qkv, B, T, d, heads, key_len=None):
    out = torch.empty(B * T, d, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B * heads * T, dtype=torch.float32, device=qkv.device)
    L.check(L.lib().intel_op_attention(L.ptr(qkv), B, T, d, heads, L.ptr(key_len), L.ptr(out), L.ptr(lse),
                                       L.stream_ptr(qkv.device)), 'intel_op_attention')
    return out, lse


def attention_bwd(qkv, out, dout, lse, B, T, d, heads, key_len=None):
    dqkv = torch.empty_like(qkv)
    nbytes = L.lib().intel_op_attention_bwd_workspace_bytes(B, T, d, heads)
    dsum = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=qkv.device)
    L.check(L.lib().intel_op_attention_bwd(L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(lse), B, T, d, heads,
                                           L.ptr(key_len), L.ptr(dqkv), L.ptr(dsum), L.stream_ptr(qkv.device)),
            'intel_op_attention_bwd')
    return dqkv


def add_layernorm(x, r, gamma, beta, stash=False):
    M, N = x.shape
    y = torch.empty_like(x)
    xhat = torch.empty_like(x) if stash else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if stash else None
    L.check(L.lib().intel_op_add_layernorm(L.ptr(x), L.ptr(r), M, N, L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(xhat),
                                           L.ptr(rstd), L.stream_ptr(x.device)), 'intel_op_add_layernorm')
    return (y, xhat, rstd) if stash else y


# ------------------------------------------------------------------------------------------------------------------------
# torch.library registration: the same entry points as dispatcher-visible custom ops, `torch.ops.intel_mi355x.*`
# (north_star: "exposed ... as PyTorch-ROCm custom ops").  The C ABI stays the lowest layer; these are
# thin schemas over it with shape ("fake") functions for tracing and autograd formulas where the reference differentiates
# through the op.  The whole model is registered too: torch.ops.intel_mi355x.intel_forward (+ intel_backward as its autograd
# formula) over the struct-based C entry points; model.forward takes it with INTEL_MODEL_OP=1 (ops.model_forward), otherwise the
# equivalent autograd.Function of model.py.
# ------------------------------------------------------------------------------------------------------------------------
NAMESPACE = 'intel_mi355x'


def _register():
    from torch.library import custom_op

    @custom_op(NAMESPACE + '::linear', mutates_args=(), device_types='cuda')
    def op_linear(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, relu: bool) -> torch.Tensor:
        return linear(x.contiguous(), w.contiguous(), bias if bias.numel() else None, relu)

    @op_linear.register_fake
    def _(x, w, bias, relu):
        return x.new_empty(x.shape[0], w.shape[0])

    @custom_op(NAMESPACE + '::linear_dgrad', mutates_args=(), device_types='cuda')
    def op_linear_dgrad(dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        return linear_dgrad(dy.contiguous(), w.contiguous())

    @op_linear_dgrad.register_fake
    def _(dy, w):
        return dy.new_empty(dy.shape[0], w.shape[1])

    @custom_op(NAMESPACE + '::linear_wgrad', mutates_args=(), device_types='cuda')
    def op_linear_wgrad(dy: torch.Tensor, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        return linear_wgrad(dy.contiguous(), x.contiguous(), True)

    @op_linear_wgrad.register_fake
    def _(dy, x):
        return dy.new_empty(dy.shape[1], x.shape[1]), dy.new_empty(dy.shape[1])

    def _linear_setup(ctx, inputs, output):
        x, w, bias, relu = inputs
        ctx.save_for_backward(x, w, output)
        ctx.relu, ctx.has_bias = relu, bias.numel() > 0

    def _linear_backward(ctx, g):
        x, w, y = ctx.saved_tensors
        g = g.contiguous()
        if ctx.relu:
            g = g * (y > 0)
        dw, db = torch.ops.intel_mi355x.linear_wgrad(g, x)
        return torch.ops.intel_mi355x.linear_dgrad(g, w), dw, (db if ctx.has_bias else None), None

    op_linear.register_autograd(_linear_backward, setup_context=_linear_setup)

    @custom_op(NAMESPACE + '::attention', mutates_args=(), device_types='cuda')
    def op_attention(qkv: torch.Tensor, B: int, T: int, d: int, heads: int) -> tuple[torch.Tensor, torch.Tensor]:
        return attention(qkv.contiguous(), B, T, d, heads)

    @op_attention.register_fake
    def _(qkv, B, T, d, heads):
        return qkv.new_empty(B * T, d), qkv.new_empty(B * heads * T)

    @custom_op(NAMESPACE + '::attention_bwd', mutates_args=(), device_types='cuda')
    def op_attention_bwd(qkv: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor, B: int, T: int, d: int,
                         heads: int) -> torch.Tensor:
        return attention_bwd(qkv.contiguous(), out.contiguous(), dout.contiguous(), lse.contiguous(), B, T, d, heads)

    @op_attention_bwd.register_fake
    def _(qkv, out, dout, lse, B, T, d, heads):
        return torch.empty_like(qkv)

    def _attn_setup(ctx, inputs, output):
        qkv, B, T, d, heads = inputs
        ctx.save_for_backward(qkv, output[0], output[1])
        ctx.shape = (B, T, d, heads)

    def _attn_backward(ctx, g_out, g_lse):
        qkv, out, lse = ctx.saved_tensors
        return torch.ops.intel_mi355x.attention_bwd(qkv, out, g_out.contiguous(), lse, *ctx.shape), None, None, None, None

    op_attention.register_autograd(_attn_backward, setup_context=_attn_setup)

    @custom_op(NAMESPACE + '::add_layernorm', mutates_args=(), device_types='cuda')
    def op_add_layernorm(x: torch.Tensor, r: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
        return add_layernorm(x.contiguous(), r.contiguous(), gamma, beta)

    @op_add_layernorm.register_fake
    def _(x, r, gamma, beta):
        return torch.empty_like(x)

    @custom_op(NAMESPACE + '::ndcg', mutates_args=(), device_types='cuda')
    def op_ndcg(ens_score: torch.Tensor, ranking: torch.Tensor, session_len: torch.Tensor, k: int) -> torch.Tensor:
        B, Lm = ens_score.shape
        out = torch.empty(B, dtype=torch.float32, device=ens_score.device)
        L.check(L.lib().intel_ndcg(B, Lm, k, L.ptr(ens_score.contiguous()), L.ptr(ranking.to(torch.int32).contiguous()),
                                   L.ptr(session_len.to(torch.int32).contiguous()), L.ptr(out), L.stream_ptr(ens_score.device)), 'intel_ndcg')
        return out

    @op_ndcg.register_fake
    def _(ens_score, ranking, session_len, k):
        return ens_score.new_empty(ens_score.shape[0])

    # ---- the whole model (models/IntEL/IntEL.py:117-124: IntEL.forward) as ONE dispatcher-visible op + its hand-written backward.
    # `handle` names the module (model_handle(model)): the op reads the hyper-parameters and the workspace from it; the parameter
    # tensors travel as a Tensor[] argument in the order of model.slot_items(), so autograd sees them and receives their gradients.
    # EAGER-ONLY: the op keeps hidden per-module state that its schema does not declare (the workspace with the activation stash, the
    # one-slot _op_stash that intel_backward continues from, _generation) -- exactly the state the reference's autograd graph would hold.
    # A tracing compiler (torch.compile / functionalization) may reorder, deduplicate or re-run a "pure" op and would corrupt that
    # stash: do not trace it.  One training forward per backward, as in the reference's loop (BaseRunner.py:283-289).
    from typing import List, Optional

    @custom_op(NAMESPACE + '::intel_forward', mutates_args=(), device_types='cuda')
    def op_intel_forward(handle: int, train: bool, i_id_s: torch.Tensor, i_class_c: torch.Tensor, scores: torch.Tensor, session_len: torch.Tensor,
                         u_id_c: torch.Tensor, context_mh: torch.Tensor, his_context_mh: torch.Tensor, his_intents: torch.Tensor,
                         history_len: torch.Tensor, his_item_id: torch.Tensor, his_item_idx: Optional[torch.Tensor],
                         his_item_int: Optional[torch.Tensor], history_item_len: torch.Tensor, his_rows: int, hisitem_rows: int,
                         params: List[torch.Tensor]) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        model = _model_of(handle)
        data = {'i_id_s': i_id_s, 'i_class_c': i_class_c, 'scores': scores, 'session_len': session_len, 'u_id_c': u_id_c,
                'context_mh': context_mh, 'his_context_mh': his_context_mh, 'his_intents': his_intents, 'history_len': history_len,
                'his_item_id': his_item_id, 'history_item_len': history_item_len}
        if his_item_idx is not None:
            data['his_item_idx'] = his_item_idx
        else:
            data['his_item_int'] = his_item_int
        if his_rows >= 0 and hisitem_rows >= 0:
            data['his_rows'], data['hisitem_rows'] = his_rows, hisitem_rows
        batch, keep = model.prepare_batch(data)
        out = model.run_forward(batch, keep, [p.detach() for p in params], train=train)
        model._generation = getattr(model, '_generation', 0) + 1
        if train:
            model._op_stash = (model._generation, batch, keep)      # what intel_backward continues from (one forward / backward pair at a time)
        return out

    @op_intel_forward.register_fake
    def _(handle, train, i_id_s, i_class_c, scores, session_len, u_id_c, context_mh, his_context_mh, his_intents, history_len, his_item_id,
          his_item_idx, his_item_int, history_item_len, his_rows, hisitem_rows, params):
        model = _model_of(handle)
        B, Lm = i_id_s.shape
        f = dict(dtype=torch.float32, device=i_id_s.device)
        return torch.empty(B, Lm, model.model_num, **f), torch.empty(B, Lm, **f), torch.empty(B, model.intent_num, **f)

    @custom_op(NAMESPACE + '::intel_backward', mutates_args=(), device_types='cuda')
    def op_intel_backward(handle: int, d_weights: torch.Tensor, d_ens_score: torch.Tensor, d_intents: torch.Tensor,
                          params: List[torch.Tensor]) -> List[torch.Tensor]:
        model = _model_of(handle)
        stash = getattr(model, '_op_stash', None)
        if stash is None or stash[0] != model._generation:
            raise L.IntelHipError('intel_backward: no matching intel_forward(train=True) (the activation stash was overwritten)')
        _, batch, keep = stash
        grads = model.run_backward(batch, keep, [p.detach() for p in params], d_weights.contiguous().float(), d_ens_score.contiguous().float(),
                                   d_intents.contiguous().float())
        items = model.slot_items()
        return [grads[s] if s in grads else torch.zeros_like(p) for (s, _, _), p in zip(items, params)]

    @op_intel_backward.register_fake
    def _(handle, d_weights, d_ens_score, d_intents, params):
        return [torch.empty_like(p) for p in params]

    def _model_setup(ctx, inputs, output):
        ctx.handle = inputs[0]
        ctx.n_inputs = len(inputs)
        ctx.save_for_backward(*inputs[-1])

    def _model_backward(ctx, d_weights, d_ens, d_intents):
        params = list(ctx.saved_tensors)
        model = _model_of(ctx.handle)
        B, Lm = model._op_stash[1].B, model._op_stash[1].L
        dev = params[0].device
        zw = lambda t, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if t is None else t
        grads = torch.ops.intel_mi355x.intel_backward(ctx.handle, zw(d_weights, (B, Lm, model.model_num)), zw(d_ens, (B, Lm)),
                                                      zw(d_intents, (B, model.intent_num)), params)
        return (None,) * (ctx.n_inputs - 1) + (list(grads),)

    op_intel_forward.register_autograd(_model_backward, setup_context=_model_setup)

    return ['linear', 'linear_dgrad', 'linear_wgrad', 'attention', 'attention_bwd', 'add_layernorm', 'ndcg', 'intel_forward', 'intel_backward']


_HANDLES = {}


def model_handle(model):
    """Integer handle of an IntEL module for torch.ops.intel_mi355x.intel_forward / intel_backward (a weak reference is kept)."""
    import weakref
    h = id(model)
    _HANDLES[h] = weakref.ref(model)
    return h


def _model_of(handle):
    ref = _HANDLES.get(int(handle))
    model = ref() if ref is not None else None
    if model is None:
        raise L.IntelHipError('intel_forward: unknown or dead model handle %r (ops.model_handle(model))' % (handle,))
    return model


def model_forward(model, data):
    """IntEL.forward(data) through the dispatcher: torch.ops.intel_mi355x.intel_forward over the module's parameters
    (models/IntEL/IntEL.py:117-124).  Differentiable: the registered autograd formula is intel_backward."""
    if 'intel_forward' not in REGISTERED_OPS:
        raise L.IntelHipError('torch.library.custom_op is not available in this torch: %r' % (globals().get('_REGISTER_ERROR'),))
    params = [p for _, _, p in model.slot_items()]
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    i32 = lambda k: data[k]
    w, e, i = torch.ops.intel_mi355x.intel_forward(
        model_handle(model), bool(need_grad), i32('i_id_s'), data.get('i_class_c') if data.get('i_class_c') is not None else torch.zeros_like(data['i_id_s']),
        data['scores'], i32('session_len'), i32('u_id_c'), i32('context_mh'), i32('his_context_mh'), data['his_intents'], i32('history_len'),
        i32('his_item_id'), data.get('his_item_idx'), data.get('his_item_int') if 'his_item_idx' not in data else None, i32('history_item_len'),
        int(data.get('his_rows', -1)), int(data.get('hisitem_rows', -1)), params)
    return {'weights': w, 'ens_score': e, 'intents': i}


try:
    REGISTERED_OPS = _register()
except Exception as _e:        # an older torch without torch.library.custom_op: the ctypes wrappers above still work
    REGISTERED_OPS = []
    _REGISTER_ERROR = _e
