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
# through the op.  The whole-model forward / backward keep their struct-based entry points (intel_forward / intel_backward:
# model.py drives them through one autograd.Function).
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

    return ['linear', 'linear_dgrad', 'linear_wgrad', 'attention', 'attention_bwd', 'add_layernorm', 'ndcg']


try:
    REGISTERED_OPS = _register()
except Exception as _e:        # an older torch without torch.library.custom_op: the ctypes wrappers above still work
    REGISTERED_OPS = []
    _REGISTER_ERROR = _e
