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
