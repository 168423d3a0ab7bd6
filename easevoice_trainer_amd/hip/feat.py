"""Bindings of the feature extractors' own launches (csrc/feat_ops.hip) and the LayerNorm entry point they share with the s1
blocks.  Inference only: nothing here records an autograd graph."""
import ctypes as C

import torch

from . import lib as L


def gelu_rows(x, bias=None, t_out=None):
    """x [B, T, C] -> gelu(x + bias)[:, :t_out] (erf form), one launch; bias fp32 [C] or None"""
    if x.dim() != 3 or not x.is_contiguous():
        raise L.EvtError(f"gelu_rows: contiguous [B, T, C] expected, got {tuple(x.shape)}")
    B, T, Cc = x.shape
    t_out = T if t_out is None else int(t_out)
    out = torch.empty((B, t_out, Cc), dtype=x.dtype, device=x.device)
    L.check(L.lib().evt_gelu_rows_fwd(L.dt_of(x), L.ptr(x), L.ptr(bias), L.ptr(out), C.c_int64(B), T, t_out, Cc,
                                      L.stream_ptr()), "evt_gelu_rows_fwd")
    return out


def channel_norm_gelu(x, gamma, beta, eps, gelu=True):
    """GroupNorm with one channel per group over the frames of [B, T, C] rows (+ GELU), one launch"""
    if x.dim() != 3 or not x.is_contiguous():
        raise L.EvtError(f"channel_norm_gelu: contiguous [B, T, C] expected, got {tuple(x.shape)}")
    B, T, Cc = x.shape
    out = torch.empty_like(x)
    L.check(L.lib().evt_channel_norm_gelu_fwd(L.dt_of(x), L.ptr(x), L.ptr(gamma), L.ptr(beta), C.c_float(eps), L.ptr(out), B,
                                              T, Cc, 1 if gelu else 0, L.stream_ptr()), "evt_channel_norm_gelu_fwd")
    return out


def add_layernorm(x, r, gamma, beta, eps):
    """LayerNorm(x + r) over the last axis (r may be None): evt_add_layernorm_fwd, statistics discarded"""
    x = x.contiguous()
    r = r.contiguous() if r is not None else None
    Cc = x.size(-1)
    rows = x.numel() // Cc
    y = torch.empty_like(x)
    stats = torch.empty(2, rows, dtype=torch.float32, device=x.device)
    L.check(L.lib().evt_add_layernorm_fwd(L.dt_of(x), L.ptr(x), L.ptr(r), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(stats[0]),
                                          L.ptr(stats[1]), C.c_int64(rows), Cc, C.c_float(eps), L.stream_ptr()),
            "evt_add_layernorm_fwd")
    return y
