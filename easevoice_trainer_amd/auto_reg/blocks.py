"""The two sub-blocks of the s1 post-LN transformer layer (transformer.py:311-334 of the reference) as ONE autograd node
each, built from the library's launches:

  attention block   x -> in_proj GEMM -> prefix-LM attention -> out_proj GEMM -> LayerNorm(x + dropout(.))
  FFN block         x -> linear1 GEMM [relu + dropout in its store] -> linear2 GEMM -> LayerNorm(x + dropout(.))

What being one node buys: the sum of the residual branch's gradient and the branch's own input gradient is the
add-epilogue of the first GEMM's backward-data launch (no element-wise add), the relu + dropout derivative of the FFN is
the gate-epilogue of linear2's backward-data launch read off the saved activation (no relu_dropout_bwd pass, no mask
regeneration), and the inner relu + dropout is linear1's store (no relu_dropout_fwd pass).  Parameter gradients go
straight into the engine's fp32 arena views (hip/linear.py::gemm_bwd_weight, hip/enc.py::grad_sink)."""
import ctypes as C

import torch

from ..hip import lib as L
from ..hip.enc import grad_sink, rng_counter
from ..hip.linear import gemm_bwd_data, gemm_bwd_weight, gemm_fwd


def _slot(w):
    s = getattr(w, "_evt_slot", None)
    if s is None:
        raise L.EvtError("s1 block: the weight is not attached to a LinearBank (S1Engine builds it); no eager fallback")
    return s


def _res_drop_ln_fwd(x, y, gamma, beta, p, site, eps):
    Cc = x.size(-1)
    rows = x.numel() // Cc
    out = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    L.check(L.lib().evt_res_dropout_ln_fwd(L.dt_of(x), L.ptr(x), L.ptr(y), L.ptr(gamma), L.ptr(beta), None, x.size(-2),
                                           C.c_float(p), L.ptr(rng_counter(x.device)), C.c_uint32(site), L.ptr(out),
                                           L.ptr(mean), L.ptr(rstd), C.c_int64(rows), Cc, C.c_float(eps), L.stream_ptr()),
            "evt_res_dropout_ln_fwd")
    return out, mean, rstd


def _res_drop_ln_bwd(x, y, gamma, beta_param, gamma_param, dout, mean, rstd, p, site):
    """-> (dx of the residual branch, dy of the dropped branch, dgamma, dbeta); the last two are None when the kernel
    accumulated into the arena views"""
    Cc = x.size(-1)
    rows = x.numel() // Cc
    dx = torch.empty_like(x)
    dy = torch.empty_like(x) if p > 0.0 else None
    sg, sb = grad_sink(gamma_param), grad_sink(beta_param)
    sunk = sg is not None and sb is not None
    if sunk:
        dgamma, dbeta = sg, sb
    else:
        dgb = torch.zeros(2, Cc, dtype=torch.float32, device=x.device)
        dgamma, dbeta = dgb[0], dgb[1]
    L.check(L.lib().evt_res_dropout_ln_bwd(L.dt_of(x), L.ptr(x), L.ptr(y), L.ptr(gamma), L.ptr(dout), L.ptr(mean),
                                           L.ptr(rstd), None, x.size(-2), C.c_float(p), L.ptr(rng_counter(x.device)),
                                           C.c_uint32(site), L.ptr(dx), L.ptr(dy), L.ptr(dgamma), L.ptr(dbeta),
                                           C.c_int64(rows), Cc, L.stream_ptr()), "evt_res_dropout_ln_bwd")
    return dx, (dy if dy is not None else dx), (None if sunk else dgamma), (None if sunk else dbeta)


def _wgrad(slot, x, dy, need_w, need_b):
    """(dw, db) of one dense layer, nothing launched when neither is wanted"""
    if not (need_w or need_b):
        return None, None
    dw, db = gemm_bwd_weight(slot, x, dy, want_bias=bool(need_b))
    return (dw if need_w else None), db


class FFNBlockFn(torch.autograd.Function):
    """out = LayerNorm(x + dropout(linear2(dropout(relu(linear1(x))))))"""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, p, site_inner, site_res, eps):
        s1, s2 = _slot(w1), _slot(w2)
        x = x.contiguous()
        h = gemm_fwd(s1, x, relu=True, drop=(p, site_inner))
        ff = gemm_fwd(s2, h)
        out, mean, rstd = _res_drop_ln_fwd(x, ff, gamma, beta, p, site_res, eps)
        ctx.save_for_backward(x, h, ff, gamma, mean, rstd)
        ctx.cfg = (s1, s2, p, site_res)
        ctx.params = (gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h, ff, gamma, mean, rstd = ctx.saved_tensors
        s1, s2, p, site_res = ctx.cfg
        g_param, b_param = ctx.params
        dx_res, dff, dgamma, dbeta = _res_drop_ln_bwd(x, ff, gamma, b_param, g_param, dout.contiguous(), mean, rstd, p,
                                                      site_res)
        need = ctx.needs_input_grad          # x, w1, b1, w2, b2, ...: a frozen layer gets no gradient launch (the bias
        dw2, db2 = _wgrad(s2, h, dff, need[3], need[4])          # gradient comes out of the weight-gradient launch)
        # d(linear1 output before relu) = (dff W2) * dropout multiplier * (pre-activation > 0): both read off h
        dz1 = gemm_bwd_data(s2, dff, gate=h, gate_pos=1.0 / (1.0 - p) if p > 0.0 else 1.0)
        dw1, db1 = _wgrad(s1, x, dz1, need[1], need[2])
        dx = gemm_bwd_data(s1, dz1, add=dx_res) if need[0] else None
        return dx, dw1, db1, dw2, db2, dgamma, dbeta, None, None, None, None


class AttnBlockFn(torch.autograd.Function):
    """out = LayerNorm(x + dropout(out_proj(prefix_lm_attention(in_proj(x)))))"""

    @staticmethod
    def forward(ctx, x, w_in, b_in, w_out, b_out, gamma, beta, x_lens, y_lens, x_len, n_head, p_attn, seed, p, site, eps):
        si, so = _slot(w_in), _slot(w_out)
        x = x.contiguous()
        B, Lq, E = x.shape
        qkv = gemm_fwd(si, x)
        D = E // n_head
        E3 = 3 * E
        o = torch.empty((B, Lq, E), dtype=x.dtype, device=x.device)
        lse = torch.empty((B, n_head, Lq), dtype=torch.float32, device=x.device)
        ap = L.AttnParams(L.dt_of(x), B, Lq, n_head, D, x_len, Lq * E3, E3, D, Lq * E, E, D, float(p_attn),
                          int(seed) & 0xFFFFFFFF)
        esz, base = qkv.element_size(), qkv.data_ptr()
        L.check(L.lib().evt_attn_prefixlm_fwd(C.byref(ap), C.c_void_p(base), C.c_void_p(base + E * esz),
                                              C.c_void_p(base + 2 * E * esz), L.ptr(x_lens), L.ptr(y_lens), L.ptr(o),
                                              L.ptr(lse), L.stream_ptr()), "evt_attn_prefixlm_fwd")
        sa = gemm_fwd(so, o)
        out, mean, rstd = _res_drop_ln_fwd(x, sa, gamma, beta, p, site, eps)
        ctx.save_for_backward(x, qkv, o, lse, sa, gamma, mean, rstd, x_lens, y_lens)
        ctx.cfg = (si, so, ap, p, site)
        ctx.params = (gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, qkv, o, lse, sa, gamma, mean, rstd, x_lens, y_lens = ctx.saved_tensors
        si, so, ap, p, site = ctx.cfg
        g_param, b_param = ctx.params
        dx_res, dsa, dgamma, dbeta = _res_drop_ln_bwd(x, sa, gamma, b_param, g_param, dout.contiguous(), mean, rstd, p, site)
        need = ctx.needs_input_grad          # x, w_in, b_in, w_out, b_out, ...
        dwo, dbo = _wgrad(so, o, dsa, need[3], need[4])
        d_o = gemm_bwd_data(so, dsa)
        B, Lq, E = x.shape
        dqkv = torch.empty_like(qkv)
        delta = torch.empty((B, ap.H, Lq), dtype=torch.float32, device=x.device)
        esz, qb, gb = qkv.element_size(), qkv.data_ptr(), dqkv.data_ptr()
        if si.bank.side_on:
            si.bank.flush_pending()      # the queued dW GEMMs of this layer (and in_proj's of the one above) beside the attention kernels
        L.check(L.lib().evt_attn_prefixlm_bwd(
            C.byref(ap), C.c_void_p(qb), C.c_void_p(qb + E * esz), C.c_void_p(qb + 2 * E * esz), L.ptr(o), L.ptr(d_o),
            L.ptr(lse), L.ptr(x_lens), L.ptr(y_lens), C.c_void_p(gb), C.c_void_p(gb + E * esz),
            C.c_void_p(gb + 2 * E * esz), L.ptr(delta), L.stream_ptr()), "evt_attn_prefixlm_bwd")
        dwi, dbi = _wgrad(si, x, dqkv, need[1], need[2])
        dx = gemm_bwd_data(si, dqkv, add=dx_res) if need[0] else None
        return dx, dwi, dbi, dwo, dbo, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def ffn_block(x, linear1, linear2, norm, p, site_inner, site_res):
    return FFNBlockFn.apply(x, linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm.weight, norm.bias, float(p),
                            int(site_inner), int(site_res), float(norm.eps))


def attn_block(x, attn, norm, x_lens, y_lens, x_len, seed, p, site):
    p_attn = attn.dropout if attn.training else 0.0
    return AttnBlockFn.apply(x, attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias,
                             norm.weight, norm.bias, x_lens, y_lens, int(x_len), int(attn.num_heads), float(p_attn),
                             int(seed), float(p), int(site), float(norm.eps))
