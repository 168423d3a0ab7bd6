"""autograd wrappers of the s1 HIP kernels (C ABI: evt_attn_prefixlm_*, evt_add_layernorm_*, evt_ce_sum_fwd_bwd,
evt_ce_rows_fwd_bwd)."""
import ctypes as C

import torch

from ..hip import lib as L


class PrefixLMAttentionFn(torch.autograd.Function):
    """softmax(QK^T/sqrt(d) + prefix-LM/padding mask) V on the packed in_proj output qkv [B, L, 3*E] -> [B, L, E].
    Replaces patched_mha_with_cache.py:441-454 + the mask tensor of t2s_model.py:456-479."""

    @staticmethod
    def forward(ctx, qkv, x_lens, y_lens, x_len, n_head, dropout_p, seed):
        B, Lq, E3 = qkv.shape
        E = E3 // 3
        D = E // n_head
        if not qkv.is_contiguous():
            raise L.EvtError("qkv must be contiguous")
        o = torch.empty((B, Lq, E), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B, n_head, Lq), dtype=torch.float32, device=qkv.device)
        p = L.AttnParams(L.dt_of(qkv), B, Lq, n_head, D, x_len, Lq * E3, E3, D, Lq * E, E, D, float(dropout_p),
                         int(seed) & 0xFFFFFFFF)
        esz = qkv.element_size()
        base = qkv.data_ptr()
        L.check(L.lib().evt_attn_prefixlm_fwd(C.byref(p), C.c_void_p(base), C.c_void_p(base + E * esz),
                                              C.c_void_p(base + 2 * E * esz), L.ptr(x_lens), L.ptr(y_lens), L.ptr(o),
                                              L.ptr(lse), L.stream_ptr()), "evt_attn_prefixlm_fwd")
        ctx.save_for_backward(qkv, o, lse, x_lens, y_lens)
        ctx.p = p
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, lse, x_lens, y_lens = ctx.saved_tensors
        p = ctx.p
        d_o = d_o.contiguous()
        B, Lq, E3 = qkv.shape
        E = E3 // 3
        dqkv = torch.empty_like(qkv)
        delta = torch.empty((B, p.H, Lq), dtype=torch.float32, device=qkv.device)
        esz = qkv.element_size()
        qb, gb = qkv.data_ptr(), dqkv.data_ptr()
        L.check(L.lib().evt_attn_prefixlm_bwd(
            C.byref(p), C.c_void_p(qb), C.c_void_p(qb + E * esz), C.c_void_p(qb + 2 * E * esz), L.ptr(o), L.ptr(d_o),
            L.ptr(lse), L.ptr(x_lens), L.ptr(y_lens), C.c_void_p(gb), C.c_void_p(gb + E * esz),
            C.c_void_p(gb + 2 * E * esz), L.ptr(delta), L.stream_ptr()), "evt_attn_prefixlm_bwd")
        return dqkv, None, None, None, None, None, None


class AddLayerNormFn(torch.autograd.Function):
    """LayerNorm(x + r) — the post-LN residual of transformer.py:311-315 in one pass each way."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps):
        x = x.contiguous()
        r = r.contiguous() if r is not None else None
        C_ = x.size(-1)
        rows = x.numel() // C_
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        L.check(L.lib().evt_add_layernorm_fwd(L.dt_of(x), L.ptr(x), L.ptr(r), L.ptr(gamma), L.ptr(beta), L.ptr(y),
                                              L.ptr(mean), L.ptr(rstd), C.c_int64(rows), C_, C.c_float(eps),
                                              L.stream_ptr()), "evt_add_layernorm_fwd")
        ctx.save_for_backward(x, r, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, r, gamma, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        C_ = x.size(-1)
        rows = x.numel() // C_
        dxr = torch.empty_like(x)
        dgamma = torch.zeros(C_, dtype=torch.float32, device=x.device)
        dbeta = torch.zeros(C_, dtype=torch.float32, device=x.device)
        L.check(L.lib().evt_add_layernorm_bwd(L.dt_of(x), L.ptr(x), L.ptr(r), L.ptr(gamma), L.ptr(dy), L.ptr(mean),
                                              L.ptr(rstd), L.ptr(dxr), L.ptr(dgamma), L.ptr(dbeta), C.c_int64(rows), C_,
                                              L.stream_ptr()), "evt_add_layernorm_bwd")
        return dxr, (dxr if r is not None else None), dgamma, dbeta, None


class CrossEntropySumFn(torch.autograd.Function):
    """F.cross_entropy(reduction="sum") over [rows, V] logits with the gradient and the top-k hit count produced by
    the same pass (t2s_model.py:486-489).  Returns (loss, hits int32[2] = (hits, counted rows))."""

    @staticmethod
    def forward(ctx, logits, targets, topk, ignore_index, V=None):
        """logits [rows, ld] with ld >= V (columns >= V are padding of the GEMM that produced them)"""
        logits = logits.contiguous()
        rows, ld = logits.shape
        V = ld if V is None else int(V)
        dlogits = torch.empty_like(logits)
        loss = torch.zeros(1, dtype=torch.float32, device=logits.device)
        hits = torch.zeros(2, dtype=torch.int32, device=logits.device)
        L.check(L.lib().evt_ce_sum_fwd_bwd_ld(L.dt_of(logits), L.ptr(logits), L.ptr(targets.contiguous()), L.ptr(dlogits),
                                              L.ptr(loss), L.ptr(hits), C.c_int64(rows), V, C.c_int64(ld), int(topk),
                                              C.c_int64(int(ignore_index)), C.c_float(1.0), L.stream_ptr()),
                "evt_ce_sum_fwd_bwd_ld")
        ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(hits)
        return loss[0], hits

    @staticmethod
    def backward(ctx, dloss, _):
        (dlogits,) = ctx.saved_tensors
        return dlogits * dloss.to(dlogits.dtype), None, None, None, None


class CrossEntropyRowsFn(torch.autograd.Function):
    """Per-row -log p(target) over [rows, V] logits, with softmax-onehot saved by the same pass: the DPO branch needs
    both the summed cross-entropy and per-sequence target log-probabilities of the same logits (t2s_model.py:420-427,
    get_batch_logps models/utils.py:176-183).  Returns (row_loss fp32 [rows], hits int32[2])."""

    @staticmethod
    def forward(ctx, logits, targets, topk, ignore_index, V=None):
        logits = logits.contiguous()
        rows, ld = logits.shape
        V = ld if V is None else int(V)
        dlogits = torch.empty_like(logits)
        row_loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
        hits = torch.zeros(2, dtype=torch.int32, device=logits.device)
        L.check(L.lib().evt_ce_rows_fwd_bwd_ld(L.dt_of(logits), L.ptr(logits), L.ptr(targets.contiguous()), L.ptr(dlogits),
                                               L.ptr(row_loss), L.ptr(hits), C.c_int64(rows), V, C.c_int64(ld), int(topk),
                                               C.c_int64(int(ignore_index)), L.stream_ptr()), "evt_ce_rows_fwd_bwd_ld")
        ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(hits)
        return row_loss, hits

    @staticmethod
    def backward(ctx, drow, _):
        (dlogits,) = ctx.saved_tensors
        return dlogits * drow.to(dlogits.dtype).unsqueeze(1), None, None, None, None
