"""The WaveNet-style gated stack of the posterior encoder and the flow (WN, src/easevoice/module/modules.py:135-212) as ONE
autograd node per stack.

Per layer the arithmetic is the reference's: in_layer conv (k = 5) -> tanh * sigmoid gate with the conditioning slice ->
res_skip conv (1x1) -> x <- (x + rs[:H]) * mask, out <- out + rs[H:] (last layer: out <- (out + rs) * mask).  The launches
are the library's (evt_conv1d_*, evt_gated_act_*, evt_wn_residual_*), five forward and eight backward per layer; what the
single node removes is everything between them that torch issued: the element-wise sum of the two gradient branches of
every layer input (now the add-epilogue of the in_layer backward-data launch), a zero fill and a cast per layer for the
conditioning gradient (one fp32 buffer and one cast per stack), the unbind / stack bookkeeping, and ~80 autograd nodes.
"""
import ctypes as C

import torch

from . import conv as HC
from . import lib as L


class WNStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g_lbh, lens, anchor, in_slots, rs_slots, hidden):
        """x [B, T, H] compute dtype, contiguous; g_lbh [n_layers, B, 2H] (same dtype) or None; lens [B] int32."""
        n_layers, H = len(in_slots), hidden
        B, T, _ = x.shape
        if not x.is_contiguous() or x.size(2) != H or (g_lbh is not None and not g_lbh.is_contiguous()):
            raise L.EvtError("WN stack: contiguous [B, T, H] input and [n_layers, B, 2H] conditioning expected")
        dt = L.dt_of(x)
        rows = B * T
        saved = []
        out = None
        for i in range(n_layers):
            last = i == n_layers - 1
            x_in = HC._fwd(in_slots[i], x, None, 1.0, L.ACT_NONE, 1.0)
            acts = torch.empty((B, T, H), dtype=x.dtype, device=x.device)
            L.check(L.lib().evt_gated_act_fwd(dt, L.ptr(x_in), L.ptr(g_lbh[i] if g_lbh is not None else None), L.ptr(acts),
                                              B, T, H, L.stream_ptr()), "evt_gated_act_fwd")
            rs = HC._fwd(rs_slots[i], acts, None, 1.0, L.ACT_NONE, 1.0)
            acc_out = torch.empty((B, T, H), dtype=x.dtype, device=x.device)
            x_out = None if last else torch.empty_like(acc_out)
            L.check(L.lib().evt_wn_residual_fwd(dt, L.ptr(None if last else x), L.ptr(rs), L.ptr(out), L.ptr(lens), T,
                                                L.ptr(x_out), L.ptr(acc_out), C.c_int64(rows), H, int(last),
                                                L.stream_ptr()), "evt_wn_residual_fwd")
            saved += [x, x_in, acts]
            x, out = x_out, acc_out
        ctx.save_for_backward(lens, g_lbh, *saved)
        ctx.cfg = (in_slots, rs_slots, H, B, T)
        return out

    @staticmethod
    def backward(ctx, dout):
        lens, g_lbh, *saved = ctx.saved_tensors
        in_slots, rs_slots, H, B, T = ctx.cfg
        n_layers = len(in_slots)
        dtype, dev = dout.dtype, dout.device
        dt = L.dt_code(dtype)
        rows = B * T
        dacc = dout.contiguous()
        dx_next = None
        dg32 = torch.zeros((n_layers, B, 2 * H), dtype=torch.float32, device=dev) if g_lbh is not None else None
        for i in reversed(range(n_layers)):
            last = i == n_layers - 1
            x, x_in, acts = saved[3 * i: 3 * i + 3]
            rs_w = H if last else 2 * H
            drs = torch.empty((B, T, rs_w), dtype=dtype, device=dev)
            dx_res = None if last else torch.empty((B, T, H), dtype=dtype, device=dev)
            L.check(L.lib().evt_wn_residual_bwd(dt, L.ptr(dx_next), L.ptr(dacc), L.ptr(lens), T, L.ptr(dx_res), L.ptr(drs),
                                                C.c_int64(rows), H, int(last), L.stream_ptr()), "evt_wn_residual_bwd")
            if last:
                dacc = drs          # out = (acc + rs) * mask: the skip sum's gradient is masked too
            s = rs_slots[i]
            if s.bank.weight_grads:
                HC._bwd_weight(s, acts, drs, None, B, T, 1.0, L.ACT_NONE, 1.0)
            dacts = HC._bwd_data(s, drs, None, acts, None, B, T, 1.0, L.ACT_NONE, 1.0)
            dx_in = torch.empty_like(x_in)
            L.check(L.lib().evt_gated_act_bwd(dt, L.ptr(x_in), L.ptr(g_lbh[i] if g_lbh is not None else None), L.ptr(dacts),
                                              L.ptr(dx_in), L.ptr(dg32[i] if dg32 is not None else None), B, T, H,
                                              L.stream_ptr()), "evt_gated_act_bwd")
            s = in_slots[i]
            if s.bank.weight_grads:
                HC._bwd_weight(s, x, dx_in, None, B, T, 1.0, L.ACT_NONE, 1.0)
            need_dx = i > 0 or ctx.needs_input_grad[0]
            dx_next = HC._bwd_data(s, dx_in, None, x, dx_res, B, T, 1.0, L.ACT_NONE, 1.0) if need_dx else None
        dg = dg32.to(g_lbh.dtype) if (dg32 is not None and ctx.needs_input_grad[1]) else None
        return dx_next, dg, None, None, None, None, None


def wn_stack(x, g_lbh, lens, in_layers, rs_layers, hidden):
    in_slots = tuple(m._slot for m in in_layers)
    rs_slots = tuple(m._slot for m in rs_layers)
    if any(s is None for s in in_slots + rs_slots):
        raise L.EvtError("WN stack before WeightBank.attach(); there is no eager fallback")
    return WNStackFn.apply(x, g_lbh, lens, in_slots[0].bank.anchor, in_slots, rs_slots, int(hidden))
