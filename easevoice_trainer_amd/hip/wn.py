"""The WaveNet-style gated stack of the posterior encoder and the flow (WN, src/easevoice/module/modules.py:135-212) as ONE
autograd node per stack.

Per layer the arithmetic is the reference's: in_layer conv (k = 5) -> tanh * sigmoid gate with the conditioning slice ->
res_skip conv (1x1) -> x <- (x + rs[:H]) * mask, out <- out + rs[H:] (last layer: out <- (out + rs) * mask).  The launches
are the library's: forward ONE per layer in the 16-bit types at the model's shape (evt_wn_layer_fwd, csrc/wn_layer.hip: H =
192, k = 5; the gate output stays in LDS between the two convolutions), otherwise four (evt_conv1d_fwd, evt_gated_act_fwd,
evt_conv1d_fwd, evt_wn_residual_fwd: fp32, other shapes, EVT_NO_WN_LAYER=1); backward likewise ONE launch for the data half
(evt_wn_layer_bwd_data) plus the two weight gradients, otherwise four + two; what the
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
        fused = FUSED_FORWARD and _layer_fused(dt, H, in_slots, rs_slots)
        frag_in, frag_rs = _frag_images(in_slots, rs_slots) if fused else (None, None)
        for i in range(n_layers):
            last = i == n_layers - 1
            if fused:
                mi, mr = in_slots[i].module, rs_slots[i].module
                x_in = torch.empty((B, T, 2 * H), dtype=x.dtype, device=x.device)
                acts = torch.empty((B, T, H), dtype=x.dtype, device=x.device)
                acc_out = torch.empty((B, T, H), dtype=x.dtype, device=x.device)
                x_out = None if last else torch.empty_like(acc_out)
                e0 = HC._t0()
                L.check(L.lib().evt_wn_layer_fwd(dt, L.ptr(x), L.ptr(frag_in[i]),
                                                 L.ptr(mi.bias.data if mi.bias is not None else None), L.ptr(frag_rs[i]),
                                                 L.ptr(mr.bias.data if mr.bias is not None else None),
                                                 L.ptr(g_lbh[i] if g_lbh is not None else None), L.ptr(out), L.ptr(lens),
                                                 L.ptr(x_in), L.ptr(acts), L.ptr(x_out), L.ptr(acc_out), B, T, H, mi.k,
                                                 int(last), L.stream_ptr()), "evt_wn_layer_fwd")
                if e0 is not None:
                    _t1_layer(e0, mi, mr, x)
                saved += [x, x_in, acts]
                x, out = x_out, acc_out
                continue
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
        fused = FUSED_BACKWARD and _layer_fused(dt, H, in_slots, rs_slots)
        for i in reversed(range(n_layers)):
            last = i == n_layers - 1
            x, x_in, acts = saved[3 * i: 3 * i + 3]
            rs_w = H if last else 2 * H
            if fused:
                # one launch for the data half (csrc/wn_layer.hip); the weight gradients read the drs / dx_in it writes
                si, sr = in_slots[i], rs_slots[i]
                drs = torch.empty((B, T, rs_w), dtype=dtype, device=dev)
                dx_in = torch.empty((B, T, 2 * H), dtype=dtype, device=dev)
                dx = torch.empty((B, T, H), dtype=dtype, device=dev)
                e0 = HC._t0()
                L.check(L.lib().evt_wn_layer_bwd_data(dt, L.ptr(dx_next), L.ptr(dacc), L.ptr(x_in),
                                                      L.ptr(g_lbh[i] if g_lbh is not None else None),
                                                      L.ptr(sr.bank.frag(sr, "alt")), L.ptr(si.bank.frag(si, "alt")),
                                                      L.ptr(lens), L.ptr(drs), L.ptr(dx_in), L.ptr(dx),
                                                      L.ptr(dg32[i] if dg32 is not None else None), B, T, H, si.module.k,
                                                      int(last), L.stream_ptr()), "evt_wn_layer_bwd_data")
                if e0 is not None:
                    _t1_layer(e0, si.module, sr.module, x, "bwd_data")
                if last:
                    dacc = drs
                if sr.bank.weight_grads:
                    HC._bwd_weight(sr, acts, drs, None, B, T, 1.0, L.ACT_NONE, 1.0)
                if si.bank.weight_grads:
                    HC._bwd_weight(si, x, dx_in, None, B, T, 1.0, L.ACT_NONE, 1.0)
                dx_next = dx
                continue
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


FUSED_FORWARD = True     # tests / measurements: False = the four-launch layer forward at every shape
FUSED_BACKWARD = True    # likewise: False = the four-launch data half of the layer backward


def _layer_fused(dt, H, in_slots, rs_slots):
    """the one-launch layer kernels serve this stack: 16-bit type, every in_layer [H -> 2H, k = 5, dilation 1, 'same' padding],
    every res_skip 1 x 1 [H -> 2H, last H -> H] -- the two WN stacks of the model (posterior encoder, flow).  Decided once per
    stack and bank (the geometry of a module does not change)."""
    bank = in_slots[0].bank
    cache = bank.__dict__.setdefault("_wn_fused", {})
    key = (id(in_slots[0]), len(in_slots), dt, H)
    ok = cache.get(key)
    if ok is None:
        ok = cache[key] = _layer_fused_check(dt, H, in_slots, rs_slots)
    return ok


def _layer_fused_check(dt, H, in_slots, rs_slots):
    n = len(in_slots)
    for i, (si, sr) in enumerate(zip(in_slots, rs_slots)):
        mi, mr = si.module, sr.module
        if not L.lib().evt_wn_layer_supported(dt, H, mi.k, mi.dil):
            return False
        if (mi.cin, mi.cout, mi.stride, mi.groups, mi.pad, mi.transposed) != (H, 2 * H, 1, 1, (mi.k - 1) // 2, False):
            return False
        if (mr.cin, mr.cout, mr.k, mr.stride, mr.groups, mr.pad, mr.transposed) != (H, H if i == n - 1 else 2 * H, 1, 1, 1, 0,
                                                                                    False):
            return False
    return True


def _frag_images(in_slots, rs_slots):
    """the stack's REG images in MFMA fragment order (WeightBank.frag: copies the bank re-makes behind every fold)"""
    bank = in_slots[0].bank
    return [bank.frag(s, "reg") for s in in_slots], [bank.frag(s, "reg") for s in rs_slots]


def _t1_layer(e0, mi, mr, x, kind="fwd"):
    """trace record of one fused layer launch (bench.py's per-launch table): flops of both convolutions; bytes: forward = x and
    the skip sum in, x_in / acts / x / skip sum out; backward-data = the two gradients and x_in in, drs / dx_in / dx out; both
    weight images either way"""
    e0, rf = e0
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    if rf is not None:
        rf.__exit__(None, None, None)
    n, ln, h = x.shape
    macs = n * ln * (mi.cin * mi.cout * mi.k + mr.cin * mr.cout)
    HC.TRACE.append((L.lib().evt_last_kernel_tag().decode(), kind, 2 * macs,
                     (7 * x.numel() + mi.v.numel() + mr.v.numel()) * 2, e0, e1,
                     f"WN layer {h}>{mi.cout}>{mr.cout} k{mi.k} n{n} L{ln}", mi))


def wn_stack(x, g_lbh, lens, in_layers, rs_layers, hidden):
    in_slots = tuple(m._slot for m in in_layers)
    rs_slots = tuple(m._slot for m in rs_layers)
    if any(s is None for s in in_slots + rs_slots):
        raise L.EvtError("WN stack before WeightBank.attach(); there is no eager fallback")
    return WNStackFn.apply(x, g_lbh, lens, in_slots[0].bank.anchor, in_slots, rs_slots, int(hidden))
