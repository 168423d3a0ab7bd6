"""Fused element-wise blocks of the s2 transformer encoders (csrc/enc_ops.hip) as autograd functions, plus the
device-side RNG counter their dropout masks are keyed on.

The counter lives in device memory and is bumped by a 1-thread launch (`bump_rng`), never by a host argument, so a
captured HIP graph of the training step draws a new dropout mask on every replay."""
import ctypes as C
import itertools

import torch

from . import lib as L

_RNG = {}
_SITES = itertools.count(1)


def rng_counter(device) -> torch.Tensor:
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _RNG.get(key)
    if t is None:
        t = _RNG[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def seed_rng(device, seed: int):
    rng_counter(device).fill_(int(seed) & 0x7FFFFFFF)


def bump_rng(device):
    """advance the dropout stream by one step (call once per training step, inside the captured region)"""
    t = rng_counter(device)
    L.check(L.lib().evt_counter_inc(L.ptr(t), C.c_uint32(1), L.stream_ptr()), "evt_counter_inc")


def grad_sink(p):
    """fp32 arena gradient view of a runtime-managed parameter (runtime.ModelRuntime): a fused backward kernel adds its
    parameter gradient there and returns None to autograd.  None for a parameter outside a runtime (stand-alone tests):
    the caller then allocates a zeroed tensor and returns it the ordinary way."""
    v = getattr(p, "_evt_grad_view", None)
    if v is not None and v.dtype == torch.float32 and v.is_contiguous() and v.numel() == p.numel() and v.device == p.device:
        return v
    return None


def new_site() -> int:
    """a distinct dropout stream id per call site (module instance)"""
    return next(_SITES)


class ResDropLNFn(torch.autograd.Function):
    """out = LayerNorm(x + dropout(y)) * row_mask   (attentions.py:60-75 of the reference: drop, add, norm, mask)"""

    @staticmethod
    def forward(ctx, x, y, gamma, beta, lens, p, site, eps):
        if x.shape != y.shape or x.dtype != y.dtype or not x.is_contiguous() or not y.is_contiguous():
            raise L.EvtError(f"res_drop_ln: x/y must be contiguous and alike, got {tuple(x.shape)} {x.dtype} / "
                             f"{tuple(y.shape)} {y.dtype}")
        Cc = x.size(-1)
        rows = x.numel() // Cc
        rps = x.size(-2)
        out = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        seed = rng_counter(x.device)
        L.check(L.lib().evt_res_dropout_ln_fwd(L.dt_of(x), L.ptr(x), L.ptr(y), L.ptr(gamma), L.ptr(beta), L.ptr(lens),
                                               rps, C.c_float(p), L.ptr(seed), C.c_uint32(site), L.ptr(out), L.ptr(mean),
                                               L.ptr(rstd), C.c_int64(rows), Cc, C.c_float(eps), L.stream_ptr()),
                "evt_res_dropout_ln_fwd")
        ctx.save_for_backward(x, y, gamma, mean, rstd, lens)
        ctx.cfg = (p, site, rps, rows, Cc)
        ctx.sinks = (grad_sink(gamma), grad_sink(beta))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, gamma, mean, rstd, lens = ctx.saved_tensors
        p, site, rps, rows, Cc = ctx.cfg
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        dy = torch.empty_like(x) if p > 0.0 else None
        sunk = ctx.sinks[0] is not None and ctx.sinks[1] is not None
        if sunk:
            dgamma, dbeta = ctx.sinks
        else:
            dgb = torch.zeros(2, Cc, dtype=torch.float32, device=x.device)     # one fill for both accumulators
            dgamma, dbeta = dgb[0], dgb[1]
        seed = rng_counter(x.device)
        L.check(L.lib().evt_res_dropout_ln_bwd(L.dt_of(x), L.ptr(x), L.ptr(y), L.ptr(gamma), L.ptr(dout), L.ptr(mean),
                                               L.ptr(rstd), L.ptr(lens), rps, C.c_float(p), L.ptr(seed),
                                               C.c_uint32(site), L.ptr(dx), L.ptr(dy), L.ptr(dgamma), L.ptr(dbeta),
                                               C.c_int64(rows), Cc, L.stream_ptr()), "evt_res_dropout_ln_bwd")
        if sunk:
            dgamma = dbeta = None
        return dx, (dy if dy is not None else dx), dgamma, dbeta, None, None, None, None


def res_drop_ln(x, y, gamma, beta, lens, p, site, eps=1e-5):
    return ResDropLNFn.apply(x, y, gamma, beta, lens, float(p), int(site), float(eps))


def _mha_params(dtype, B, H, D, Tq, Tk, window, n_heads_rel, ldq, ldk, ldo, scale, p, site, device):
    """evt_mha_params (include/evt.h): window None = no relative positions"""
    return L.MhaParams(L.dt_code(dtype), B, H, D, Tq, Tk,
                       -1 if window is None else int(window), int(n_heads_rel), ldq, ldk, ldo, float(scale), float(p),
                       int(site), 0, rng_counter(device).data_ptr())


class MhaCoreFn(torch.autograd.Function):
    """Multi-head attention core (csrc/mha.hip) on projected rows: q [B, Tq, H*D], k / v [B, Tk, H*D] in bf16 or fp32 ->
    [B, Tq, H*D].  window None: plain attention (MRTE cross-attention, style self-attention); window w: the
    relative-position self-attention of enc_p with its two [Hr, 2w+1, D] embeddings.  One launch forward, two backward."""

    @staticmethod
    def forward(ctx, q, k, v, emb_k, emb_v, lens_q, lens_k, n_heads, window, p, site, scale):
        B, Tq, Cc = q.shape
        Tk = k.size(1)
        if q.dtype not in (torch.bfloat16, torch.float16, torch.float32) or k.dtype != q.dtype or v.dtype != q.dtype:
            raise L.EvtError(f"mha: q / k / v must share bf16 or fp32, got {q.dtype} {k.dtype} {v.dtype}")
        if Cc % n_heads or k.shape != (B, Tk, Cc) or v.shape != k.shape:
            raise L.EvtError(f"mha: shapes q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)} heads {n_heads}")
        if not (q.is_contiguous() and k.is_contiguous() and v.is_contiguous()):
            raise L.EvtError("mha: contiguous q / k / v expected (a packed projection goes through rel_attention)")
        D = Cc // n_heads
        ek = emb_k.contiguous() if window is not None else None
        ev = emb_v.contiguous() if window is not None else None
        out = torch.empty((B, Tq, Cc), dtype=q.dtype, device=q.device)
        lse = torch.empty((B * n_heads, Tq), dtype=torch.float32, device=q.device)
        hr = ek.size(0) if ek is not None else 1
        prm = _mha_params(q.dtype, B, n_heads, D, Tq, Tk, window, hr, Cc, Cc, Cc, scale, p, site, q.device)
        L.check(L.lib().evt_mha_fwd(C.byref(prm), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(ek), L.ptr(ev), L.ptr(lens_q), L.ptr(lens_k),
                                    L.ptr(out), L.ptr(lse), L.stream_ptr()), "evt_mha_fwd")
        ctx.save_for_backward(q, k, v, out, lse, ek, ev, lens_q, lens_k)
        ctx.cfg = (n_heads, window, p, site, scale)
        ctx.sinks = (grad_sink(emb_k), grad_sink(emb_v)) if window is not None else (None, None)
        return out

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, out, lse, ek, ev, lens_q, lens_k = ctx.saved_tensors
        n_heads, window, p, site, scale = ctx.cfg
        B, Tq, Cc = q.shape
        Tk = k.size(1)
        d_o = d_o.contiguous()
        dq = torch.empty((B, Tq, Cc), dtype=q.dtype, device=q.device)
        dkv = torch.empty((2, B, Tk, Cc), dtype=q.dtype, device=q.device)
        sunk = ctx.sinks[0] is not None and ctx.sinks[1] is not None
        demb = None
        if window is not None:
            demb = ctx.sinks if sunk else torch.zeros((2,) + tuple(ek.shape), dtype=torch.float32, device=q.device)
        delta = torch.empty_like(lse)
        hr = ek.size(0) if ek is not None else 1
        prm = _mha_params(q.dtype, B, n_heads, Cc // n_heads, Tq, Tk, window, hr, Cc, Cc, Cc, scale, p, site, q.device)
        L.check(L.lib().evt_mha_bwd(C.byref(prm), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(d_o), L.ptr(lse), L.ptr(ek), L.ptr(ev),
                                    L.ptr(lens_q), L.ptr(lens_k), L.ptr(dq), L.ptr(dkv[0]), L.ptr(dkv[1]),
                                    L.ptr(demb[0]) if demb is not None else None,
                                    L.ptr(demb[1]) if demb is not None else None, L.ptr(delta), L.stream_ptr()),
                "evt_mha_bwd")
        dek = dev = None
        if window is not None and not sunk:
            dek, dev = demb[0], demb[1]
        return dq, dkv[0], dkv[1], dek, dev, None, None, None, None, None, None, None


def mha_core(q, k, v, lens_q, lens_k, n_heads, p, site, scale, emb_k=None, emb_v=None, window=None):
    return MhaCoreFn.apply(q, k, v, emb_k, emb_v, lens_q, lens_k, int(n_heads), window, float(p), int(site), float(scale))


class RelAttnFn(torch.autograd.Function):
    """The attention core on a packed projection qkv [B, T, 3*H*D] (bf16 or fp32; rows [q | k | v], the layout the packed
    [3C, C] projection GEMM writes): gradients come back as one packed [B, T, 3*H*D] tensor."""

    @staticmethod
    def forward(ctx, qkv, emb_k, emb_v, lens, n_heads, window, p, site, scale):
        B, T, C3 = qkv.shape
        Cc = C3 // 3
        if qkv.dtype not in (torch.bfloat16, torch.float16, torch.float32) or not qkv.is_contiguous() or C3 % 3 or Cc % n_heads:
            raise L.EvtError(f"mha: packed contiguous [B, T, 3*H*D] expected, got {tuple(qkv.shape)} {qkv.dtype}")
        D = Cc // n_heads
        ek = emb_k.contiguous() if window is not None else None
        ev = emb_v.contiguous() if window is not None else None
        out = torch.empty((B, T, Cc), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B * n_heads, T), dtype=torch.float32, device=qkv.device)
        hr = ek.size(0) if ek is not None else 1
        scale = D ** -0.5 if scale is None else scale
        prm = _mha_params(qkv.dtype, B, n_heads, D, T, T, window, hr, C3, C3, Cc, scale, p, site, qkv.device)
        base, esz = qkv.data_ptr(), qkv.element_size()
        L.check(L.lib().evt_mha_fwd(C.byref(prm), C.c_void_p(base), C.c_void_p(base + Cc * esz),
                                    C.c_void_p(base + 2 * Cc * esz), L.ptr(ek), L.ptr(ev), L.ptr(lens), L.ptr(lens),
                                    L.ptr(out), L.ptr(lse), L.stream_ptr()), "evt_mha_fwd")
        ctx.save_for_backward(qkv, out, lse, ek, ev, lens)
        ctx.cfg = (n_heads, window, p, site, scale)
        return out

    @staticmethod
    def backward(ctx, d_o):
        qkv, out, lse, ek, ev, lens = ctx.saved_tensors
        n_heads, window, p, site, scale = ctx.cfg
        B, T, C3 = qkv.shape
        Cc = C3 // 3
        d_o = d_o.contiguous()
        dqkv = torch.empty_like(qkv)
        demb = None
        if window is not None:
            demb = torch.zeros((2,) + tuple(ek.shape), dtype=torch.float32, device=qkv.device)   # one fill for both
        delta = torch.empty_like(lse)
        hr = ek.size(0) if ek is not None else 1
        prm = _mha_params(qkv.dtype, B, n_heads, Cc // n_heads, T, T, window, hr, C3, C3, Cc, scale, p, site, qkv.device)
        base, dbase, esz = qkv.data_ptr(), dqkv.data_ptr(), qkv.element_size()
        L.check(L.lib().evt_mha_bwd(C.byref(prm), C.c_void_p(base), C.c_void_p(base + Cc * esz),
                                    C.c_void_p(base + 2 * Cc * esz), L.ptr(out), L.ptr(d_o), L.ptr(lse), L.ptr(ek),
                                    L.ptr(ev), L.ptr(lens), L.ptr(lens), C.c_void_p(dbase),
                                    C.c_void_p(dbase + Cc * esz), C.c_void_p(dbase + 2 * Cc * esz),
                                    L.ptr(demb[0]) if demb is not None else None,
                                    L.ptr(demb[1]) if demb is not None else None, L.ptr(delta), L.stream_ptr()),
                "evt_mha_bwd")
        return (dqkv, demb[0] if demb is not None else None, demb[1] if demb is not None else None, None, None, None,
                None, None, None)


def rel_attention(qkv, emb_k, emb_v, lens, n_heads, window, p, site, scale=None):
    return RelAttnFn.apply(qkv, emb_k, emb_v, lens, int(n_heads), window, float(p), int(site), scale)


class RelSelfAttnFn(torch.autograd.Function):
    """The q / k / v projections (three 1x1 convs of the fused conv family, attentions.py:196-205 of the reference; the
    style encoder's three nn.Linear, modules.py:611-613) AND the fused attention core as ONE autograd node:
    x [B, T, C] (bf16 or fp32) -> [B, T, C].
    Forward: three k = 1 conv launches write the planes of one [3, B, T, C] buffer (or ONE packed [3C, C] GEMM writes
    rows [q | k | v]), evt_mha_fwd reads them through its three row pointers.  Backward: evt_mha_bwd, the
    weight-gradient launches (each also produces the bias gradient), the backward-data launches chained through their
    add-epilogue -- so the sum over the three branches needs no element-wise launch.  (Through F.linear this was: two
    weight concatenations, a cast of the packed weight, a vendor GEMM, two more GEMMs, a column-sum launch and the
    concatenation's backward per layer and step.)"""

    @staticmethod
    def forward(ctx, x, anchor, emb_k, emb_v, lens, slots, n_heads, window, p, site, scale):
        from . import conv as HC

        B, T, _ = x.shape
        Cc = slots[0].module.cout // 3 if len(slots) == 1 else slots[0].module.cout
        if x.dtype != slots[0].bank.dtype or not x.is_contiguous() or Cc % n_heads:
            raise L.EvtError(f"self-attention: contiguous {slots[0].bank.dtype} [B, T, C] expected, got "
                             f"{tuple(x.shape)} {x.dtype}")
        D = Cc // n_heads
        ek = emb_k.contiguous() if window is not None else None
        ev = emb_v.contiguous() if window is not None else None
        packed = len(slots) == 1
        if packed:       # one [3C, C] GEMM: rows [q | k | v], row stride 3C
            qkv = torch.empty((B, T, 3 * Cc), dtype=x.dtype, device=x.device)
            HC._fwd(slots[0], x, None, 1.0, L.ACT_NONE, 1.0, out=qkv)
            ld, esz, base = 3 * Cc, x.element_size(), qkv.data_ptr()
            qp, kp, vp = C.c_void_p(base), C.c_void_p(base + Cc * esz), C.c_void_p(base + 2 * Cc * esz)
        else:            # three launches into the planes of one buffer, row stride C
            qkv = torch.empty((3, B, T, Cc), dtype=x.dtype, device=x.device)
            for i, s in enumerate(slots):
                HC._fwd(s, x, None, 1.0, L.ACT_NONE, 1.0, out=qkv[i])
            ld, (qp, kp, vp) = Cc, (L.ptr(qkv[0]), L.ptr(qkv[1]), L.ptr(qkv[2]))
        out = torch.empty((B, T, Cc), dtype=x.dtype, device=x.device)
        lse = torch.empty((B * n_heads, T), dtype=torch.float32, device=x.device)
        hr = ek.size(0) if ek is not None else 1
        prm = _mha_params(x.dtype, B, n_heads, D, T, T, window, hr, ld, ld, Cc, scale, p, site, x.device)
        L.check(L.lib().evt_mha_fwd(C.byref(prm), qp, kp, vp, L.ptr(ek), L.ptr(ev), L.ptr(lens), L.ptr(lens),
                                    L.ptr(out), L.ptr(lse), L.stream_ptr()), "evt_mha_fwd")
        ctx.save_for_backward(x, qkv, out, lse, ek, ev, lens)
        ctx.cfg = (slots, n_heads, window, p, site, scale, Cc)
        ctx.sinks = (grad_sink(emb_k), grad_sink(emb_v)) if window is not None else (None, None)
        return out

    @staticmethod
    def backward(ctx, d_o):
        from . import conv as HC

        x, qkv, out, lse, ek, ev, lens = ctx.saved_tensors
        slots, n_heads, window, p, site, scale, Cc = ctx.cfg
        B, T, _ = x.shape
        packed = len(slots) == 1
        d_o = d_o.contiguous()
        dqkv = torch.empty_like(qkv)
        sunk = ctx.sinks[0] is not None and ctx.sinks[1] is not None
        demb = None
        if window is not None:
            demb = ctx.sinks if sunk else torch.zeros((2,) + tuple(ek.shape), dtype=torch.float32, device=x.device)
        delta = torch.empty_like(lse)
        if packed:
            ld, esz = 3 * Cc, x.element_size()
            ptrs = [C.c_void_p(t.data_ptr() + i * Cc * esz) for t in (qkv, dqkv) for i in range(3)]
        else:
            ld, ptrs = Cc, [L.ptr(t[i]) for t in (qkv, dqkv) for i in range(3)]
        hr = ek.size(0) if ek is not None else 1
        prm = _mha_params(x.dtype, B, n_heads, Cc // n_heads, T, T, window, hr, ld, ld, Cc, scale, p, site, x.device)
        L.check(L.lib().evt_mha_bwd(C.byref(prm), ptrs[0], ptrs[1], ptrs[2], L.ptr(out), L.ptr(d_o), L.ptr(lse),
                                    L.ptr(ek), L.ptr(ev), L.ptr(lens), L.ptr(lens), ptrs[3], ptrs[4], ptrs[5],
                                    L.ptr(demb[0]) if demb is not None else None,
                                    L.ptr(demb[1]) if demb is not None else None, L.ptr(delta), L.stream_ptr()),
                "evt_mha_bwd")
        dx = None
        for i, s in enumerate(slots):
            dy = dqkv if packed else dqkv[i]
            if s.bank.weight_grads:
                HC._bwd_weight(s, x, dy, None, B, T, 1.0, L.ACT_NONE, 1.0)
            if ctx.needs_input_grad[0]:
                dx = HC._bwd_data(s, dy, None, x, dx, B, T, 1.0, L.ACT_NONE, 1.0)
        if sunk or window is None:
            return dx, None, None, None, None, None, None, None, None, None, None
        return dx, None, demb[0], demb[1], None, None, None, None, None, None, None


def rel_self_attention(x, conv_q, conv_k, conv_v, emb_k, emb_v, lens, n_heads, window, p, site, packed=None, scale=None):
    """packed: hip/conv.py::PackedConv of the three projections (one launch each way) or None (three); window None: no
    relative positions (emb_k / emb_v ignored); scale None: 1/sqrt(head width)"""
    slots = (packed._slot,) if packed is not None and packed._slot is not None else (conv_q._slot, conv_k._slot, conv_v._slot)
    if any(s is None for s in slots):
        raise L.EvtError("rel_self_attention before WeightBank.attach(); there is no eager fallback")
    if scale is None:
        scale = (conv_q.cout // n_heads) ** -0.5
    return RelSelfAttnFn.apply(x, slots[0].bank.anchor, emb_k, emb_v, lens, slots, int(n_heads), window, float(p),
                               int(site), float(scale))


class WNResidualFn(torch.autograd.Function):
    """x_new = (x + rs[..., :H]) * mask, acc_new = acc + rs[..., H:]   (last layer: acc_new = (acc + rs) * mask)"""

    @staticmethod
    def forward(ctx, x, rs, acc, lens, last):
        H = rs.size(-1) if last else rs.size(-1) // 2
        rows = rs.numel() // rs.size(-1)
        rps = rs.size(-2)
        for t in (x, rs, acc):
            if t is not None and not t.is_contiguous():
                raise L.EvtError("wn_residual: contiguous tensors expected")
        acc_out = torch.empty(rs.shape[:-1] + (H,), dtype=rs.dtype, device=rs.device)
        x_out = None if last else torch.empty_like(acc_out)
        L.check(L.lib().evt_wn_residual_fwd(L.dt_of(rs), L.ptr(x), L.ptr(rs), L.ptr(acc), L.ptr(lens), rps, L.ptr(x_out),
                                            L.ptr(acc_out), C.c_int64(rows), H, int(last), L.stream_ptr()),
                "evt_wn_residual_fwd")
        ctx.save_for_backward(lens)
        ctx.cfg = (H, rows, rps, last, rs.shape, rs.dtype, acc is not None)
        if last:
            return acc_out
        return x_out, acc_out

    @staticmethod
    def backward(ctx, *grads):
        (lens,) = ctx.saved_tensors
        H, rows, rps, last, rs_shape, dtype, has_acc = ctx.cfg
        if last:
            dx_out, dacc = None, grads[0]
        else:
            dx_out, dacc = grads
        dx_out = dx_out.contiguous() if dx_out is not None else None
        dacc = dacc.contiguous() if dacc is not None else None
        dev = (dacc if dacc is not None else dx_out).device
        drs = torch.empty(rs_shape, dtype=dtype, device=dev)
        dx = None if last else torch.empty(rs_shape[:-1] + (H,), dtype=dtype, device=dev)
        L.check(L.lib().evt_wn_residual_bwd(L.dt_code(dtype), L.ptr(dx_out), L.ptr(dacc),
                                            L.ptr(lens), rps, L.ptr(dx), L.ptr(drs), C.c_int64(rows), H, int(last),
                                            L.stream_ptr()), "evt_wn_residual_bwd")
        # d(acc): the skip sum passes through unchanged, except for the last layer where the row mask applies to it too
        return dx, drs, ((drs if last else dacc) if has_acc else None), None, None


def wn_residual(x, rs, acc, lens):
    return WNResidualFn.apply(x, rs, acc, lens, False)


def wn_residual_last(rs, acc, lens):
    return WNResidualFn.apply(None, rs, acc, lens, True)


class MishDropoutFn(torch.autograd.Function):
    """y = dropout(x * tanh(softplus(x))) in one pass (modules.py:521-545 Mish + nn.Dropout of the style encoder); the
    backward regenerates the mask from the device counter and applies mish'(x).  out_dtype: y's dtype (fp32 next to a
    bf16 x where the reference's autocast leaves the activation in fp32)."""

    @staticmethod
    def forward(ctx, x, p, site, out_dtype):
        x = x.contiguous()
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        L.check(L.lib().evt_mish_dropout_fwd(L.dt_of(x), L.dt_of(y), L.ptr(x), C.c_float(p), L.ptr(rng_counter(x.device)),
                                             C.c_uint32(site), L.ptr(y), C.c_int64(x.numel()), L.stream_ptr()),
                "evt_mish_dropout_fwd")
        ctx.save_for_backward(x)
        ctx.cfg = (p, site, out_dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        p, site, out_dtype = ctx.cfg
        dy = dy.to(out_dtype).contiguous()
        dx = torch.empty_like(x)
        L.check(L.lib().evt_mish_dropout_bwd(L.dt_of(x), L.dt_of(dy), L.ptr(x), L.ptr(dy), C.c_float(p),
                                             L.ptr(rng_counter(x.device)), C.c_uint32(site), L.ptr(dx), C.c_int64(x.numel()),
                                             L.stream_ptr()), "evt_mish_dropout_bwd")
        return dx, None, None, None


def mish_dropout(x, p, site, out_dtype=None):
    return MishDropoutFn.apply(x, float(p), int(site), out_dtype or x.dtype)


class GluDropoutResFn(torch.autograd.Function):
    """y = res + dropout(h[..., :C] * sigmoid(h[..., C:])) in one pass (Conv1dGLU, modules.py:548-566); h [..., 2C] in the
    compute dtype, res / y in res's dtype (fp32 or the compute dtype)"""

    @staticmethod
    def forward(ctx, h, res, p, site):
        h = h.contiguous()
        if res.dtype not in (h.dtype, torch.float32):
            res = res.to(h.dtype)
        res = res.contiguous()
        Cc = h.size(-1) // 2
        y = torch.empty_like(res)
        L.check(L.lib().evt_glu_dropout_res_fwd(L.dt_of(h), L.dt_of(res), L.ptr(h), L.ptr(res), C.c_float(p),
                                                L.ptr(rng_counter(h.device)), C.c_uint32(site), L.ptr(y),
                                                C.c_int64(h.numel() // (2 * Cc)), Cc, L.stream_ptr()),
                "evt_glu_dropout_res_fwd")
        ctx.save_for_backward(h)
        ctx.cfg = (p, site, Cc, res.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, = ctx.saved_tensors
        p, site, Cc, rdt = ctx.cfg
        dy = dy.to(rdt).contiguous()
        dh = torch.empty_like(h)
        L.check(L.lib().evt_glu_dropout_res_bwd(L.dt_of(h), L.dt_of(dy), L.ptr(h), L.ptr(dy), C.c_float(p),
                                                L.ptr(rng_counter(h.device)), C.c_uint32(site), L.ptr(dh),
                                                C.c_int64(h.numel() // (2 * Cc)), Cc, L.stream_ptr()),
                "evt_glu_dropout_res_bwd")
        return dh, dy, None, None


def glu_dropout_res(h, res, p, site):
    return GluDropoutResFn.apply(h, res, float(p), int(site))


class ReparamFn(torch.autograd.Function):
    """posterior encoder tail (models.py:352-358): stats [B, T, 2C] (the projection's output, compute dtype), eps [B, T, C]
    fp32, lens -> z, m, logs fp32 (masked) in one launch; one launch back"""

    @staticmethod
    def forward(ctx, stats, eps, lens):
        stats = stats.contiguous()
        eps = eps.float().contiguous()
        B, T, C2 = stats.shape
        Cc = C2 // 2
        z = torch.empty((B, T, Cc), dtype=torch.float32, device=stats.device)
        m, logs = torch.empty_like(z), torch.empty_like(z)
        L.check(L.lib().evt_reparam_fwd(L.dt_of(stats), L.ptr(stats), L.ptr(eps), L.ptr(lens), T, C.c_int64(B * T), Cc,
                                        L.ptr(z), L.ptr(m), L.ptr(logs), L.stream_ptr()), "evt_reparam_fwd")
        ctx.save_for_backward(eps, logs, lens)
        ctx.sdt, ctx.dims = stats.dtype, (B, T, Cc)
        return z, m, logs

    @staticmethod
    def backward(ctx, dz, dm, dlogs):
        eps, logs, lens = ctx.saved_tensors
        B, T, Cc = ctx.dims
        gs = [g.float().contiguous() if g is not None else None for g in (dz, dm, dlogs)]
        dstats = torch.empty((B, T, 2 * Cc), dtype=ctx.sdt, device=eps.device)
        dt = L.dt_code(ctx.sdt)
        L.check(L.lib().evt_reparam_bwd(dt, L.ptr(gs[0]), L.ptr(gs[1]), L.ptr(gs[2]), L.ptr(eps), L.ptr(logs), L.ptr(lens), T,
                                        C.c_int64(B * T), Cc, L.ptr(dstats), L.stream_ptr()), "evt_reparam_bwd")
        return dstats, None, None


def reparam(stats, eps, lens):
    return ReparamFn.apply(stats, eps, lens)


class CouplingFlipFn(torch.autograd.Function):
    """mean-only residual coupling + Flip after the layer's `post` projection (modules.py:404-458 with logs == 0, followed by
    the Flip of models.py:273-315):  y = flip([x0, (x1 + stats) * mask]) in fp32 and x0n = y[..., :h] in the compute dtype
    -- the next layer's projection input -- in ONE launch (evt_coupling_flip_fwd); the backward takes the gradients of both
    outputs in one launch.  x [B, T, 2h] fp32, stats [B, T, h] compute dtype (unmasked), lens [B] int32."""

    @staticmethod
    def forward(ctx, x, stats, lens, want_x0n):
        B, T, C2 = x.shape
        h = C2 // 2
        x = x.contiguous()
        stats = stats.contiguous()
        y = torch.empty_like(x)
        x0n = torch.empty((B, T, h), dtype=stats.dtype, device=x.device) if want_x0n else None
        L.check(L.lib().evt_coupling_flip_fwd(L.dt_of(stats), L.ptr(x), L.ptr(stats), L.ptr(lens), T, C.c_int64(B * T), h,
                                              L.ptr(y), L.ptr(x0n), L.stream_ptr()), "evt_coupling_flip_fwd")
        ctx.save_for_backward(lens)
        ctx.dims, ctx.sdt = (B, T, h), stats.dtype
        if want_x0n:
            return y, x0n
        return y, None

    @staticmethod
    def backward(ctx, dy, dx0n):
        lens, = ctx.saved_tensors
        B, T, h = ctx.dims
        if dy is None:
            dy = torch.zeros((B, T, 2 * h), dtype=torch.float32, device=lens.device)
        dy = dy.contiguous()
        if dx0n is not None:
            dx0n = dx0n.to(ctx.sdt).contiguous()
        dx = torch.empty_like(dy)
        dstats = torch.empty((B, T, h), dtype=ctx.sdt, device=dy.device)
        dt = L.dt_code(ctx.sdt)
        L.check(L.lib().evt_coupling_flip_bwd(dt, L.ptr(dy), L.ptr(dx0n), L.ptr(lens), T, C.c_int64(B * T), h, L.ptr(dx),
                                              L.ptr(dstats), L.stream_ptr()), "evt_coupling_flip_bwd")
        return dx, dstats, None, None


def coupling_flip(x, stats, lens, want_x0n=True):
    return CouplingFlipFn.apply(x, stats, lens, want_x0n)


class UnbindRowsFn(torch.autograd.Function):
    """t [L, ...] -> L views; the backward stacks the L gradients with ONE concatenation (torch's unbind backward
    zero-fills and copies per slice)."""

    @staticmethod
    def forward(ctx, t):
        ctx.n = t.size(0)
        return tuple(t.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        return torch.stack([g.contiguous() for g in grads], dim=0)


def unbind_rows(t):
    return UnbindRowsFn.apply(t)


class ReluDropoutFn(torch.autograd.Function):
    """y = dropout(relu(x)) * row_mask in one pass; the backward regenerates the mask and applies the relu gate in one
    pass.  lens [B] int32 (or None): rows t >= lens[b] of x [B, T, C] are zeroed."""

    @staticmethod
    def forward(ctx, x, p, site, lens):
        x = x.contiguous()
        y = torch.empty_like(x)
        rps, Cc = (x.size(-2), x.size(-1)) if lens is not None else (0, 0)
        L.check(L.lib().evt_relu_dropout_fwd(L.dt_of(x), L.ptr(x), C.c_float(p), L.ptr(rng_counter(x.device)),
                                             C.c_uint32(site), L.ptr(lens), rps, Cc, L.ptr(y), C.c_int64(x.numel()),
                                             L.stream_ptr()), "evt_relu_dropout_fwd")
        ctx.save_for_backward(x, lens)
        ctx.cfg = (p, site, rps, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, lens = ctx.saved_tensors
        p, site, rps, Cc = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        L.check(L.lib().evt_relu_dropout_bwd(L.dt_of(x), L.ptr(x), L.ptr(dy), C.c_float(p), L.ptr(rng_counter(x.device)),
                                             C.c_uint32(site), L.ptr(lens), rps, Cc, L.ptr(dx), C.c_int64(x.numel()),
                                             L.stream_ptr()), "evt_relu_dropout_bwd")
        return dx, None, None, None


def relu_dropout(x, p, site, lens=None):
    return ReluDropoutFn.apply(x, float(p), int(site), lens)
