"""GAN / feature / KL losses of the s2 step (src/easevoice/module/losses.py:7-61) as fused HIP
reductions: one launch over a table of all (real, fake) feature-map pairs instead of 37 mean-abs
kernels, one launch for the 6 LSGAN terms, and no `.item()` host syncs inside the step (the
reference does 12 per step at losses.py:28-29)."""
import ctypes as C

import torch

from ..hip import lib as L


def _table(pairs, scales, grads, device):
    segs = []
    for (a, b), sc, da in zip(pairs, scales, grads):
        segs.append(L.Seg(a.data_ptr(), b.data_ptr() if b is not None else None,
                          da.data_ptr() if da is not None else None, a.numel(), sc, 0))
    return L.struct_to_device(segs, device)


class _SegLossFn(torch.autograd.Function):
    """sum_i scale_i * reduce_i(a_i, b_i); mode 0 = sum|a-b| (b detached), mode 1 = sum (target-a)^2."""

    @staticmethod
    def forward(ctx, mode, target, scales, n_a, *tensors):
        a_list, b_list = tensors[:n_a], tensors[n_a:]
        a_list = [t.contiguous() for t in a_list]
        b_list = [t.contiguous() for t in b_list] if mode == 0 else [None] * n_a
        dev = a_list[0].device
        dt = L.dt_of(a_list[0])
        out = torch.zeros(1, dtype=torch.float32, device=dev)
        tab = _table(list(zip(a_list, b_list)), scales, [None] * n_a, dev)
        fn = L.lib().evt_l1_multi_fwd if mode == 0 else L.lib().evt_lsgan_multi_fwd
        if mode == 0:
            L.check(fn(dt, L.ptr(tab), n_a, L.ptr(out), L.stream_ptr()), "evt_l1_multi_fwd")
        else:
            L.check(fn(dt, L.ptr(tab), n_a, C.c_float(target), L.ptr(out), L.stream_ptr()), "evt_lsgan_multi_fwd")
        ctx.mode, ctx.target, ctx.scales, ctx.n_a, ctx.dt = mode, target, scales, n_a, dt
        ctx.save_for_backward(*a_list, *[b for b in b_list if b is not None])
        return out[0]

    @staticmethod
    def backward(ctx, dloss):
        saved = ctx.saved_tensors
        n_a = ctx.n_a
        a_list = saved[:n_a]
        b_list = saved[n_a:] if ctx.mode == 0 else [None] * n_a
        dev = a_list[0].device
        grads = [torch.empty_like(a) if ctx.needs_input_grad[4 + i] else None for i, a in enumerate(a_list)]
        tab = _table(list(zip(a_list, b_list)), ctx.scales, grads, dev)
        dl = dloss.reshape(1).float().contiguous()
        if ctx.mode == 0:
            L.check(L.lib().evt_l1_multi_bwd(ctx.dt, L.ptr(tab), n_a, L.ptr(dl), L.stream_ptr()), "evt_l1_multi_bwd")
        else:
            L.check(L.lib().evt_lsgan_multi_bwd(ctx.dt, L.ptr(tab), n_a, C.c_float(ctx.target), L.ptr(dl),
                                                L.stream_ptr()), "evt_lsgan_multi_bwd")
        return (None, None, None, None, *grads, *([None] * (len(saved) - n_a)))


def feature_loss(fmap_r, fmap_g):
    """losses.py:7-15: 2 * sum over all feature maps of mean|r - g| (r detached)."""
    a, b, sc = [], [], []
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            a.append(gl)
            b.append(rl.detach())
            sc.append(2.0 / gl.numel())
    return _SegLossFn.apply(0, 0.0, sc, len(a), *a, *b)


def l1_mean_scaled(a, b, scale):
    """scale * mean|a - b| with b as the constant side (the mel reconstruction term, sovits.py:513: F.l1_loss * c_mel):
    one reduction launch forward, one backward, where F.l1_loss runs sub / abs / mean / mul and their four gradients"""
    return _SegLossFn.apply(0, 0.0, [float(scale) / a.numel()], 1, a, b.detach())


def discriminator_loss(disc_real_outputs, disc_generated_outputs):
    """losses.py:18-32: sum_i mean((1-dr_i)^2) + mean(dg_i^2).  Returns the scalar only (the per-term
    python lists of the reference exist to be `.item()`-ed for logging, which this path avoids)."""
    r = _SegLossFn.apply(1, 1.0, [1.0 / t.numel() for t in disc_real_outputs], len(disc_real_outputs),
                         *disc_real_outputs)
    g = _SegLossFn.apply(1, 0.0, [1.0 / t.numel() for t in disc_generated_outputs], len(disc_generated_outputs),
                         *disc_generated_outputs)
    return r + g


class _DLossBatchedFn(torch.autograd.Function):
    """discriminator_loss over logits that hold BOTH halves, o_i = [real ; generated] rows (how the D step runs every
    sub-discriminator): the same two reductions, their gradients written into the two halves of ONE tensor per o_i --
    no zero-padded slice gradients and no sums of them (3 launches per sub-discriminator through o[:n] / o[n:])."""

    @staticmethod
    def forward(ctx, *outs):
        outs = [o.contiguous() for o in outs]
        dev, dt = outs[0].device, L.dt_of(outs[0])
        halves = [(o[: o.size(0) // 2], o[o.size(0) // 2:]) for o in outs]
        out = torch.zeros(1, dtype=torch.float32, device=dev)
        for k, target in ((0, 1.0), (1, 0.0)):
            segs = [h[k] for h in halves]
            tab = _table([(t, None) for t in segs], [1.0 / t.numel() for t in segs], [None] * len(segs), dev)
            L.check(L.lib().evt_lsgan_multi_fwd(dt, L.ptr(tab), len(segs), C.c_float(target), L.ptr(out), L.stream_ptr()),
                    "evt_lsgan_multi_fwd")
        ctx.dt = dt
        ctx.save_for_backward(*outs)
        return out[0]

    @staticmethod
    def backward(ctx, dloss):
        outs = ctx.saved_tensors
        dev = outs[0].device
        grads = [torch.empty_like(o) for o in outs]
        dl = dloss.reshape(1).float().contiguous()
        for k, target in ((0, 1.0), (1, 0.0)):
            segs = [(o[: o.size(0) // 2], g[: o.size(0) // 2]) if k == 0 else (o[o.size(0) // 2:], g[o.size(0) // 2:])
                    for o, g in zip(outs, grads)]
            tab = _table([(a, None) for a, _ in segs], [1.0 / a.numel() for a, _ in segs], [g for _, g in segs], dev)
            L.check(L.lib().evt_lsgan_multi_bwd(ctx.dt, L.ptr(tab), len(segs), C.c_float(target), L.ptr(dl),
                                                L.stream_ptr()), "evt_lsgan_multi_bwd")
        return tuple(grads)


def discriminator_loss_batched(outs):
    """discriminator_loss(real halves, generated halves) of logits o_i [2n, ...] = [real ; generated]"""
    return _DLossBatchedFn.apply(*outs)


def generator_loss(disc_outputs):
    """losses.py:35-43: sum_i mean((1-dg_i)^2)."""
    return _SegLossFn.apply(1, 1.0, [1.0 / t.numel() for t in disc_outputs], len(disc_outputs), *disc_outputs)


class _MaskedKLFn(torch.autograd.Function):
    """sum(kl * z_mask) / sum(z_mask), one streaming HIP pass each way (csrc/losses.hip)."""

    @staticmethod
    def forward(ctx, z_p, logs_q, m_p, logs_p, lens, time_inner):
        ts = (z_p, logs_q, m_p, logs_p)
        if any(t.shape != z_p.shape or not t.is_contiguous() for t in ts):
            raise L.EvtError("masked_kl: four contiguous tensors of one shape expected")
        B, d1, d2 = z_p.shape
        T, Cc = (d2, d1) if time_inner else (d1, d2)
        prm = L.KlParams(B, T, Cc, int(time_inner), *[L.dt_of(t) for t in ts])
        out2 = torch.zeros(2, dtype=torch.float32, device=z_p.device)
        L.check(L.lib().evt_masked_kl_fwd(C.byref(prm), *[L.ptr(t) for t in ts], L.ptr(lens), L.ptr(out2),
                                          L.stream_ptr()), "evt_masked_kl_fwd")
        ctx.save_for_backward(z_p, logs_q, m_p, logs_p, lens, out2)
        ctx.prm = prm
        return out2[0] / out2[1]

    @staticmethod
    def backward(ctx, dloss):
        z_p, logs_q, m_p, logs_p, lens, out2 = ctx.saved_tensors
        ts = (z_p, logs_q, m_p, logs_p)
        grads = [torch.empty_like(t) if ctx.needs_input_grad[i] else None for i, t in enumerate(ts)]
        dl = dloss.reshape(1).float().contiguous()
        L.check(L.lib().evt_masked_kl_bwd(C.byref(ctx.prm), *[L.ptr(t) for t in ts], L.ptr(lens), L.ptr(dl),
                                          L.ptr(out2[1:]), *[L.ptr(g) for g in grads], L.stream_ptr()),
                "evt_masked_kl_bwd")
        return (*grads, None, None)


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask, lens=None):
    """losses.py:46-61.  The four tensors are [B, C, T] like the reference's (z_mask [B, 1, T]); their storage is either
    channels-last (what SynthesizerTrn returns: [B, C, T] views of contiguous [B, T, C], read in place) or the
    reference's own contiguous [B, C, T] (or any other strides: copied).  The logical layout is fixed -- [B, C, T] with
    a [B, 1, T] mask, anything else raises -- and the storage layout is read off the strides, never guessed from sizes
    (a batch with T == C is ambiguous by size).  `lens` [B] (the sequence mask as lengths; recovered as the per-row mask
    sums otherwise)."""
    ts = [z_p, logs_q, m_p, logs_p]
    if z_mask.dim() != 3 or z_mask.size(1) != 1 or any(t.dim() != 3 or t.shape != z_p.shape for t in ts) \
            or z_mask.size(-1) != z_p.size(-1) or z_mask.size(0) != z_p.size(0):
        raise L.EvtError(f"kl_loss: [B, C, T] tensors and a [B, 1, T] mask expected, got {[tuple(t.shape) for t in ts]} "
                         f"/ {tuple(z_mask.shape)}")
    if lens is None:
        lens = z_mask.reshape(z_mask.size(0), -1).sum(1)
    lens = lens.to(torch.int32).contiguous()
    B, Cc, T = z_p.shape
    # [B, C, T] views of contiguous [B, T, C] storage are read in place; anything else (the reference's own contiguous
    # [B, C, T], slices of a wider statistics tensor, a mix) is copied into contiguous [B, C, T] first -- the LOGICAL
    # layout is fixed by the mask's shape above, so no choice depends on sizes
    if T > 1 and Cc > 1 and all(t.stride() == (T * Cc, 1, Cc) for t in ts):
        return _MaskedKLFn.apply(*[t.transpose(1, 2) for t in ts], lens, False)
    return _MaskedKLFn.apply(*[t.contiguous() for t in ts], lens, True)
