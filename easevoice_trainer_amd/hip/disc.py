"""The discriminators' part of the GENERATOR step (src/train/sovits.py:509-516: net_d(y, y_hat) -> feature_loss +
generator_loss) as ONE autograd node over all sub-discriminators.

The reference runs D on the real and on the generated audio and differentiates the two losses through the generated
half only (the real feature maps are detached, losses.py:11).  Here, per sub-discriminator:
  * forward: real and generated sequences go through every convolution as ONE batch [real ; fake] -- the wide
    1024-channel layers run at the efficiency of the D step's batched pass (a fake-only launch fills half the chip);
  * losses: one launch for all 37 feature-map L1 terms (pointers to the two halves of each map), one for the six LSGAN
    terms;
  * backward: backward-data launches over the FAKE half only (the halves are contiguous), each adding the feature-loss
    gradient of its input map through the add-epilogue; no weight gradients (the reference computes and discards them),
    no zero-padded slice gradients, no element-wise sums, ~45 autograd nodes less per sub-discriminator.
"""
import ctypes as C
import os

import torch

from . import conv as HC
from . import lib as L

# The six sub-discriminators are independent between the fold of their inputs and the loss launches: with
# EVT_MPD_STREAMS = n > 1 they are dealt round-robin onto the current stream and n - 1 side streams (forward and
# backward of the generator step's node), so that one's under-filled launches (backward-data over the generated half:
# 264 blocks on 512 slots; the 520-block period-7 layers) run beside another's.  Every tensor a branch allocates
# is allocated and freed on the branch's own stream; what the joining stream reads afterwards is record_stream()ed.
_ALL = os.environ.get("EVT_BRANCH_STREAMS", "1") != "0"       # master switch: "0" = the one-stream step, whatever the others say
MPD_STREAMS = int(os.environ.get("EVT_MPD_STREAMS", "2")) if _ALL else 1
_side = {}


def _new_stream(dev, k):
    """lane k of the device: a library-owned HIP stream (hip/lib.py::role_stream says why not a pooled torch stream)"""
    return L.role_stream(dev, f"lane{k}", ring=1)


def _branches(dev, n_items):
    """stream of every sub-discriminator (None = the current stream) and the distinct side streams among them"""
    n = MPD_STREAMS if (dev.type == "cuda" and HC.TRACE is None) else 1
    if n <= 1:
        return [None] * n_items, []
    pool = _side.setdefault(dev, [])
    while len(pool) < n - 1:
        pool.append(_new_stream(dev, len(pool)))
    lanes = [None] + pool[: n - 1]
    per = [lanes[i % n] for i in range(n_items)]
    return per, pool[: n - 1]


ENC_STREAM = _ALL and os.environ.get("EVT_ENC_STREAM", "1") != "0"


def enc_lane(dev):
    """side stream for the prior encoder of SynthesizerTrn.forward (independent of posterior encoder / flow / vocoder until
    the KL term), or None"""
    if not ENC_STREAM or dev.type != "cuda" or HC.TRACE is not None:
        return None
    pool = _side.setdefault(dev, [])
    if not pool:
        pool.append(_new_stream(dev, 0))
    return pool[0]


DEC_STREAM = _ALL and os.environ.get("EVT_DEC_STREAM", "1") != "0"


def dec_lane(dev):
    """side stream for two of the three parallel ResBlocks of a wide vocoder stage (models.py:461-466: xs = sum of the blocks'
    outputs; the narrow stages run the three as grouped launches instead), or None"""
    if not DEC_STREAM or dev.type != "cuda" or HC.TRACE is not None:
        return None
    pool = _side.setdefault(dev, [])
    while len(pool) < 2:
        pool.append(_new_stream(dev, len(pool)))
    return pool[1]


class _On:
    """`with _On(stream)`: torch.cuda.stream(stream), or nothing for None"""

    def __init__(self, s):
        self.cm = torch.cuda.stream(s) if s is not None else None

    def __enter__(self):
        if self.cm is not None:
            self.cm.__enter__()

    def __exit__(self, *a):
        if self.cm is not None:
            self.cm.__exit__(*a)


def mpd_fold(periods, cd, src0, src1=None):
    """[n, T] waveform batches (fp32 or bf16; src1 stacked behind src0) -> the prepared input of every sub-discriminator
    in ONE launch (csrc/mpd_fold.hip): for period p a [(n0 + n1) * p, ceil(T / p), 1] tensor of dtype `cd`, period 1 =
    DiscriminatorS.  No gradient (MPDFoldFn / MPDGenLossFn.backward call mpd_unfold)."""
    src0 = src0.contiguous()
    n0, T = src0.shape
    n1, dt1 = 0, L.dt_of(src0)
    if src1 is not None:
        src1 = src1.contiguous()
        n1, dt1 = src1.size(0), L.dt_of(src1)
    outs = [torch.empty(((n0 + n1) * p, (T + p - 1) // p, 1), dtype=cd, device=src0.device) for p in periods]
    per = (C.c_int32 * len(periods))(*periods)
    ptrs = (C.c_void_p * len(periods))(*[o.data_ptr() for o in outs])
    L.check(L.lib().evt_mpd_fold(L.dt_of(src0), L.ptr(src0), n0, dt1, L.ptr(src1), n1, T, per, len(periods), ptrs,
                                 L.dt_of(outs[0]), L.stream_ptr()), "evt_mpd_fold")
    return outs


def mpd_unfold(periods, grads, b0, n, T, dtype):
    """sum of the prepared batches' gradients (rows of items b0 .. b0 + n - 1) back onto the [n, T] waveform"""
    grads = [g.contiguous() for g in grads]
    out = torch.empty((n, T), dtype=dtype, device=grads[0].device)
    per = (C.c_int32 * len(periods))(*periods)
    ptrs = (C.c_void_p * len(periods))(*[g.data_ptr() for g in grads])
    L.check(L.lib().evt_mpd_unfold(L.dt_of(grads[0]), ptrs, per, len(periods), b0, n, T, L.dt_of(out), L.ptr(out),
                                   L.stream_ptr()), "evt_mpd_unfold")
    return out


class MPDFoldFn(torch.autograd.Function):
    """differentiable mpd_fold of ONE waveform batch (the D path through the per-layer conv nodes; the generator step
    folds inside MPDGenLossFn)"""

    @staticmethod
    def forward(ctx, x, periods, cd):
        ctx.periods, ctx.shape, ctx.dtype = periods, tuple(x.shape), x.dtype
        return tuple(mpd_fold(periods, cd, x))

    @staticmethod
    def backward(ctx, *grads):
        n, T = ctx.shape
        gs = []
        for g, p in zip(grads, ctx.periods):
            gs.append(g if g is not None else torch.zeros((n * p, (T + p - 1) // p, 1), dtype=grads[0].dtype,
                                                          device=grads[0].device))
        return mpd_unfold(ctx.periods, gs, 0, n, T, ctx.dtype), None, None


def _seg_table(a, b, da, scales, device):
    segs = [L.Seg(x.data_ptr(), y.data_ptr() if y is not None else None, g.data_ptr() if g is not None else None,
                  x.numel(), sc, 0) for x, y, g, sc in zip(a, b, da, scales)]
    return L.struct_to_device(segs, device)


class MPDGenLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, plan, periods, y_hat, y_real):
        """plan: per sub-discriminator (conv slots, (act, slope) per conv); periods: its input period (1 = DiscriminatorS);
        y_hat [n, T] the generated waveforms (differentiable), y_real [n, T].
        Returns (loss_gen, loss_fm, *generated logits [n, -1])."""
        nd = len(plan)
        n_items, T = y_hat.shape
        dev = y_hat.device
        cd = plan[0][0][0].bank.dtype
        dt = L.dt_code(cd)
        # [real ; generated] of every sub-discriminator, prepared in one launch
        both = mpd_fold(periods, cd, y_real, y_hat)
        maps, saved = [], []           # per sub-D: list of batched outputs
        lanes, sides = _branches(dev, nd)
        main = torch.cuda.current_stream(dev) if sides else None
        for st in sides:
            st.wait_stream(main)
        for (slots, acts), x, lane in zip(plan, both, lanes):
            ys = []
            with _On(lane):
                for s, (act, slope) in zip(slots, acts):
                    x = HC._fwd(s, x, None, 1.0, act, slope)
                    ys.append(x)
            if lane is not None:
                for y in ys:
                    y.record_stream(main)
            maps.append(ys)
        for st in sides:
            main.wait_stream(st)
        fa, fb, fs, la, ls = [], [], [], [], []
        for ys in maps:
            for y in ys:
                h = y.size(0) // 2
                fa.append(y[h:])
                fb.append(y[:h])
                fs.append(2.0 / y[h:].numel())           # losses.py:7-15: 2 * mean|r - g| per map
            lg = ys[-1][ys[-1].size(0) // 2:]
            la.append(lg)
            ls.append(1.0 / lg.numel())                   # losses.py:35-43: mean((1 - dg)^2)
        out = torch.zeros(2, dtype=torch.float32, device=dev)
        L.check(L.lib().evt_l1_multi_fwd(dt, L.ptr(_seg_table(fa, fb, [None] * len(fa), fs, dev)), len(fa),
                                         L.ptr(out[1:]), L.stream_ptr()), "evt_l1_multi_fwd")
        L.check(L.lib().evt_lsgan_multi_fwd(dt, L.ptr(_seg_table(la, [None] * nd, [None] * nd, ls, dev)), nd,
                                            C.c_float(1.0), L.ptr(out[:1]), L.stream_ptr()), "evt_lsgan_multi_fwd")
        ctx.plan, ctx.dt = plan, dt
        ctx.counts = [len(ys) for ys in maps]
        ctx.in_shapes = [(x.size(0) // 2, x.size(1), 1) for x in both]
        ctx.periods, ctx.wav = periods, (n_items, T, y_hat.dtype)
        ctx.save_for_backward(*[y for ys in maps for y in ys])
        logits = [lg.reshape(n_items, -1).detach() for lg in la]
        ctx.mark_non_differentiable(*logits)
        return (out[0], out[1], *logits)

    @staticmethod
    def backward(ctx, dgen, dfm, *_unused):
        plan, dt = ctx.plan, ctx.dt
        flat = list(ctx.saved_tensors)
        maps, at = [], 0
        for c in ctx.counts:
            maps.append(flat[at: at + c])
            at += c
        dev = flat[0].device
        nd = len(plan)
        # feature-loss gradients of every generated map, LSGAN gradients of the generated logits: two launches
        fa, fb, fs, fg, la, ls, lgr = [], [], [], [], [], [], []
        for ys in maps:
            for y in ys:
                h = y.size(0) // 2
                fa.append(y[h:])
                fb.append(y[:h])
                fs.append(2.0 / y[h:].numel())
                fg.append(torch.empty_like(y[h:]))
            lg = ys[-1][ys[-1].size(0) // 2:]
            la.append(lg)
            ls.append(1.0 / lg.numel())
            lgr.append(torch.empty_like(lg))
        dl = torch.stack([dgen.reshape(()).float(), dfm.reshape(()).float()]).contiguous()
        L.check(L.lib().evt_l1_multi_bwd(dt, L.ptr(_seg_table(fa, fb, fg, fs, dev)), len(fa), L.ptr(dl[1:]),
                                         L.stream_ptr()), "evt_l1_multi_bwd")
        L.check(L.lib().evt_lsgan_multi_bwd(dt, L.ptr(_seg_table(la, [None] * nd, lgr, ls, dev)), nd, C.c_float(1.0),
                                            L.ptr(dl[:1]), L.stream_ptr()), "evt_lsgan_multi_bwd")
        grads, gi = [], 0
        lanes, sides = _branches(dev, nd)
        main = torch.cuda.current_stream(dev) if sides else None
        for st in sides:
            st.wait_stream(main)
        for (slots, acts), ys, lg_grad, in_shape, lane in zip(plan, maps, lgr, ctx.in_shapes, lanes):
            nl = len(slots)
            mg = fg[gi: gi + nl]
            gi += nl
            with _On(lane):
                dy = _branch_bwd(slots, acts, ys, mg, lg_grad, in_shape, dt)
            if lane is not None:
                dy.record_stream(main)
            grads.append(dy)
        for st in sides:
            main.wait_stream(st)
        n_items, T, wdt = ctx.wav
        return None, None, None, mpd_unfold(ctx.periods, grads, 0, n_items, T, wdt), None


def _branch_bwd(slots, acts, ys, mg, lg_grad, in_shape, dt):
    """backward-data chain of one sub-discriminator over the generated half; returns the gradient of its prepared input"""
    nl = len(slots)
    dy = mg[-1] + lg_grad                                    # the last map is also the logits (a few K values)
    for l in reversed(range(nl)):
        s, (act, slope) = slots[l], acts[l]
        y_act = ys[l][ys[l].size(0) // 2:]
        half = y_act.size(0)
        lin = ys[l - 1].size(1) if l > 0 else in_shape[1]
        add = mg[l - 1] if l > 0 else None
        a_kind, a_slope, y_arg = act, slope, (y_act if act != L.ACT_NONE else None)
        if act != L.ACT_NONE and L.lib().evt_conv1d_wants_plain_dy(C.byref(s.params(half, lin, 1.0, act, slope))):
            dy_eff = torch.empty_like(dy)
            L.check(L.lib().evt_dact_mul(dt, L.ptr(dy), L.ptr(y_act), int(act), C.c_float(slope), L.ptr(dy_eff),
                                         C.c_int64(dy.numel()), L.stream_ptr()), "evt_dact_mul")
            dy, y_arg, a_kind, a_slope = dy_eff, None, L.ACT_NONE, 1.0
        if add is not None and s.module.groups > 1:
            # the grouped kernels take no add operand (the dispatcher would fall back to the generic path)
            dy = HC._bwd_data(s, dy.contiguous(), y_arg, None, None, half, lin, 1.0, a_kind, a_slope)
            dy.add_(add)
        else:
            dy = HC._bwd_data(s, dy.contiguous(), y_arg, None, add, half, lin, 1.0, a_kind, a_slope)
    return dy
