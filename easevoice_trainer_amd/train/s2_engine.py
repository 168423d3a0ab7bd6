"""The s2 GAN training step (the hot loop of src/train/sovits.py:428-525) on one MI355X.

Differences from the reference that do not change the arithmetic of the losses / updates:
  * weights of every conv are weight-norm folded once per model per step (one launch), not per layer call;
  * D sees [real ; fake] as one batch in the D step; in the G step the real half runs without autograd and
    the D weight gradients (which the reference computes and then discards at the next `optim_d.zero_grad()`,
    sovits.py:503,511-521) are not computed at all;
  * gradients live in one flat arena per model: zeroing, grad-norm, AdamW and the data-parallel all-reduce
    are single launches over it; no `.item()` host syncs inside the step;
  * dtype torch.bfloat16 needs no loss scaling; dtype torch.float16 is the reference's `fp16_run` mode: the same kernels
    built for IEEE half (libevt_hip_f16.so) and the reference's GradScaler protocol (sovits.py:378,504-507,521-525:
    scale -> backward -> unscale_ -> step (skipped on overflow) for D, then for G, then ONE update()) with the scaler's
    state on the device (runtime.DeviceGradScaler): no host read, captured into the HIP graphs with the rest.
"""
import os
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Optional

import torch
from torch.nn import functional as F

from ..hip import lib as L
from ..hip.enc import bump_rng
from ..module import commons
from ..module.losses import (discriminator_loss, discriminator_loss_batched, feature_loss, generator_loss, kl_loss,
                             l1_mean_scaled)
from ..module.mel_processing import mel_spectrogram_torch, spec_to_mel_slices
from ..module.models import MultiPeriodDiscriminator, SynthesizerTrn
from ..runtime import DeviceGradScaler, FlatAdamW, ModelRuntime


@dataclass
class S2Losses:
    disc: torch.Tensor
    gen: torch.Tensor
    fm: torch.Tensor
    mel: torch.Tensor
    kl: torch.Tensor
    kl_ssl: torch.Tensor
    gen_all: torch.Tensor
    grad_sumsq_d: Optional[torch.Tensor] = None
    grad_sumsq_g: Optional[torch.Tensor] = None
    extras: dict = field(default_factory=dict)


class S2Engine:
    def __init__(self, hps: dict, device="cuda:0", dtype=torch.bfloat16, impl=L.IMPL_AUTO, reducer=None, scaler_args=None):
        """hps: the parsed configs/s2.json (train/data/model sections).  scaler_args: GradScaler keyword arguments for
        dtype torch.float16 (the reference constructs it with torch's defaults: init_scale 65536, growth 2 every 2000
        clean steps, backoff 0.5)."""
        self.hps, self.device, self.dtype = hps, torch.device(device), dtype
        L.set_half(dtype)
        self.scaler = DeviceGradScaler(self.device, enabled=dtype == torch.float16, **(scaler_args or {}))
        d, m, t = hps["data"], hps["model"], hps["train"]
        self.net_g = SynthesizerTrn(d["filter_length"] // 2 + 1, t["segment_size"] // d["hop_length"],
                                    n_speakers=d["n_speakers"], **m)
        self.net_d = MultiPeriodDiscriminator(m["use_spectral_norm"])
        self.rt_g = ModelRuntime(self.net_g, dtype, self.device, impl)
        self.rt_d = ModelRuntime(self.net_d, dtype, self.device, impl)
        self.reducer = reducer  # dist.GradReducer or None
        self.optim_g = self.optim_d = None
        self.graphs_enabled = False
        # data parallelism: gradients are reduced sub-model by sub-model on a side stream while the backward of the next
        # sub-model runs (EVT_DP_OVERLAP=0: the two whole-arena reductions between the phases, nothing overlapped)
        self.overlap = (reducer is not None and reducer.active and os.environ.get("EVT_DP_OVERLAP", "1") != "0")
        # EVT_DP_CUT=1: the cut program (nine pieces by default) on one GPU WITHOUT collectives -- what the decomposition itself costs
        # next to the three-phase program (bench.py --dp-program 1)
        self.cut_only = (not self.overlap and os.environ.get("EVT_DP_CUT", "0") == "1")
        # one GPU, EVT_BOOK_PIPE=1: the same cuts, used to run each sub-model's bookkeeping (weight-norm gradient, AdamW,
        # refold) on a side stream under the backward of the next sub-model (ModelRuntime.book_piece).  Off by default:
        # measured round 4, the launches do overlap (kernel trace) and the step does not get shorter -- 24.7-24.8 ms
        # against 24.6-24.7 -- although it is 2.0 ms shorter without them: the backward next to them slows down by as much
        self.pipe = ((reducer is None or not reducer.active) and self.device.type == "cuda" and not self.cut_only
                     and not self.scaler.enabled      # the piped bookkeeping updates a range before the whole arena was checked
                     and os.environ.get("EVT_BOOK_PIPE", "0") == "1")
        # pieces of the data-parallel program (see _program): six discriminator pieces, the generator's backward as one
        self.dp_d_pieces = max(1, int(os.environ.get("EVT_DP_D_PIECES", "6")))
        self.dp_g_pieces = min(3, max(1, int(os.environ.get("EVT_DP_G_PIECES", "1"))))
        if self.overlap or self.pipe or self.cut_only:
            # the generator's autograd graph is only built with its cuts when a piece of its backward is reduced early
            self.net_g.split_backward = self.pipe or self.dp_g_pieces > 1
            nd = len(self.net_d.discriminators)
            self._d_ranges = [self.rt_d.arena.range_of_prefix(f"discriminators.{i}.") for i in range(nd)]
            self._d_convs = [[m for m in d.modules() if hasattr(m, "_slot")] for d in self.net_d.discriminators]
            # the vocoder's convolutions (conv_pre .. conv_post): ~58 MB of the generator's 205 MB, reduced under the flow /
            # encoder backward.  Its 1x1 conditioning layer (dec.cond, a conv of the bank too, registered last) is left
            # out of the early range: its input gradient continues into the style encoder, it is reduced with the rest.
            self._dec_convs = [m for n, m in self.net_g.dec.named_modules() if hasattr(m, "_slot")]
            self._dec_range = self.rt_g.arena.range_of_prefix("dec.", stop_before="dec.cond.")
            # flow + posterior encoder (adjacent in the arena, convolutions only): complete after the first part of the
            # generator's remaining backward, reduced under the prior / style encoders' backward
            self._fq_convs = [m for sub in (self.net_g.flow, self.net_g.enc_q) for m in sub.modules() if hasattr(m, "_slot")]
            flo, fhi = self.rt_g.arena.range_of_prefix("flow.")
            qlo, qhi = self.rt_g.arena.range_of_prefix("enc_q.")
            if fhi != qlo and qhi != flo:
                raise L.EvtError("data-parallel overlap: flow and enc_q are expected to be adjacent in the gradient arena")
            self._fq_range = (min(flo, qlo), max(fhi, qhi))
            # a range is reduced as soon as its convolutions' gradients are finished; a parameter whose gradient only
            # arrives with the final gather (an autograd-owned one) must not sit inside such a range -- it would be
            # reduced before it was written: silent gradient loss under data parallelism
            # row ranges of the same pieces in the banks' tables; dec.cond's rows get their gradient with the vocoder's (its
            # operands are complete then) but are refolded with the rest, after its parameters were updated
            self._d_rows = [self.rt_d.bank.rows_of(c) for c in self._d_convs]
            self._dec_rows = self.rt_g.bank.rows_of(self._dec_convs)
            self._dec_rows_fold = self.rt_g.bank.rows_of([m for m in self._dec_convs if m is not self.net_g.dec.cond])
            self._fq_rows = self.rt_g.bank.rows_of(self._fq_convs)
            early = [(self._dec_range, self.rt_g), (self._fq_range, self.rt_g)] + [(r, self.rt_d) for r in self._d_ranges]
            for (lo, hi), rt in early:
                for p_, view in rt._free:
                    off = (view.data_ptr() - rt.arena.grad.data_ptr()) // 4
                    if lo <= off < hi:
                        raise L.EvtError("data-parallel overlap: an autograd-gathered parameter lies inside an early-"
                                         f"reduced gradient range [{lo}, {hi}) at offset {off}")

    # -- optimisers: 4 groups for G exactly as sovits.py:286-319 (text_embedding / encoder_text / mrte at a
    #    lower lr), everything that receives no gradient (ssl_proj) left out
    @staticmethod
    def g_param_groups(net_g, lr, low):
        """the four AdamW groups of src/train/sovits.py:285-313, by parameter name: everything else / text_embedding /
        encoder_text / mrte (the last three at text_low_lr_rate).  ssl_proj never receives a gradient (models.py:912-921)
        and is not updated, but it sits in the reference's first group, so it keeps its slot in the numbering."""
        names = [n for n, _ in net_g.named_parameters()]
        frozen = {n for n in names if n.startswith("ssl_proj.")}
        te = [n for n in names if n.startswith("enc_p.text_embedding.")]
        et = [n for n in names if n.startswith("enc_p.encoder_text.")]
        mr = [n for n in names if n.startswith("enc_p.mrte.")]
        special = set(te + et + mr)
        slots = [n for n in names if n not in special]
        return [dict(names=[n for n in slots if n not in frozen], slots=slots, lr=lr), dict(names=te, lr=low),
                dict(names=et, lr=low), dict(names=mr, lr=low)]

    def build_optimizers(self):
        t = self.hps["train"]
        lr, low = t["learning_rate"], t["learning_rate"] * t["text_low_lr_rate"]
        self.optim_g = FlatAdamW(self.rt_g.arena, self.g_param_groups(self.net_g, lr, low), betas=tuple(t["betas"]),
                                 eps=t["eps"])
        self.optim_d = FlatAdamW(self.rt_d.arena, [dict(names=[n for n, _ in self.net_d.named_parameters()], lr=lr)],
                                 betas=tuple(t["betas"]), eps=t["eps"])
        return self.optim_g, self.optim_d

    # ------------------------------------------------------------------------------------------------------------
    # The step is three phases with the two gradient exchanges between them:
    #   A  zero grads, fold weights, G forward, mel targets, D step forward + backward          -> all-reduce(D grads)
    #   B  D grad-norm + AdamW, refold D, G step through D forward + backward                   -> all-reduce(G grads)
    #   C  G grad-norm + AdamW
    # Each phase has no host synchronisation and no per-step host argument, so with fixed batch shapes it is captured
    # once into a HIP graph and replayed (enable_graphs): ~4600 launches per step become three graph launches.
    # ------------------------------------------------------------------------------------------------------------
    def _phase_a(self, st, backward=True):
        d, t = self.hps["data"], self.hps["train"]
        hop, seg = d["hop_length"], t["segment_size"]
        net_g, net_d, rt_g, rt_d = self.net_g, self.net_d, self.rt_g, self.rt_d
        bump_rng(self.device)     # new dropout masks for this step (device counter: replays advance it too)
        rt_g.zero_grad()
        rt_d.zero_grad()
        rt_g.prepare()
        rt_d.prepare()
        (st.y_hat, st.kl_ssl, st.ids_slice, st.x_mask, st.z_mask, st.lat, st.q) = net_g(
            st.ssl, st.spec, st.spec_lengths, st.text, st.text_lengths, eps=st.eps, ids_slice=st.ids_slice_in)
        # the two mel spectrograms are inputs of the generator step's mel loss only: beside the discriminator step, on a
        # lane (hip/disc.py; joined at the end of this phase -- a graph of its own)
        lane = None
        if self.device.type == "cuda" and backward and os.environ.get("EVT_MEL_LANE", "1") != "0":
            from ..hip.disc import dec_lane, enc_lane

            # (the second lane when there is one: the first carries half of the sub-discriminators during the D step)
            lane = dec_lane(self.device) or enc_lane(self.device)
        if lane is not None:
            main = torch.cuda.current_stream(self.device)
            lane.wait_stream(main)
            with torch.cuda.stream(lane):
                st.y_mel = spec_to_mel_slices(st.spec, st.ids_slice, seg // hop, d["filter_length"], d["n_mel_channels"],
                                              d["sampling_rate"], d["mel_fmin"], d["mel_fmax"])
                st.y_hat_mel = mel_spectrogram_torch(st.y_hat.squeeze(1), d["filter_length"], d["n_mel_channels"],
                                                     d["sampling_rate"], hop, d["win_length"], d["mel_fmin"], d["mel_fmax"])
            for t in (st.spec, st.ids_slice, st.y_hat):
                t.record_stream(lane)
        else:
            st.y_mel = spec_to_mel_slices(st.spec, st.ids_slice, seg // hop, d["filter_length"], d["n_mel_channels"],
                                          d["sampling_rate"], d["mel_fmin"], d["mel_fmax"])
            st.y_hat_mel = mel_spectrogram_torch(st.y_hat.squeeze(1), d["filter_length"], d["n_mel_channels"],
                                                 d["sampling_rate"], hop, d["win_length"], d["mel_fmin"], d["mel_fmax"])
        st.y_seg = commons.slice_segments_1d(st.y.squeeze(1), st.ids_slice * hop, seg)
        # ---- discriminator step (sovits.py:497-507) ----
        rt_d.bank.weight_grads = True
        if self.device.type == "cuda":
            # logits over [real ; generated] as one tensor per sub-discriminator: the loss writes both halves' gradients
            outs = net_d.forward_batched(st.y_seg, st.y_hat.detach())
            if not backward:
                st.d_losses = [discriminator_loss_batched([o]) for o in outs]
            else:
                st.loss_disc = discriminator_loss_batched(outs)
        else:
            y_d_hat_r, y_d_hat_g, _, _ = net_d(st.y_seg, st.y_hat.detach())
            if not backward:
                st.d_losses = [discriminator_loss([r], [g]) for r, g in zip(y_d_hat_r, y_d_hat_g)]
            else:
                st.loss_disc = discriminator_loss(y_d_hat_r, y_d_hat_g)
        if not backward:
            st.d_done = []
            st.loss_disc = torch.stack([l.detach() for l in st.d_losses]).sum()
            return
        self.scaler.scale(st.loss_disc).backward()
        rt_d.finish_grads()
        if lane is not None:
            main.wait_stream(lane)
            st.y_mel.record_stream(main)
            st.y_hat_mel.record_stream(main)

    def _phase_b(self, st, backward=True):
        t = self.hps["train"]
        net_d, rt_g, rt_d = self.net_d, self.rt_g, self.rt_d
        # fp16 mode: scaler.unscale_(optim_d) -- after the data-parallel sum, so every rank sees the same flag (sovits.py:505)
        self.scaler.unscale_(rt_d)
        st.gss_d = rt_d.grad_sumsq() * (st.inv_world * st.inv_world)   # norm of the AVERAGED gradient, as DDP logs it
        if st.hook_after_d is not None:
            st.hook_after_d()
        if st.do_opt and not getattr(st, "piped", False):
            # 1/world averaging folded into the AdamW launch; scaler.step(optim_d): skipped on an overflow (sovits.py:507)
            self.optim_d.step(grad_scale=st.inv_world, **self._skip(rt_d))
            rt_d.prepare()   # D weights changed: refold before the generator's pass through D
        # ---- generator step (sovits.py:509-525) ----
        rt_d.bank.weight_grads = False
        # D on real + generated audio, feature / generator losses, gradients towards y_hat: one autograd node
        st.loss_gen, st.loss_fm, st.y_d_hat_g = net_d.generator_losses(st.y_seg, st.y_hat)
        z, z_p, m_p, logs_p, m_q, logs_q = st.lat
        if self.device.type == "cuda" and st.y_mel.dtype == st.y_hat_mel.dtype:
            st.loss_mel = l1_mean_scaled(st.y_hat_mel, st.y_mel, t["c_mel"])
        else:
            st.loss_mel = F.l1_loss(st.y_mel, st.y_hat_mel) * t["c_mel"]
        st.loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, st.z_mask, lens=st.spec_lengths) * t["c_kl"]
        st.loss_gen_all = st.loss_gen + st.loss_fm + st.loss_mel + st.kl_ssl * 1 + st.loss_kl
        if not backward:
            return
        self.scaler.scale(st.loss_gen_all).backward()
        rt_d.bank.weight_grads = True
        rt_g.finish_grads()

    def _skip(self, rt):
        return dict(skip=self.scaler.found_inf(rt)) if self.scaler.enabled else {}

    def _phase_c(self, st):
        self.scaler.unscale_(self.rt_g)                  # sovits.py:522
        st.gss_g = self.rt_g.grad_sumsq() * (st.inv_world * st.inv_world)
        if st.do_opt:
            self.optim_g.step(grad_scale=st.inv_world, **self._skip(self.rt_g))     # scaler.step(optim_g), sovits.py:524
            self.scaler.update()                                                    # sovits.py:525: once, after both

    @staticmethod
    def _result(st) -> S2Losses:
        z, z_p, m_p, logs_p, m_q, logs_q = st.lat
        return S2Losses(st.loss_disc.detach(), st.loss_gen.detach(), st.loss_fm.detach(), st.loss_mel.detach(),
                        st.loss_kl.detach(), st.kl_ssl.detach(), st.loss_gen_all.detach(), st.gss_d, st.gss_g,
                        extras=dict(y_hat=st.y_hat.detach(), y_hat_mel=st.y_hat_mel.detach(), y_mel=st.y_mel.detach(),
                                    ids_slice=st.ids_slice, z=z.detach(), z_p=z_p.detach(), m_p=m_p.detach(),
                                    logs_p=logs_p.detach(), m_q=m_q.detach(), logs_q=logs_q.detach(),
                                    d_logits=[o.detach() for o in st.y_d_hat_g], quantized=st.q.detach()))

    def _reduce(self, arena):
        if self.reducer is not None:
            self.reducer.all_reduce(arena.grad)

    # ---- data-parallel variant of the phases: the same arithmetic cut into pieces at the points where a sub-model's
    #      gradients are complete; `_program` pairs every piece with the reductions to start right after it ----
    def _phase_a0(self, st):
        """phase A up to the discriminator losses, one loss per sub-discriminator (their graphs are disjoint in the D
        step: the inputs are detached), so that each can be differentiated -- and its gradients reduced -- on its own"""
        self._phase_a(st, backward=False)

    def _phase_a_bwd(self, st, group):
        """backward of the sub-discriminators `group` (indices; adjacent in the arena and in the bank's rows)"""
        torch.autograd.backward([self.scaler.scale(st.d_losses[i]) for i in group])
        st.d_done.append(self.rt_d.finish_conv_grads([m for i in group for m in self._d_convs[i]]))
        if sum(hi - lo for lo, hi in st.d_done) == sum(hi - lo for lo, hi in self._d_rows):
            self.rt_d.finish_grads(done=st.d_done)

    def _phase_b0(self, st, finish=True):
        """D optimiser, the generator's losses, and the backward through D and the vocoder (down to the cut).
        finish=False: the vocoder's range is not reduced on its own, so its weight-gradient rows are left to the final
        finish_grads -- finishing them here would make the main stream wait for the side stream's deferred weight-gradient
        launches, which otherwise run under the flow / encoder backward (that join, not arithmetic, is what a cut costs)"""
        self._phase_b(st, backward=False)
        self.scaler.scale(st.loss_gen + st.loss_fm + st.loss_mel).backward()
        self.rt_d.bank.weight_grads = True
        st.g_done = [self.rt_g.finish_conv_grads(self._dec_convs)] if finish else []

    def _phase_b1(self, st, finish=True):
        """the generator's backward below the vocoder, first part: KL term + the gradient saved at the vocoder's input
        -> flow, posterior encoder (both end at the second cut: their own copy of the style vector, detached prior
        statistics)"""
        self._bwd_b1(st)
        if finish:
            st.g_done.append(self.rt_g.finish_conv_grads(self._fq_convs))

    def _bwd_b1(self, st):
        z_full, z_cut = self.net_g._cut[0]
        roots, grads = [self.scaler.scale(st.loss_kl + st.kl_ssl * 1)], [None]
        if z_cut.grad is not None:
            roots.append(z_full)
            grads.append(z_cut.grad)
        torch.autograd.backward(roots, grads)

    def _phase_b2(self, st):
        """second part: the gradients that arrived at the cuts -> prior encoder (m_p, logs_p) and style encoder (the style
        vector as the vocoder, the flow and the posterior encoder used it)"""
        self._bwd_b2(st)
        self.rt_g.finish_grads(done=st.g_done)

    def _bwd_b2(self, st):
        (ge, ge_dec), = self.net_g._cut[1:]
        (ge2, ge_fq), (m_p, m_p_cut), (logs_p, logs_p_cut) = self.net_g._cut2
        roots, grads = [], []
        gsum = None
        for leaf in (ge_dec, ge_fq):
            if leaf.grad is not None:
                gsum = leaf.grad if gsum is None else gsum + leaf.grad
        if gsum is not None:
            roots.append(ge)
            grads.append(gsum)
        for full, leaf in ((m_p, m_p_cut), (logs_p, logs_p_cut)):
            if leaf.grad is not None:
                roots.append(full)
                grads.append(leaf.grad)
        if roots:
            torch.autograd.backward(roots, grads)
        self.net_g._cut = self.net_g._cut2 = None

    # ---- one GPU: the whole step as one phase, every sub-model's bookkeeping enqueued right behind its backward ----
    @staticmethod
    def _complement(ivs, n):
        out, at = [], 0
        for lo, hi in sorted(ivs):
            if lo > at:
                out.append((at, lo))
            at = max(at, hi)
        if at < n:
            out.append((at, n))
        return out

    def _phase_pipe(self, st):
        rt_g, rt_d = self.rt_g, self.rt_d
        st.piped = True
        self._phase_a(st, backward=False)
        order = list(reversed(range(len(self._d_convs))))
        for n, i in enumerate(order):
            st.d_losses[i].backward()
            rt_d.book_piece(self.optim_d, [self._d_rows[i]], [self._d_rows[i]], [self._d_ranges[i]], first=n == 0)
        # autograd-owned gradients reach the arena here, BEFORE the leftover ranges are updated (today's discriminators
        # have neither: every parameter is a convolution of the bank inside one of the six ranges)
        rt_d.gather_free_grads()
        left = self._complement(self._d_ranges, rt_d.arena.numel)
        if left:
            rt_d.book_piece(self.optim_d, [], [], left, first=False)
        rt_d.book_join()
        # ---- generator step: D update and refold are done; losses, then the backward in its three parts ----
        self._phase_b(st, backward=False)
        (st.loss_gen + st.loss_fm + st.loss_mel).backward()       # (the piped program is never built in fp16 mode)
        rt_d.bank.weight_grads = True
        rt_g.book_piece(self.optim_g, [self._dec_rows], [self._dec_rows_fold], [self._dec_range], first=True)
        self._bwd_b1(st)
        rt_g.book_piece(self.optim_g, [self._fq_rows], [self._fq_rows], [self._fq_range], first=False)
        self._bwd_b2(st)
        rt_g.gather_free_grads()
        nrows = rt_g.bank._nrows
        rt_g.book_piece(self.optim_g, self._complement([self._dec_rows, self._fq_rows], nrows),
                        self._complement([self._dec_rows_fold, self._fq_rows], nrows),
                        self._complement([self._dec_range, self._fq_range], rt_g.arena.numel), first=False)
        rt_g.book_join()
        st.gss_g = rt_g.grad_sumsq()

    def _d_groups(self):
        """[(sub-discriminator indices, arena lo, arena hi)]: the EVT_DP_D_PIECES groups of neighbouring sub-discriminators
        the data-parallel D step differentiates and reduces one by one, in issue order (last sub-discriminator first)"""
        order = list(reversed(range(len(self._d_convs))))
        nd = min(len(order), self.dp_d_pieces)
        base, extra = divmod(len(order), nd)
        groups, at = [], 0
        for n in range(nd):
            size = base + (1 if n < extra else 0)
            grp = order[at:at + size]
            at += size
            lo, hi = min(self._d_ranges[i][0] for i in grp), max(self._d_ranges[i][1] for i in grp)
            if hi - lo != sum(self._d_ranges[i][1] - self._d_ranges[i][0] for i in grp):
                raise L.EvtError("data-parallel pieces: neighbouring sub-discriminators are expected to be adjacent in the arena")
            groups.append((grp, lo, hi))
        return groups

    def exchange_ranges(self):
        """[(name, fp32 elements)] of the flat gradient ranges the data-parallel step reduces one by one, in issue order
        (bench.py prints the collective plan per range through GradReducer.describe)"""
        if not self.overlap:
            return [("D arena", self.rt_d.arena.grad.numel()), ("G arena", self.rt_g.arena.grad.numel())]
        ng = self.dp_g_pieces
        groups = self._d_groups()                     # the very groups _program reduces (sizes differ a lot: d0 is small)
        out = [(f"D piece {i + 1}/{len(groups)}", hi - lo) for i, (_grp, lo, hi) in enumerate(groups)]
        early = []
        if ng >= 2:
            out.append(("G vocoder", self._dec_range[1] - self._dec_range[0]))
            early.append(self._dec_range)
        if ng == 3:
            out.append(("G flow + posterior encoder", self._fq_range[1] - self._fq_range[0]))
            early.append(self._fq_range)
        for lo, hi in self._complement(early, self.rt_g.arena.grad.numel()):
            out.append(("G rest", hi - lo))
        return out

    def _reduce_async(self, flat, lo, hi):
        self.reducer.all_reduce(flat[lo:hi], async_op=True)

    def _program(self, piped=True):
        """[(phase, host action after it)]: the phases are what a HIP graph captures, the actions run between them.
        piped=False: the caller needs the optimiser calls at their serial places (do_opt=False, hook_after_d)."""
        if self.pipe and piped:
            return [(self._phase_pipe, None)]
        if not self.overlap and not self.pipe and not self.cut_only:
            return [(self._phase_a, lambda: self._reduce(self.rt_d.arena)),
                    (self._phase_b, lambda: self._reduce(self.rt_g.arena)), (self._phase_c, None)]
        # the cut program (the generator's graph is built with its cuts: only this order of backward calls walks all of it)
        gd, gg, red = self.rt_d.arena.grad, self.rt_g.arena.grad, self.reducer
        dp = self.overlap
        prog = [(self._phase_a0, None)]
        order = list(reversed(range(len(self._d_convs))))          # the reference's engine would also end with d0
        # EVT_DP_D_PIECES (1 .. 6, default 6): the six sub-discriminators differentiated -- and reduced -- in that many
        # groups of neighbours; EVT_DP_G_PIECES (1 .. 3, default 1): 1 = the whole generator backward in one piece, 2 = vocoder |
        # the rest, 3 = vocoder | flow + posterior encoder | rest.  Every early piece is a graph boundary at which the main
        # stream waits for the side stream's deferred weight-gradient launches of that piece (which otherwise run under the
        # next sub-model's backward).  Measured on one GPU (profiles/r05_dp_program.txt): the six discriminator pieces are
        # free, an early generator piece costs 0.5-1 ms -- more than the ~0.25-0.5 ms of exchange it would hide on 8 GPUs'
        # xGMI -- so the discriminators' 187 MB are reduced under their own backward and the generator's 205 MB after its.
        ng = self.dp_g_pieces
        groups = self._d_groups()
        for n, (grp, lo, hi) in enumerate(groups):
            last = n == len(groups) - 1

            def after(lo=lo, hi=hi, last=last):
                self._reduce_async(gd, lo, hi)
                if last:
                    red.wait()                                      # optim_d reads the reduced gradients
            prog.append((lambda st, grp=tuple(grp): self._phase_a_bwd(st, grp), after if dp else None))
        dlo, dhi = self._dec_range
        flo, fhi = self._fq_range

        def rest_of(done):
            def after():
                # what is left: everything outside the early ranges (prior / style encoders, dec.cond)
                for lo, hi in self._complement(done, gg.numel()):
                    self._reduce_async(gg, lo, hi)
                red.wait()
            return after
        if ng == 3:
            prog.append((self._phase_b0, (lambda: self._reduce_async(gg, dlo, dhi)) if dp else None))
            prog.append((self._phase_b1, (lambda: self._reduce_async(gg, flo, fhi)) if dp else None))
            prog.append((self._phase_b2, rest_of([(dlo, dhi), (flo, fhi)]) if dp else None))
        elif ng == 2:
            def b12(st):
                self._phase_b1(st, finish=False)
                self._phase_b2(st)
            prog.append((self._phase_b0, (lambda: self._reduce_async(gg, dlo, dhi)) if dp else None))
            prog.append((b12, rest_of([(dlo, dhi)]) if dp else None))
        else:
            # one piece: the plain generator phase (no cuts in the autograd graph: split_backward is off), whole arena after it
            prog.append((self._phase_b, rest_of([]) if dp else None))
        prog.append((self._phase_c, None))
        return prog

    def step(self, ssl, spec, spec_lengths, y, text, text_lengths, eps=None, ids_slice=None, do_opt=True,
             hook_after_d=None) -> S2Losses:
        """One GAN step.  Layouts as in the reference: ssl [B,768,T], spec [B,1025,T], y [B,1,T*hop], text [B,Tt]."""
        L.set_half(self.dtype)
        if self.graphs_enabled and do_opt and hook_after_d is None:
            return self._step_graphed((ssl, spec, spec_lengths, y, text, text_lengths, eps, ids_slice))
        st = SimpleNamespace(ssl=ssl, spec=spec, spec_lengths=spec_lengths, y=y, text=text, text_lengths=text_lengths,
                             eps=eps, ids_slice_in=ids_slice, do_opt=do_opt, hook_after_d=hook_after_d,
                             inv_world=1.0 / self.reducer.world if self.reducer is not None else 1.0)
        for phase, after in self._program(piped=do_opt and hook_after_d is None):
            phase(st)
            if after is not None:
                after()
        return self._result(st)

    # ---- HIP-graph replay of the step for repeated batch shapes ----
    def enable_graphs(self, warmup_steps: int = 2, max_shapes: int = 16):
        """Capture the three phases per distinct batch shape after `warmup_steps` eager steps of that shape and replay
        them afterwards.  Every captured shape keeps its own activation pool in HBM (a few GB each; 288 GB holds the
        bucketed shapes of a run), `max_shapes` bounds it -- further shapes run eagerly."""
        from .. import hip_graphs_safe

        self._graph_warmup, self._graph_max = warmup_steps, max_shapes
        self._graph_cache = {}
        self.graph_steps = {"replayed": 0, "eager": 0}      # how the steps of a run were launched (tools/bench_reader.py)
        if not hip_graphs_safe():
            import warnings

            warnings.warn("HIP-graph replay of the s2 step is off: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not in the "
                          "environment before torch was imported (see easevoice_trainer_amd/__init__.py); launching eagerly")
            return
        self.graphs_enabled = True

    def _room_for_capture(self):
        """a captured shape keeps its own activation pool (a few GB at B = 16 x 4 s): stop capturing new shapes -- they run
        eagerly then -- once less than a sixth of the device memory (at least 16 GB) is free"""
        try:
            free, total = torch.cuda.mem_get_info(self.device)
        except Exception:
            return True
        return free >= max(total // 6, 16 << 30)

    def _step_graphed(self, inputs) -> S2Losses:
        key = tuple((tuple(t.shape), t.dtype) if t is not None else None for t in inputs)
        ent = self._graph_cache.get(key)
        if ent is None:
            ent = self._graph_cache[key] = dict(seen=0, graphs=None)
        if ent["graphs"] is None:
            ent["seen"] += 1
            captured = sum(1 for e in self._graph_cache.values() if e["graphs"] is not None)
            if ent["seen"] <= self._graph_warmup or captured >= self._graph_max or not self._room_for_capture():
                self.graph_steps["eager"] += 1
                self.graphs_enabled = False
                try:
                    return self.step(*inputs)
                finally:
                    self.graphs_enabled = True
            counts = (self.optim_d.step_count, self.optim_g.step_count)
            try:
                self._capture(ent, inputs)
            except Exception as e:   # capture is an optimisation: never let it take the run down
                import warnings

                warnings.warn(f"HIP-graph capture of the s2 step failed ({e!r}); continuing with eager launches")
                self.optim_d.step_count, self.optim_g.step_count = counts
                torch.cuda.synchronize()
                self.graphs_enabled = False
                return self.step(*inputs)
        for dst, src in zip(ent["static"], inputs):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        self.optim_d._segments()    # scheduler changes reach the device tables (in place) before the replay
        self.optim_g._segments()
        if self.pipe:
            # the captured step refolds every image right after its update and holds no fold at its top: parameters
            # written from outside since the last step (load_state_dict, broadcast) are folded here
            self.rt_g.prepare()
            self.rt_d.prepare()
        for g, after in zip(ent["graphs"], ent["after"]):
            g.replay()
            if after is not None:
                after()
        self.optim_d.note_replayed_step()
        self.optim_g.note_replayed_step()
        if self.pipe:
            self.rt_g.mark_folded()
            self.rt_d.mark_folded()
        self.graph_steps["replayed"] += 1
        return ent["result"]

    def _capture(self, ent, inputs):
        static = [t.clone() if t is not None else None for t in inputs]
        ssl, spec, spec_lengths, y, text, text_lengths, eps, ids_slice = static
        st = SimpleNamespace(ssl=ssl, spec=spec, spec_lengths=spec_lengths, y=y, text=text, text_lengths=text_lengths,
                             eps=eps, ids_slice_in=ids_slice, do_opt=True, hook_after_d=None,
                             inv_world=1.0 / self.reducer.world if self.reducer is not None else 1.0)
        self.optim_d._segments()   # device tables exist before capture
        self.optim_g._segments()
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        graphs, afters = [], []
        for phase, after in self._program():
            g = torch.cuda.CUDAGraph()
            # "relaxed": pinned staging buffers of the loss tables may be allocated while capturing
            with torch.cuda.graph(g, pool=pool, capture_error_mode="relaxed"):
                phase(st)
            graphs.append(g)
            afters.append(after)
        # capture records the optimiser launches without running them: undo the python-side counters it bumped
        self.optim_d.step_count -= 1
        self.optim_g.step_count -= 1
        # only detached results are kept: a live autograd graph would pin its AccumulateGrad nodes (created on the capture
        # stream) and later eager steps of other shapes would run their gradient accumulation on that stream
        ent.update(graphs=tuple(graphs), after=tuple(afters), static=static, result=self._result(st))
