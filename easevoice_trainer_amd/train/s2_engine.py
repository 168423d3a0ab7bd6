"""The s2 GAN training step (the hot loop of src/train/sovits.py:428-525) on one MI355X.

Differences from the reference that do not change the arithmetic of the losses / updates:
  * weights of every conv are weight-norm folded once per model per step (one launch), not per layer call;
  * D sees [real ; fake] as one batch in the D step; in the G step the real half runs without autograd and
    the D weight gradients (which the reference computes and then discards at the next `optim_d.zero_grad()`,
    sovits.py:503,511-521) are not computed at all;
  * gradients live in one flat arena per model: zeroing, grad-norm, AdamW and the data-parallel all-reduce
    are single launches over it; no `.item()` host syncs inside the step;
  * bf16 compute needs no GradScaler (the reference's fp16 autocast + GradScaler is SURVEY §8(f) N4).
"""
from dataclasses import dataclass, field
from typing import Optional

import torch
from torch.nn import functional as F

from ..hip import lib as L
from ..module import commons
from ..module.losses import discriminator_loss, feature_loss, generator_loss, kl_loss
from ..module.mel_processing import mel_spectrogram_torch, spec_to_mel_torch
from ..module.models import MultiPeriodDiscriminator, SynthesizerTrn
from ..runtime import FlatAdamW, ModelRuntime


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
    def __init__(self, hps: dict, device="cuda:0", dtype=torch.bfloat16, impl=L.IMPL_AUTO, reducer=None):
        """hps: the parsed configs/s2.json (train/data/model sections)."""
        self.hps, self.device, self.dtype = hps, torch.device(device), dtype
        d, m, t = hps["data"], hps["model"], hps["train"]
        self.net_g = SynthesizerTrn(d["filter_length"] // 2 + 1, t["segment_size"] // d["hop_length"],
                                    n_speakers=d["n_speakers"], **m)
        self.net_d = MultiPeriodDiscriminator(m["use_spectral_norm"])
        self.rt_g = ModelRuntime(self.net_g, dtype, self.device, impl)
        self.rt_d = ModelRuntime(self.net_d, dtype, self.device, impl)
        self.reducer = reducer  # dist.GradReducer or None
        self.optim_g = self.optim_d = None

    # -- optimisers: 4 groups for G exactly as sovits.py:286-319 (text_embedding / encoder_text / mrte at a
    #    lower lr), everything that receives no gradient (ssl_proj) left out
    def build_optimizers(self):
        t = self.hps["train"]
        lr, low = t["learning_rate"], t["learning_rate"] * t["text_low_lr_rate"]
        names = [n for n, _ in self.net_g.named_parameters()]
        frozen = [n for n in names if n.startswith("ssl_proj.")]
        te = [n for n in names if n.startswith("enc_p.text_embedding.")]
        et = [n for n in names if n.startswith("enc_p.encoder_text.")]
        mr = [n for n in names if n.startswith("enc_p.mrte.")]
        special = set(te + et + mr + frozen)
        base = [n for n in names if n not in special]
        self.optim_g = FlatAdamW(self.rt_g.arena, [dict(names=base, lr=lr), dict(names=te, lr=low),
                                                   dict(names=et, lr=low), dict(names=mr, lr=low)],
                                 betas=tuple(t["betas"]), eps=t["eps"])
        self.optim_d = FlatAdamW(self.rt_d.arena, [dict(names=[n for n, _ in self.net_d.named_parameters()], lr=lr)],
                                 betas=tuple(t["betas"]), eps=t["eps"])
        return self.optim_g, self.optim_d

    def step(self, ssl, spec, spec_lengths, y, text, text_lengths, eps=None, ids_slice=None, do_opt=True,
             hook_after_d=None) -> S2Losses:
        """One GAN step.  Layouts as in the reference: ssl [B,768,T], spec [B,1025,T], y [B,1,T*hop], text [B,Tt]."""
        d, t = self.hps["data"], self.hps["train"]
        hop, seg = d["hop_length"], t["segment_size"]
        net_g, net_d, rt_g, rt_d = self.net_g, self.net_d, self.rt_g, self.rt_d
        rt_g.zero_grad()
        rt_d.zero_grad()
        rt_g.prepare()
        rt_d.prepare()

        (y_hat, kl_ssl, ids_slice, x_mask, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), _q) = net_g(
            ssl, spec, spec_lengths, text, text_lengths, eps=eps, ids_slice=ids_slice)
        mel = spec_to_mel_torch(spec, d["filter_length"], d["n_mel_channels"], d["sampling_rate"], d["mel_fmin"],
                                d["mel_fmax"])
        y_mel = commons.slice_segments(mel.transpose(1, 2), ids_slice, seg // hop).transpose(1, 2)
        y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1), d["filter_length"], d["n_mel_channels"],
                                          d["sampling_rate"], hop, d["win_length"], d["mel_fmin"], d["mel_fmax"])
        y_seg = commons.slice_segments_1d(y.squeeze(1), ids_slice * hop, seg)

        # ---- discriminator step (sovits.py:497-507) ----
        rt_d.bank.weight_grads = True
        y_d_hat_r, y_d_hat_g, _, _ = net_d(y_seg, y_hat.detach())
        loss_disc = discriminator_loss(y_d_hat_r, y_d_hat_g)
        loss_disc.backward()
        rt_d.finish_grads()
        inv_world = 1.0
        if self.reducer is not None:
            # averaged inside the AdamW launch (grad_scale); the generator-side work below does not depend on it
            self.reducer.all_reduce(rt_d.arena.grad)
            inv_world = 1.0 / self.reducer.world
        gss_d = rt_d.grad_sumsq().clone()
        if hook_after_d is not None:
            hook_after_d()
        if do_opt:
            self.optim_d.step(grad_scale=inv_world)
            rt_d.prepare()   # D weights changed: refold before the generator's pass through D

        # ---- generator step (sovits.py:509-525) ----
        rt_d.bank.weight_grads = False
        with torch.no_grad():
            _, fmap_r = net_d.forward_single(y_seg)
        y_d_hat_g, fmap_g = net_d.forward_single(y_hat)
        loss_mel = F.l1_loss(y_mel, y_hat_mel) * t["c_mel"]
        loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * t["c_kl"]
        loss_fm = feature_loss(fmap_r, fmap_g)
        loss_gen = generator_loss(y_d_hat_g)
        loss_gen_all = loss_gen + loss_fm + loss_mel + kl_ssl * 1 + loss_kl
        loss_gen_all.backward()
        rt_d.bank.weight_grads = True
        rt_g.finish_grads()
        if self.reducer is not None:
            self.reducer.all_reduce(rt_g.arena.grad)
        gss_g = rt_g.grad_sumsq().clone()
        if do_opt:
            self.optim_g.step(grad_scale=inv_world)
        return S2Losses(loss_disc.detach(), loss_gen.detach(), loss_fm.detach(), loss_mel.detach(), loss_kl.detach(),
                        kl_ssl.detach(), loss_gen_all.detach(), gss_d, gss_g,
                        extras=dict(y_hat=y_hat.detach(), y_hat_mel=y_hat_mel.detach(), y_mel=y_mel.detach(),
                                    ids_slice=ids_slice, z=z.detach(), z_p=z_p.detach(), m_p=m_p.detach(),
                                    logs_p=logs_p.detach(), m_q=m_q.detach(), logs_q=logs_q.detach(),
                                    d_logits=[o.detach() for o in y_d_hat_g], quantized=_q.detach()))
