"""Fixture of the s2 fp16 mode: the REFERENCE's own modules run through the reference's training-loop
body (src/train/sovits.py:459-525, restated line by line below: autocast regions, the fp32 islands at :498 and :512, ONE
GradScaler shared by both optimisers -- scale / backward / unscale_ / step for D, then for G, then update) for three steps
on the C1 batch, with torch's own autocast (device "cpu", dtype float16: convolutions and matrix products in IEEE half)
and torch's own torch.amp.GradScaler and torch.optim.AdamW.  Build container only (/root/reference is imported read-only
through oracle/refshim.py); the GPU test tests/test_s2_fp16_gpu.py reads the .pt this writes.

The scaler is constructed with init_scale = 2**40, backoff_factor = 2**-32, growth_interval = 2 (the reference uses torch's
defaults, 65536 / 0.5 / 2000: the ARITHMETIC of scale / unscale / skip / update does not depend on the constants, and these
make a three-step run visit every branch with wide margins): step 1 overflows in both backward passes (a loss scaled by
2**40 cannot be differentiated in half precision) -> both optimiser steps are skipped, the scale backs off to 2**8; steps
2 and 3 are clean -> both optimisers step, and after the second clean step the scale grows to 2**9.

  python tests/golden/make_golden_fp16.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]
import make_golden as MG  # noqa: E402  (installs the shims)
from make_golden import refshim  # noqa: E402
from make_golden_r2 import checksums, g_groups  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_fill import fill_module, s2_batch  # noqa: E402

SCALER = dict(init_scale=2.0 ** 40, backoff_factor=2.0 ** -32, growth_factor=2.0, growth_interval=2)
SLICES_G = ["dec.conv_post.weight", "dec.resblocks.14.convs1.0.weight_v", "dec.ups.0.weight_v", "flow.flows.0.enc.in_layers.0.bias",
            "enc_q.enc.res_skip_layers.15.weight_g", "enc_p.text_embedding.weight", "enc_p.mrte.c_post.weight",
            "ref_enc.fc.fc.bias"]
SLICES_D = ["discriminators.0.convs.0.weight_v", "discriminators.3.conv_post.bias", "discriminators.5.convs.1.weight_v"]


def main(B=2, T=100, t_text=40, steps=3):
    from src.easevoice.module import commons, models
    from src.easevoice.module.losses import discriminator_loss, feature_loss, generator_loss, kl_loss
    from src.easevoice.module.mel_processing import mel_spectrogram_torch, spec_to_mel_torch, spectrogram_torch

    cfg = json.load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "s2.json")))
    t = cfg["train"]
    lr, low = t["learning_rate"], t["learning_rate"] * t["text_low_lr_rate"]
    net_g = models.SynthesizerTrn(1025, 32, n_speakers=300, **cfg["model"])
    net_d = models.MultiPeriodDiscriminator(False)
    fill_module(net_g, 1)
    fill_module(net_d, 2)
    MG.zero_dropout(net_g)
    net_g.train(); net_d.train()
    optim_g = torch.optim.AdamW(g_groups(net_g, lr, low), lr, betas=t["betas"], eps=t["eps"])
    optim_d = torch.optim.AdamW(net_d.parameters(), lr, betas=t["betas"], eps=t["eps"])
    scaler = torch.amp.GradScaler("cpu", **SCALER)
    b = s2_batch(B, T, t_text)
    spec = spectrogram_torch(b["wav"].squeeze(1), 2048, 32000, 640, 2048, center=False)
    eps, ids = b["eps"], b["ids_slice"]
    orig_randn_like, orig_rand_slice = torch.randn_like, commons.rand_slice_segments
    autocast = lambda enabled=True: torch.autocast("cpu", dtype=torch.float16, enabled=enabled)
    p0_g = {n: p.detach().clone() for n, p in net_g.named_parameters()}
    p0_d = {n: p.detach().clone() for n, p in net_d.named_parameters()}
    out = dict(config=dict(B=B, T=T, t_text=t_text, steps=steps), scaler=SCALER, steps=[])
    for step in range(steps):
        torch.randn_like = lambda x, **kw: eps.to(x.dtype) if x.shape == eps.shape else orig_randn_like(x, **kw)
        commons.rand_slice_segments = lambda x, x_lengths=None, segment_size=4: (commons.slice_segments(x, ids, segment_size), ids)
        try:
            # ---- sovits.py:459-496 ----
            with autocast():
                (y_hat, kl_ssl, ids_slice, x_mask, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), stats_ssl) = net_g(
                    b["ssl"], spec, b["lengths"], b["text"], b["text_lengths"])
                mel = spec_to_mel_torch(spec, 2048, 128, 32000, 0.0, None)
                y_mel = commons.slice_segments(mel, ids_slice, 32)
                y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1).float(), 2048, 128, 32000, 640, 2048, 0.0, None)
                y = commons.slice_segments(b["wav"], ids_slice * 640, 20480)
                # ---- discriminator, sovits.py:497-507 ----
                y_d_hat_r, y_d_hat_g, _, _ = net_d(y, y_hat.detach())
                with autocast(False):
                    loss_disc, _, _ = discriminator_loss(y_d_hat_r, y_d_hat_g)
                    loss_disc_all = loss_disc
        finally:
            torch.randn_like, commons.rand_slice_segments = orig_randn_like, orig_rand_slice
        optim_d.zero_grad()
        scaler.scale(loss_disc_all).backward()
        scaler.unscale_(optim_d)
        inf_d = float(sum(v.item() for v in scaler._per_optimizer_states[id(optim_d)]["found_inf_per_device"].values()))
        gss_d = float(sum(p.grad.double().pow(2).sum() for p in net_d.parameters() if p.grad is not None))
        scaler.step(optim_d)
        # ---- generator, sovits.py:509-525 ----
        with autocast():
            y_d_hat_r, y_d_hat_g, fmap_r, fmap_g = net_d(y, y_hat)
            with autocast(False):
                loss_mel = torch.nn.functional.l1_loss(y_mel.float(), y_hat_mel.float()) * t["c_mel"]
                loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * t["c_kl"]
                loss_fm = feature_loss(fmap_r, fmap_g)
                loss_gen, _ = generator_loss(y_d_hat_g)
                loss_gen_all = loss_gen + loss_fm + loss_mel + kl_ssl * 1 + loss_kl
        optim_g.zero_grad()
        scaler.scale(loss_gen_all).backward()
        scaler.unscale_(optim_g)
        inf_g = float(sum(v.item() for v in scaler._per_optimizer_states[id(optim_g)]["found_inf_per_device"].values()))
        gss_g = float(sum(p.grad.double().pow(2).sum() for p in net_g.parameters() if p.grad is not None))
        scale_used = scaler.get_scale()
        scaler.step(optim_g)
        scaler.update()
        rec = dict(losses=dict(disc=float(loss_disc), gen=float(loss_gen), fm=float(loss_fm), mel=float(loss_mel),
                               kl=float(loss_kl), gen_all=float(loss_gen_all)),
                   found_inf=dict(d=inf_d > 0, g=inf_g > 0), scale_used=scale_used, scale_after=scaler.get_scale(),
                   growth_tracker=int(scaler._growth_tracker.item()),
                   grad_sumsq=dict(d=gss_d, g=gss_g),
                   opt_steps=dict(d=int(max([float(s["step"]) for s in optim_d.state.values()] or [0])),
                                  g=int(max([float(s["step"]) for s in optim_g.state.values()] or [0]))),
                   y_hat_dtype=str(y_hat.dtype), y_hat=y_hat.detach().float().squeeze(1)[:, ::37].clone())
        if not (inf_d > 0):
            rec["d_grad_slices"] = {n: p.grad.flatten()[:64].clone() for n, p in net_d.named_parameters() if n in SLICES_D}
        if not (inf_g > 0):
            rec["g_grad_slices"] = {n: p.grad.flatten()[:64].clone() for n, p in net_g.named_parameters() if n in SLICES_G}
        out["steps"].append(rec)
        print(step, rec["losses"], rec["found_inf"], rec["scale_used"], "->", rec["scale_after"], rec["opt_steps"],
              {k: f"{v:.4g}" for k, v in rec["grad_sumsq"].items()}, flush=True)
    out["post_g"], out["post_d"] = checksums(net_g), checksums(net_d)
    # the direction of the total update of selected tensors (AdamW's first steps are sign-like: compare directions)
    out["delta_g"] = {n: (p.detach() - p0_g[n]).flatten()[:256].clone() for n, p in net_g.named_parameters() if n in SLICES_G}
    out["delta_d"] = {n: (p.detach() - p0_d[n]).flatten()[:256].clone() for n, p in net_d.named_parameters() if n in SLICES_D}
    out["delta_sumsq_g"] = {}
    for n, p in net_g.named_parameters():
        top = n.split(".")[0]
        out["delta_sumsq_g"][top] = out["delta_sumsq_g"].get(top, 0.0) + float((p.detach() - p0_g[n]).double().pow(2).sum())
    out["delta_sumsq_d"] = float(sum((p.detach() - p0_d[n]).double().pow(2).sum() for n, p in net_d.named_parameters()))
    path = os.path.join(HERE, "s2_c1_fp16.pt")
    torch.save(out, path)
    print("wrote", path)


if __name__ == "__main__":
    main()
