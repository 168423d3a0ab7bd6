"""Generates the committed golden fixtures by running the REFERENCE's own modules (imported read-only from
/root/reference with the three shims of oracle/refshim.py) on seeded synthetic batches.

Run in the build container only:   python tests/golden/make_golden.py [s2] [s1] [keys]
The GPU box has no /root/reference; tests read the .pt/.json files this script wrote.
Determinism recipe: every nn.Dropout.p = 0 and attention dropout = 0, quantizer codebook marked
initialised, the randn_like / rand draws of models.py:358 and commons.py:55 are replaced by injected tensors.
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_fill import fill_module, s1_batch, s2_batch  # noqa: E402

torch.set_num_threads(8)


def zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "dropout") and isinstance(getattr(mod, "dropout"), float):
            mod.dropout = 0.0
        if hasattr(mod, "p_dropout") and isinstance(getattr(mod, "p_dropout"), float):
            mod.p_dropout = 0.0


def grad_stats(module):
    """sum-of-squares of gradients per top-level child + a few raw slices"""
    out = {}
    for n, p in module.named_parameters():
        top = n.split(".")[0]
        if p.grad is not None:
            out[top] = out.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    return out


def make_s2(B=2, T=100, t_text=40, tag="c1", hook=None, hook_d=None):
    from src.easevoice.module import commons, models
    from src.easevoice.module.losses import discriminator_loss, feature_loss, generator_loss, kl_loss
    from src.easevoice.module.mel_processing import mel_spectrogram_torch, spec_to_mel_torch, spectrogram_torch

    cfg = json.load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "s2.json")))
    net_g = models.SynthesizerTrn(1025, 32, n_speakers=300, **cfg["model"])
    net_d = models.MultiPeriodDiscriminator(False)
    fill_module(net_g, 1)
    fill_module(net_d, 2)
    zero_dropout(net_g)
    net_g.train(); net_d.train()
    b = s2_batch(B, T, t_text)
    spec = spectrogram_torch(b["wav"].squeeze(1), 2048, 32000, 640, 2048, center=False)
    assert spec.shape == (B, 1025, T), spec.shape

    # inject the two random draws
    eps, ids = b["eps"], b["ids_slice"]
    orig_randn_like, orig_rand_slice = torch.randn_like, commons.rand_slice_segments
    torch.randn_like = lambda t, **kw: eps.to(t.dtype) if t.shape == eps.shape else orig_randn_like(t, **kw)
    commons.rand_slice_segments = lambda x, x_lengths=None, segment_size=4: (
        commons.slice_segments(x, ids, segment_size), ids)
    try:
        (y_hat, kl_ssl, ids_slice, x_mask, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), stats_ssl) = net_g(
            b["ssl"], spec, b["lengths"], b["text"], b["text_lengths"])
    finally:
        torch.randn_like, commons.rand_slice_segments = orig_randn_like, orig_rand_slice
    mel = spec_to_mel_torch(spec, 2048, 128, 32000, 0.0, None)
    y_mel = commons.slice_segments(mel, ids_slice, 32)
    y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1), 2048, 128, 32000, 640, 2048, 0.0, None)
    y = commons.slice_segments(b["wav"], ids_slice * 640, 20480)

    # D step (sovits.py:497-507)
    y_d_hat_r, y_d_hat_g, _, _ = net_d(y, y_hat.detach())
    loss_disc, _, _ = discriminator_loss(y_d_hat_r, y_d_hat_g)
    net_d.zero_grad()
    loss_disc.backward()
    d_grads = grad_stats(net_d)
    if hook_d is not None:    # round-2 fixtures: the discriminator's gradients as its optimiser sees them (sovits.py:504-507)
        hook_d(net_d)
    d_slices = {n: p.grad.flatten()[:64].clone() for n, p in net_d.named_parameters()
                if n in ("discriminators.0.convs.1.weight_v", "discriminators.1.convs.3.weight_g",
                         "discriminators.5.convs.0.weight_v", "discriminators.3.conv_post.bias",
                         "discriminators.2.convs.4.weight_v")}

    # G step (sovits.py:509-525), D weights as they are (no optimiser step in the fixture)
    y_d_hat_r, y_d_hat_g, fmap_r, fmap_g = net_d(y, y_hat)
    loss_mel = torch.nn.functional.l1_loss(y_mel, y_hat_mel) * 45
    loss_kl = kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * 1.0
    loss_fm = feature_loss(fmap_r, fmap_g)
    loss_gen, _ = generator_loss(y_d_hat_g)
    loss_gen_all = loss_gen + loss_fm + loss_mel + kl_ssl * 1 + loss_kl
    net_g.zero_grad()
    loss_gen_all.backward()
    g_grads = grad_stats(net_g)
    none_grads = sorted(n for n, p in net_g.named_parameters() if p.grad is None)
    g_slices = {n: p.grad.flatten()[:64].clone() for n, p in net_g.named_parameters()
                if n in ("dec.resblocks.0.convs1.0.weight_v", "dec.resblocks.14.convs2.2.weight_g", "dec.ups.0.weight_v",
                         "dec.ups.4.weight_g", "dec.conv_post.weight", "dec.conv_pre.bias", "enc_q.enc.in_layers.3.weight_v",
                         "flow.flows.2.enc.res_skip_layers.1.bias", "enc_p.encoder_text.attn_layers.2.emb_rel_k",
                         "enc_p.text_embedding.weight", "ref_enc.fc.fc.weight", "enc_p.mrte.c_post.weight",
                         "enc_q.pre.weight", "enc_p.encoder2.ffn_layers.1.conv_1.weight")}
    out = dict(
        config=dict(B=B, T=T, t_text=t_text),
        y_hat=y_hat.detach().squeeze(1),
        y_hat_mel=y_hat_mel.detach(),
        y_mel=y_mel.detach(),
        spec_head=spec[:, :, :4].clone(),
        losses=dict(disc=float(loss_disc), gen=float(loss_gen), fm=float(loss_fm), mel=float(loss_mel),
                    kl=float(loss_kl), kl_ssl=float(kl_ssl), gen_all=float(loss_gen_all)),
        d_logits=[t.detach().clone() for t in y_d_hat_g],
        stats=dict(z=z.detach()[:, :8, :16].clone(), z_p=z_p.detach()[:, :8, :16].clone(),
                   m_p=m_p.detach()[:, :8, :16].clone(), logs_p=logs_p.detach()[:, :8, :16].clone(),
                   m_q=m_q.detach()[:, :8, :16].clone(), logs_q=logs_q.detach()[:, :8, :16].clone(),
                   quantized=stats_ssl.detach()[:, :8, :16].clone()),
        d_grad_sumsq=d_grads, g_grad_sumsq=g_grads, g_none_grads=none_grads, d_grad_slices=d_slices,
        g_grad_slices=g_slices,
        fmap_shapes=[[tuple(t.shape) for t in f] for f in fmap_g],
    )
    if hook is not None:      # round-2 fixtures (make_golden_r2.py) extend the same run; s2_c1.pt itself is unchanged
        hook(net_g, net_d, out)
    if tag == "_noise":
        return out
    # fp32 noise floor of the REFERENCE itself: same computation on a different CPU conv backend / thread count
    # (different reduction order).  Tests accept max(2e-3, 3 x noise) on gradient slices.
    nt, mk = torch.get_num_threads(), torch.backends.mkldnn.enabled
    torch.set_num_threads(1)
    torch.backends.mkldnn.enabled = False
    try:
        alt = make_s2(B, T, t_text, tag="_noise")
    finally:
        torch.set_num_threads(nt)
        torch.backends.mkldnn.enabled = mk
    relerr = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
    out["g_grad_slice_noise"] = {n: relerr(alt["g_grad_slices"][n], v) for n, v in g_slices.items()}
    out["d_grad_slice_noise"] = {n: relerr(alt["d_grad_slices"][n], v) for n, v in d_slices.items()}
    out["g_grad_sumsq_noise"] = {k: abs(alt["g_grad_sumsq"][k] - v) / v for k, v in g_grads.items()}
    out["d_grad_sumsq_noise"] = {k: abs(alt["d_grad_sumsq"][k] - v) / v for k, v in d_grads.items()}
    path = os.path.join(HERE, f"s2_{tag}.pt")
    torch.save(out, path)
    print("wrote", path, {k: round(v, 5) for k, v in out["losses"].items()}, "none grads:", none_grads)


def make_keys():
    from src.easevoice.module import models
    from src.easevoice.soundstorm.auto_reg.models.t2s_model import Text2SemanticDecoder
    import yaml

    cfg = json.load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "s2.json")))
    g = models.SynthesizerTrn(1025, 32, n_speakers=300, **cfg["model"])
    d = models.MultiPeriodDiscriminator(False)
    s1cfg = yaml.safe_load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "gpt.yaml")))
    t = Text2SemanticDecoder(config=s1cfg, top_k=3)
    keys = dict(
        s2_g={k: list(v.shape) for k, v in g.state_dict().items()},
        s2_d={k: list(v.shape) for k, v in d.state_dict().items()},
        s1={k: list(v.shape) for k, v in t.state_dict().items() if not k.startswith("t2s_transformer")},
    )
    json.dump(keys, open(os.path.join(HERE, "state_dict_keys.json"), "w"))
    print("wrote keys", {k: len(v) for k, v in keys.items()})


def make_decode():
    """inference-side entry points of the s2 model: SynthesizerTrn.decode (models.py:974-1013) at
    speed 1 and 1.25 with one and with two reference spectrograms, and extract_latent (models.py:1015-1018).  The prior
    noise draw (randn_like) is replaced by an injected tensor."""
    from src.easevoice.module import models
    from util_fill import decode_inputs

    cfg = json.load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "s2.json")))
    net_g = models.SynthesizerTrn(1025, 32, n_speakers=300, **cfg["model"])
    fill_module(net_g, 1)
    net_g.eval()
    d = decode_inputs()
    out = dict(cases=[])
    orig = torch.randn_like
    try:
        for speed, refer in ((1, d["refers"][0]), (1, d["refers"]), (1.25, d["refers"][0])):
            torch.randn_like = lambda t, **kw: d["noise"][:, :, :t.size(2)].to(t.dtype)
            o = net_g.decode(d["codes"], d["text"], refer, noise_scale=0.5, speed=speed)
            out["cases"].append(dict(speed=speed, n_refer=len(refer) if isinstance(refer, list) else 1,
                                     shape=list(o.shape), o_head=o[0, 0, :4096].clone(), o_dec=o[0, 0, ::37].clone(),
                                     abs_sum=float(o.double().abs().sum()), sq_sum=float(o.double().pow(2).sum())))
            print("decode speed", speed, "->", tuple(o.shape), "rms", float(o.pow(2).mean().sqrt()))
    finally:
        torch.randn_like = orig
    out["codes"] = net_g.extract_latent(d["ssl"])
    print("extract_latent ->", tuple(out["codes"].shape))
    torch.save(out, os.path.join(HERE, "s2_decode.pt"))


if __name__ == "__main__":
    what = sys.argv[1:] or ["keys", "s2"]
    if "keys" in what:
        make_keys()
    if "s2" in what:
        make_s2()
    if "decode" in what:
        make_decode()
    if "s1" in what:
        sys.path.insert(0, HERE)
        from make_golden_s1 import make_s1
        make_s1()
