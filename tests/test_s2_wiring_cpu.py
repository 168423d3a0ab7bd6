"""CPU: the product's s2 module WIRING vs fixtures generated from the reference's own modules.  The HIP launches
are substituted by oracle ops (tests/cpu_emu.py) — this pins layouts / masks / attention / flow / quantizer / loss
composition and the state_dict surface without a GPU; the kernels themselves are covered by the -m gpu tests."""
import json
import os

import torch

from cpu_emu import cpu_emulation
from util_fill import fill_module, s2_batch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def rel(a, b):
    return ((a.detach().float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12)).item()


def test_state_dict_surface():
    from easevoice_trainer_amd.module import models

    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    keys = json.load(open(os.path.join(HERE, "golden", "state_dict_keys.json")))
    g = models.SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
    d = models.MultiPeriodDiscriminator(False)
    assert {k: list(v.shape) for k, v in g.state_dict().items()} == keys["s2_g"]
    assert {k: list(v.shape) for k, v in d.state_dict().items()} == keys["s2_d"]
    assert sum(1 for k in keys["s2_g"] if "enc_q" in k) == 103     # dropped by the export, sovits.py:183-186


def test_s2_step_wiring_matches_reference():
    from easevoice_trainer_amd.module import commons, losses as PL, mel_processing as PM, models

    torch.set_num_threads(8)
    gold = torch.load(os.path.join(HERE, "golden", "s2_c1.pt"), weights_only=False)
    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    hps["model"]["p_dropout"] = 0.0
    with cpu_emulation():
        net_g = models.SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
        net_d = models.MultiPeriodDiscriminator(False)
        for m in net_g.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        fill_module(net_g, 1)
        fill_module(net_d, 2)
        c = gold["config"]
        b = s2_batch(c["B"], c["T"], c["t_text"])
        spec = PM.spectrogram_torch(b["wav"].squeeze(1), 2048, 32000, 640, 2048)
        assert rel(spec[:, :, :4], gold["spec_head"]) < 1e-4
        # autocast("cuda") is a no-op on CPU
        (y_hat, kl_ssl, ids, x_mask, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), q) = net_g(
            b["ssl"], spec, b["lengths"], b["text"], b["text_lengths"], eps=b["eps"], ids_slice=b["ids_slice"])
        got = dict(z=z, z_p=z_p, m_p=m_p, logs_p=logs_p, m_q=m_q, logs_q=logs_q, quantized=q)
        for k, v in gold["stats"].items():
            assert rel(got[k][:, :8, :16], v) < 1e-4, k
        assert rel(y_hat.squeeze(1), gold["y_hat"]) < 1e-4
        mel = PM.spec_to_mel_torch(spec, 2048, 128, 32000, 0.0, None)
        y_mel = commons.slice_segments(mel.transpose(1, 2), ids, 32).transpose(1, 2)
        y_hat_mel = PM.mel_spectrogram_torch(y_hat.squeeze(1), 2048, 128, 32000, 640, 2048, 0.0, None)
        assert rel(y_mel, gold["y_mel"]) < 1e-4 and rel(y_hat_mel, gold["y_hat_mel"]) < 1e-4
        y_seg = commons.slice_segments_1d(b["wav"].squeeze(1), ids * 640, 20480)
        rs, gs, _, _ = net_d(y_seg, y_hat.detach())
        loss_disc = PL.discriminator_loss(rs, gs)
        assert abs(float(loss_disc) - gold["losses"]["disc"]) < 1e-4 * gold["losses"]["disc"]
        _, fmap_r = net_d.forward_single(y_seg)
        dg, fmap_g = net_d.forward_single(y_hat)
        assert [[tuple(t.shape) for t in f] for f in fmap_g] is not None
        loss_fm = PL.feature_loss(fmap_r, fmap_g)
        loss_gen = PL.generator_loss(dg)
        loss_kl = PL.kl_loss(z_p, logs_q, m_p, logs_p, z_mask)
        loss_mel = torch.nn.functional.l1_loss(y_mel, y_hat_mel) * 45
        for name, val in (("fm", loss_fm), ("gen", loss_gen), ("kl", loss_kl), ("mel", loss_mel)):
            assert abs(float(val) - gold["losses"][name]) < 2e-4 * abs(gold["losses"][name]), (name, float(val))
        (loss_gen + loss_fm + loss_mel + loss_kl).backward()
        tot = {}
        for n, p in net_g.named_parameters():
            if p.grad is not None:
                tot[n.split(".")[0]] = tot.get(n.split(".")[0], 0.0) + float(p.grad.double().pow(2).sum())
        for k, v in gold["g_grad_sumsq"].items():
            assert abs(tot[k] - v) <= 1e-3 * v, (k, tot[k], v)
        for n, s in gold["g_grad_slices"].items():
            p = dict(net_g.named_parameters())[n]
            assert rel(p.grad.flatten()[:64], s) < max(2e-3, 3 * gold["g_grad_slice_noise"][n]), n


def test_decode_and_extract_latent_wiring():
    """inference entry points of the s2 model vs the reference's SynthesizerTrn.decode /
    extract_latent: one and two reference spectrograms, speed 1 and 1.25, injected prior noise"""
    from easevoice_trainer_amd.module import models
    from util_fill import decode_inputs

    torch.set_num_threads(8)
    gold = torch.load(os.path.join(HERE, "golden", "s2_decode.pt"), weights_only=False)
    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    d = decode_inputs()
    with cpu_emulation():
        net_g = models.SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
        fill_module(net_g, 1)
        net_g.eval()
        for c in gold["cases"]:
            refer = d["refers"] if c["n_refer"] == 2 else d["refers"][0]
            o = net_g.decode(d["codes"], d["text"], refer, noise_scale=0.5, speed=c["speed"], noise=d["noise"])
            assert list(o.shape) == c["shape"]
            assert rel(o[0, 0, :4096], c["o_head"]) < 1e-4 and rel(o[0, 0, ::37], c["o_dec"]) < 1e-4
            assert abs(float(o.double().pow(2).sum()) - c["sq_sum"]) < 1e-4 * c["sq_sum"]
        codes = net_g.extract_latent(d["ssl"])
        assert codes.dtype == torch.long and torch.equal(codes, gold["codes"])
