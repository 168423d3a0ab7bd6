"""TEST INFRASTRUCTURE ONLY — CPU fp32 restatement of the reference's s2 (SoVITS) training step.

A functional restatement (plain torch CPU ops over a state_dict, the reference's own [B, C, T] layout) of
  SynthesizerTrn.forward            src/easevoice/module/models.py:904-946
  TextEncoder / Encoder / MHA / FFN  models.py:228-251, attentions.py:68-90,233-292,408-416
  MRTE                               mrte_model.py:25-61
  MelStyleEncoder                    modules.py:739-763
  PosteriorEncoder / WN              models.py:348-359, modules.py:187-212
  ResidualCouplingBlock / Layer      models.py:308-315, modules.py:439-458
  Generator / ResBlock1              models.py:452-471, modules.py:298-311
  MultiPeriodDiscriminator / S / P   models.py:538-614
  losses, mel                        losses.py:7-61, mel_processing.py:40-142
  the step body                      src/train/sovits.py:459-525
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; it is the checker, never the
product.  PINNED: tests/test_oracle_cpu.py checks it against tests/golden/s2_c1.pt, which was produced by running the
reference's own modules (tests/golden/make_golden.py).  The mel filterbank (oracle/melbank.py) restates librosa 0.9.2,
which is absent here: that one boundary is unpinned.
"""
import math

import torch
import torch.nn.functional as F

from .melbank import slaney_mel
from .ops import weight_norm_fold

LRELU = 0.1


class SD:
    """prefix view over a flat state_dict"""

    def __init__(self, sd, prefix=""):
        self.sd, self.p = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.p + k]

    def has(self, k):
        return (self.p + k) in self.sd

    def sub(self, k):
        return SD(self.sd, self.p + k + ".")

    def w(self, name):
        """conv weight of module `name`: plain `weight` or weight-norm folded g*v/|v|"""
        if self.has(name + ".weight"):
            return self[name + ".weight"]
        return weight_norm_fold(self[name + ".weight_v"], self[name + ".weight_g"])

    def b(self, name):
        return self[name + ".bias"] if self.has(name + ".bias") else None


def seq_mask(lengths, T):
    return (torch.arange(T)[None, :] < lengths[:, None]).unsqueeze(1).float()     # [B, 1, T]


def layer_norm_c(x, gamma, beta, eps=1e-5):
    return F.layer_norm(x.transpose(1, -1), (x.size(1),), gamma, beta, eps).transpose(1, -1)


# ---- relative-position attention (attentions.py:243-365) ---------------------------------------------------------
def _rel_emb(emb, length, window=4):
    pad = max(length - (window + 1), 0)
    start = max((window + 1) - length, 0)
    if pad > 0:
        emb = F.pad(emb, (0, 0, pad, pad))
    return emb[:, start:start + 2 * length - 1]


def _rel2abs(x):
    b, h, l, _ = x.shape
    x = F.pad(x, (0, 1)).view(b, h, l * 2 * l)
    x = F.pad(x, (0, l - 1)).view(b, h, l + 1, 2 * l - 1)
    return x[:, :, :l, l - 1:]


def _abs2rel(x):
    b, h, l, _ = x.shape
    x = F.pad(x, (0, l - 1)).view(b, h, l * l + l * (l - 1))
    x = F.pad(x, (l, 0)).view(b, h, l, 2 * l)
    return x[:, :, :, 1:]


def mha(s: SD, x, c, attn_mask, n_heads, window=None):
    q = F.conv1d(x, s["conv_q.weight"], s["conv_q.bias"])
    k = F.conv1d(c, s["conv_k.weight"], s["conv_k.bias"])
    v = F.conv1d(c, s["conv_v.weight"], s["conv_v.bias"])
    b, d, t_s = k.shape
    t_t = q.size(2)
    kc = d // n_heads
    q = q.view(b, n_heads, kc, t_t).transpose(2, 3)
    k = k.view(b, n_heads, kc, t_s).transpose(2, 3)
    v = v.view(b, n_heads, kc, t_s).transpose(2, 3)
    scores = torch.matmul(q / math.sqrt(kc), k.transpose(-2, -1))
    if window is not None:
        ke = _rel_emb(s["emb_rel_k"], t_s, window)
        scores = scores + _rel2abs(torch.matmul(q / math.sqrt(kc), ke.unsqueeze(0).transpose(-2, -1)))
    if attn_mask is not None:
        scores = scores.masked_fill(attn_mask == 0, -1e4)
    p = F.softmax(scores, dim=-1)
    out = torch.matmul(p, v)
    if window is not None:
        ve = _rel_emb(s["emb_rel_v"], t_s, window)
        out = out + torch.matmul(_abs2rel(p), ve.unsqueeze(0))
    out = out.transpose(2, 3).contiguous().view(b, d, t_t)
    return F.conv1d(out, s["conv_o.weight"], s["conv_o.bias"])


def encoder(s: SD, x, x_mask, n_layers, n_heads=2, k=3):
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    x = x * x_mask
    pl, pr = (k - 1) // 2, k // 2
    for i in range(n_layers):
        y = mha(s.sub(f"attn_layers.{i}"), x, x, attn_mask, n_heads, window=4)
        x = layer_norm_c(x + y, s[f"norm_layers_1.{i}.gamma"], s[f"norm_layers_1.{i}.beta"])
        f = s.sub(f"ffn_layers.{i}")
        y = F.conv1d(F.pad(x * x_mask, (pl, pr)), f["conv_1.weight"], f["conv_1.bias"])
        y = torch.relu(y)
        y = F.conv1d(F.pad(y * x_mask, (pl, pr)), f["conv_2.weight"], f["conv_2.bias"]) * x_mask
        x = layer_norm_c(x + y, s[f"norm_layers_2.{i}.gamma"], s[f"norm_layers_2.{i}.beta"])
    return x * x_mask


def text_encoder(s: SD, y, y_mask, text, text_mask, ge, n_layers=6):
    y = F.conv1d(y * y_mask, s["ssl_proj.weight"], s["ssl_proj.bias"]) * y_mask
    y = encoder(s.sub("encoder_ssl"), y * y_mask, y_mask, n_layers // 2)
    t = F.embedding(text, s["text_embedding.weight"]).transpose(1, 2)
    t = encoder(s.sub("encoder_text"), t * text_mask, text_mask, n_layers)
    m = s.sub("mrte")
    attn_mask = text_mask.unsqueeze(2) * y_mask.unsqueeze(-1)
    ssl_enc = F.conv1d(y * y_mask, m["c_pre.weight"], m["c_pre.bias"])
    text_enc = F.conv1d(t * text_mask, m["text_pre.weight"], m["text_pre.bias"])
    x = mha(m.sub("cross_attention"), ssl_enc * y_mask, text_enc * text_mask, attn_mask, 4) + ssl_enc + ge
    y = F.conv1d(x * y_mask, m["c_post.weight"], m["c_post.bias"])
    y = encoder(s.sub("encoder2"), y * y_mask, y_mask, n_layers // 2)
    stats = F.conv1d(y, s["proj.weight"], s["proj.bias"]) * y_mask
    m_p, logs_p = torch.split(stats, stats.size(1) // 2, dim=1)
    return m_p, logs_p


def mel_style_encoder(s: SD, x, mask):
    """x [B, 704, T], mask [B, 1, T] -> [B, 512, 1]"""
    x = x.transpose(1, 2)
    pad = (mask.int() == 0).squeeze(1)
    mish = lambda v: v * torch.tanh(F.softplus(v))
    x = mish(F.linear(x, s["spectral.0.fc.weight"], s["spectral.0.fc.bias"]))
    x = mish(F.linear(x, s["spectral.3.fc.weight"], s["spectral.3.fc.bias"]))
    x = x.transpose(1, 2)
    for i in range(2):
        h = F.conv1d(x, s[f"temporal.{i}.conv1.conv.weight"], s[f"temporal.{i}.conv1.conv.bias"], padding=2)
        a, b = torch.split(h, h.size(1) // 2, dim=1)
        x = x + a * torch.sigmoid(b)
    x = x.transpose(1, 2).masked_fill(pad.unsqueeze(-1), 0)
    bsz, t, dm = x.shape
    nh, dk = 2, dm // 2
    a = s.sub("slf_attn")
    q = F.linear(x, a["w_qs.weight"], a["w_qs.bias"]).view(bsz, t, nh, dk).permute(2, 0, 1, 3).reshape(-1, t, dk)
    k = F.linear(x, a["w_ks.weight"], a["w_ks.bias"]).view(bsz, t, nh, dk).permute(2, 0, 1, 3).reshape(-1, t, dk)
    v = F.linear(x, a["w_vs.weight"], a["w_vs.bias"]).view(bsz, t, nh, dk).permute(2, 0, 1, 3).reshape(-1, t, dk)
    attn = torch.bmm(q, k.transpose(1, 2)) / (dm ** 0.5)
    attn = attn.masked_fill(pad.unsqueeze(1).expand(-1, t, -1).repeat(nh, 1, 1), -float("inf"))
    out = torch.bmm(F.softmax(attn, dim=2), v).view(nh, bsz, t, dk).permute(1, 2, 0, 3).reshape(bsz, t, -1)
    x = F.linear(out, a["fc.weight"], a["fc.bias"]) + x
    x = F.linear(x, s["fc.fc.weight"], s["fc.fc.bias"])
    n = (~pad).sum(dim=1).unsqueeze(1)
    return (x.masked_fill(pad.unsqueeze(-1), 0).sum(dim=1) / n).unsqueeze(-1)


def wn(s: SD, x, x_mask, g, n_layers, hidden=192, k=5):
    out = torch.zeros_like(x)
    g = F.conv1d(g, s.w("cond_layer"), s.b("cond_layer"))
    for i in range(n_layers):
        x_in = F.conv1d(x, s.w(f"in_layers.{i}"), s.b(f"in_layers.{i}"), padding=(k - 1) // 2)
        a = x_in + g[:, i * 2 * hidden:(i + 1) * 2 * hidden]
        acts = torch.tanh(a[:, :hidden]) * torch.sigmoid(a[:, hidden:])
        rs = F.conv1d(acts, s.w(f"res_skip_layers.{i}"), s.b(f"res_skip_layers.{i}"))
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out * x_mask


def posterior_encoder(s: SD, spec, mask, ge, eps):
    x = F.conv1d(spec, s["pre.weight"], s["pre.bias"]) * mask
    x = wn(s.sub("enc"), x, mask, ge.detach(), 16)
    stats = F.conv1d(x, s["proj.weight"], s["proj.bias"]) * mask
    m, logs = torch.split(stats, stats.size(1) // 2, dim=1)
    return (m + eps * torch.exp(logs)) * mask, m, logs


def flow(s: SD, x, mask, ge):
    for i in range(0, 8, 2):
        f = s.sub(f"flows.{i}")
        x0, x1 = torch.split(x, x.size(1) // 2, dim=1)
        h = F.conv1d(x0, f["pre.weight"], f["pre.bias"]) * mask
        h = wn(f.sub("enc"), h, mask, ge, 4)
        m = F.conv1d(h, f["post.weight"], f["post.bias"]) * mask
        x = torch.cat([x0, m + x1 * mask], dim=1)
        x = torch.flip(x, [1])
    return x


def generator(s: SD, x, ge, hps_model):
    x = F.conv1d(x, s["conv_pre.weight"], s["conv_pre.bias"], padding=3)
    x = x + F.conv1d(ge, s["cond.weight"], s["cond.bias"])
    ks, ds = hps_model["resblock_kernel_sizes"], hps_model["resblock_dilation_sizes"]
    nk = len(ks)
    for i, (u, k) in enumerate(zip(hps_model["upsample_rates"], hps_model["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU)
        x = F.conv_transpose1d(x, s.w(f"ups.{i}"), s.b(f"ups.{i}"), stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            r = s.sub(f"resblocks.{i * nk + j}")
            y = x
            for c, d in enumerate(ds[j]):
                xt = F.conv1d(F.leaky_relu(y, LRELU), r.w(f"convs1.{c}"), r.b(f"convs1.{c}"),
                              padding=(ks[j] * d - d) // 2, dilation=d)
                xt = F.conv1d(F.leaky_relu(xt, LRELU), r.w(f"convs2.{c}"), r.b(f"convs2.{c}"), padding=(ks[j] - 1) // 2)
                y = xt + y
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x)          # default slope 0.01, models.py:467
    return torch.tanh(F.conv1d(x, s["conv_post.weight"], None, padding=3))


def quantize(sd: SD, ssl):
    h = F.conv1d(ssl, sd["ssl_proj.weight"], sd["ssl_proj.bias"], stride=2)
    e = sd["quantizer.vq.layers.0._codebook.embed"]
    b, d, n = h.shape
    xf = h.transpose(1, 2).reshape(-1, d)
    dist = -(xf.pow(2).sum(1, keepdim=True) - 2 * xf @ e.t() + e.t().pow(2).sum(0, keepdim=True))
    q = F.embedding(dist.max(dim=-1).indices, e).view(b, n, d).transpose(1, 2)
    return F.interpolate(q, size=int(n * 2), mode="nearest").detach()


def synthesizer_forward(sd_g, hps, ssl, spec, lengths, text, text_lengths, eps, ids_slice):
    s = SD(sd_g)
    T = spec.size(2)
    y_mask = seq_mask(lengths, T)
    text_mask = seq_mask(text_lengths, text.size(1))
    ge = mel_style_encoder(s.sub("ref_enc"), spec[:, :704] * y_mask, y_mask)
    quantized = quantize(s, ssl)
    m_p, logs_p = text_encoder(s.sub("enc_p"), quantized, y_mask, text, text_mask, ge, hps["model"]["n_layers"])
    z, m_q, logs_q = posterior_encoder(s.sub("enc_q"), spec, y_mask, ge, eps)
    z_p = flow(s.sub("flow"), z, y_mask, ge)
    seg = hps["train"]["segment_size"] // hps["data"]["hop_length"]
    z_slice = torch.stack([z[i, :, ids_slice[i]:ids_slice[i] + seg] for i in range(z.size(0))])
    o = generator(s.sub("dec"), z_slice, ge, hps["model"])
    return o, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q), quantized


def disc_s(s: SD, x):
    spec = [(1, 7, 1), (4, 20, 4), (4, 20, 16), (4, 20, 64), (4, 20, 256), (1, 2, 1)]
    fmap = []
    for i, (st, pad, g) in enumerate(spec):
        x = F.leaky_relu(F.conv1d(x, s.w(f"convs.{i}"), s.b(f"convs.{i}"), stride=st, padding=pad, groups=g), LRELU)
        fmap.append(x)
    x = F.conv1d(x, s.w("conv_post"), s.b("conv_post"), padding=1)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def disc_p(s: SD, x, period):
    b, c, t = x.shape
    if t % period != 0:
        x = F.pad(x, (0, period - (t % period)), "reflect")
        t = x.size(2)
    x = x.view(b, c, t // period, period)
    fmap = []
    for i in range(5):
        x = F.leaky_relu(F.conv2d(x, s.w(f"convs.{i}"), s.b(f"convs.{i}"), stride=(3 if i < 4 else 1, 1),
                                  padding=(2, 0)), LRELU)
        fmap.append(x)
    x = F.conv2d(x, s.w("conv_post"), s.b("conv_post"), padding=(1, 0))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def mpd(sd_d, y):
    s = SD(sd_d)
    outs, fmaps = [], []
    o, f = disc_s(s.sub("discriminators.0"), y)
    outs.append(o); fmaps.append(f)
    for i, p in enumerate([2, 3, 5, 7, 11]):
        o, f = disc_p(s.sub(f"discriminators.{i + 1}"), y, p)
        outs.append(o); fmaps.append(f)
    return outs, fmaps


def stft_mag(y, n_fft=2048, hop=640):
    p = (n_fft - hop) // 2
    yp = F.pad(y.unsqueeze(1), (p, p), mode="reflect").squeeze(1)
    st = torch.stft(yp, n_fft, hop_length=hop, win_length=n_fft, window=torch.hann_window(n_fft), center=False,
                    pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    return torch.sqrt(st.real.pow(2) + st.imag.pow(2) + 1e-6)


_MEL = {}


def mel_from_spec(spec, hps):
    d = hps["data"]
    key = (d["sampling_rate"], d["filter_length"], d["n_mel_channels"])
    if key not in _MEL:
        _MEL[key] = torch.from_numpy(slaney_mel(d["sampling_rate"], d["filter_length"], d["n_mel_channels"],
                                                d["mel_fmin"], d["mel_fmax"]))
    return torch.log(torch.clamp(torch.matmul(_MEL[key], spec), min=1e-5))


def s2_losses(sd_g, sd_d, hps, ssl, wav, text, lengths, text_lengths, eps, ids_slice, with_grads=False):
    """The loss side of one step (sovits.py:459-518) with the two random draws injected.  Returns a dict of python
    floats / tensors; with_grads=True also returns d(loss_disc)/d(D params) and d(loss_gen_all)/d(G params)."""
    if with_grads:
        sd_g = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "quantizer" not in k else v)
                for k, v in sd_g.items()}
        sd_d = {k: v.clone().requires_grad_(True) for k, v in sd_d.items()}
    d, t = hps["data"], hps["train"]
    hop, segsz = d["hop_length"], t["segment_size"]
    spec = stft_mag(wav.squeeze(1), d["filter_length"], hop)
    y_hat, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q), _ = synthesizer_forward(
        sd_g, hps, ssl, spec, lengths, text, text_lengths, eps, ids_slice)
    mel = mel_from_spec(spec, hps)
    seg = segsz // hop
    y_mel = torch.stack([mel[i, :, ids_slice[i]:ids_slice[i] + seg] for i in range(mel.size(0))])
    y_hat_mel = mel_from_spec(stft_mag(y_hat.squeeze(1), d["filter_length"], hop), hps)
    y = torch.stack([wav[i, :, ids_slice[i] * hop: ids_slice[i] * hop + segsz] for i in range(wav.size(0))])
    dr, _ = mpd(sd_d, y)
    dg, _ = mpd(sd_d, y_hat.detach())
    loss_disc = sum(torch.mean((1 - a) ** 2) + torch.mean(b ** 2) for a, b in zip(dr, dg))
    _, fmap_r = mpd(sd_d, y)
    dg2, fmap_g = mpd(sd_d, y_hat)
    loss_mel = F.l1_loss(y_mel, y_hat_mel) * t["c_mel"]
    kl = logs_p - logs_q - 0.5 + 0.5 * ((z_p - m_p) ** 2) * torch.exp(-2.0 * logs_p)
    loss_kl = torch.sum(kl * y_mask) / torch.sum(y_mask) * t["c_kl"]
    loss_fm = 2 * sum(torch.mean(torch.abs(a.detach() - b)) for fr, fg in zip(fmap_r, fmap_g) for a, b in zip(fr, fg))
    loss_gen = sum(torch.mean((1 - a) ** 2) for a in dg2)
    loss_gen_all = loss_gen + loss_fm + loss_mel + loss_kl
    out = dict(disc=loss_disc.detach(), gen=loss_gen.detach(), fm=loss_fm.detach(), mel=loss_mel.detach(),
               kl=loss_kl.detach(), gen_all=loss_gen_all.detach(), y_hat=y_hat.detach(), y_hat_mel=y_hat_mel.detach(),
               spec=spec)
    if with_grads:
        dk = [k for k, v in sd_d.items() if v.requires_grad]
        gd = torch.autograd.grad(loss_disc, [sd_d[k] for k in dk], allow_unused=True)
        gk = [k for k, v in sd_g.items() if torch.is_tensor(v) and v.requires_grad]
        gg = torch.autograd.grad(loss_gen_all, [sd_g[k] for k in gk], allow_unused=True)
        out["d_grads"] = dict(zip(dk, gd))
        out["g_grads"] = dict(zip(gk, gg))
    return out


def adamw_step(p, g, m, v, step, lr, betas=(0.8, 0.99), eps=1e-9, wd=0.01):
    """torch.optim.AdamW single-tensor update (sovits.py:294-319 defaults), in place."""
    p.mul_(1 - lr * wd)
    m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
    v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
    bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
