"""TEST-ONLY shim: lets the product's module WIRING (layouts, masks, attention, flows, slicing, loss
composition) run on CPU by substituting oracle ops for the HIP launches.  It never ships and is never
imported by the package; GPU tests exercise the real kernels.  Usage: `with cpu_emulation(): ...`."""
import contextlib

import torch
import torch.nn.functional as F

from oracle import ops as O


def _w(m):
    if m.weight_norm:
        v = m.weight_v.squeeze(-1) if m.kdims == 2 else m.weight_v
        g = m.weight_g.squeeze(-1) if m.kdims == 2 else m.weight_g
        return O.weight_norm_fold(v, g)
    if m.kdims == 0:
        return m.weight.unsqueeze(-1)             # nn.Linear layout [cout, cin]
    return m.weight.squeeze(-1) if m.kdims == 2 else m.weight


def _attn_core(q, k, v, lens_q, lens_k, n_heads, scale, emb_k=None, emb_v=None, window=None):
    """MultiHeadAttention.attention (attentions.py:243-292) / ScaledDotProductAttention (modules.py:664-682) on
    channels-last rows, through the oracle's relative-position helpers; padded keys excluded, padded query rows zero"""
    from oracle import s2_step as OS

    b, tq, c = q.shape
    tk, d = k.size(1), c // n_heads
    qh = q.float().view(b, tq, n_heads, d).transpose(1, 2) * scale
    kh = k.float().view(b, tk, n_heads, d).transpose(1, 2)
    vh = v.float().view(b, tk, n_heads, d).transpose(1, 2)
    scores = torch.matmul(qh, kh.transpose(-2, -1))
    if window is not None:
        ke = OS._rel_emb(emb_k, tk, window)
        scores = scores + OS._rel2abs(torch.matmul(qh, ke.unsqueeze(0).transpose(-2, -1)))
    live_k = torch.arange(tk)[None, :] < (lens_k if lens_k is not None else torch.full((b,), tk))[:, None]
    live_q = torch.arange(tq)[None, :] < (lens_q if lens_q is not None else torch.full((b,), tq))[:, None]
    scores = scores.masked_fill(~live_k[:, None, None, :], float("-inf"))
    p = F.softmax(scores, dim=-1)
    out = torch.matmul(p, vh)
    if window is not None:
        ve = OS._rel_emb(emb_v, tk, window)
        out = out + torch.matmul(OS._abs2rel(p), ve.unsqueeze(0))
    out = out.transpose(1, 2).reshape(b, tq, c)
    return (out * live_q.unsqueeze(-1)).to(q.dtype)


def _conv_forward(self, x, res=None, in_slope=1.0, out_act=0, out_slope=1.0):
    if getattr(self, "src_d1", 0):
        x = x[..., :self.src_d1]                  # the zero-padded tail of a PaddedInPointwise input
    y = O.conv_block(x.transpose(1, 2), _w(self), self.bias, res.transpose(1, 2) if res is not None else None,
                     stride=self.stride, pad=self.pad, dil=self.dil, groups=self.groups, transposed=self.transposed,
                     in_slope=in_slope, out_act=out_act, out_slope=out_slope)
    return y.transpose(1, 2).contiguous()


@contextlib.contextmanager
def cpu_emulation():
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.module import attentions as PA, losses as PL, mel_processing as PM, models as PMod
    from easevoice_trainer_amd.train import s2_engine as PE

    saved_enc = (PA.res_drop_ln, PE.bump_rng)
    saved_attn = (PA.rel_self_attention, PA.mha_core, PMod.rel_self_attention, PMod.ncl_to_nlc, PMod.rvq_encode,
                  PM.spec_to_mel_torch, PM.spec_to_mel_slices, PE.spec_to_mel_slices, PA.PointwiseEvtConv.forward)

    def rel_self_attention(x, conv_q, conv_k, conv_v, emb_k, emb_v, lens, n_heads, window, p, site, packed=None,
                           scale=None):
        assert p == 0.0, "the CPU wiring emulation has no dropout stream"
        q, k, v = (_conv_forward(c, x) for c in (conv_q, conv_k, conv_v))
        scale = (conv_q.cout // n_heads) ** -0.5 if scale is None else scale
        return _attn_core(q, k, v, lens, lens, n_heads, scale, emb_k, emb_v, window)

    def mha_core(q, k, v, lens_q, lens_k, n_heads, p, site, scale, emb_k=None, emb_v=None, window=None):
        assert p == 0.0
        return _attn_core(q, k, v, lens_q, lens_k, n_heads, scale, emb_k, emb_v, window)

    def ncl_to_nlc(x, cpad=None, dtype=torch.float32):
        y = x.transpose(1, 2)
        if cpad is not None and cpad > y.size(-1):
            y = F.pad(y, (0, cpad - y.size(-1)))
        return y.to(dtype).contiguous()

    def rvq_encode(enc, quantizer, ssl, rep):
        """SynthesizerTrn's ssl_proj + EuclideanCodebook.quantize / dequantize (core_vq.py:172-190) in torch"""
        net, cb = enc                              # (the SynthesizerTrn, its codebook): see _rvq below
        w, bias = net.ssl_proj.weight.float(), net.ssl_proj.bias.float()
        h = F.conv1d(ssl.float(), w, bias, stride=w.size(-1)).transpose(1, 2)            # [B, T', D]
        quantizer.ensure_init(h)
        x = h.reshape(-1, h.size(-1))
        e = cb.embed.t()
        dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ e + e.pow(2).sum(0, keepdim=True))
        ind = dist.max(dim=-1).indices
        q = F.embedding(ind, cb.embed).view(h.shape)
        return q.repeat_interleave(rep, dim=1) if rep > 1 else q, ind.view(1, h.size(0), h.size(1))

    def pointwise_forward(self, x):
        lead = x.shape[:-1]
        y = _conv_forward(self, x.reshape(1, -1, x.size(-1)) if x.dim() != 3 else x)
        return y if x.dim() == 3 else y.reshape(*lead, self.cout)

    saved_rvq = PMod.SynthesizerTrn._rvq
    PMod.SynthesizerTrn._rvq = lambda self: (self, self.quantizer.vq.layers[0]._codebook)
    PA.rel_self_attention, PA.mha_core, PMod.rel_self_attention = rel_self_attention, mha_core, rel_self_attention
    PMod.ncl_to_nlc, PMod.rvq_encode = ncl_to_nlc, rvq_encode
    PA.PointwiseEvtConv.forward = pointwise_forward

    def res_drop_ln(x, y, gamma, beta, lens, p, site, eps=1e-5):
        assert p == 0.0, "the CPU wiring emulation has no dropout stream"
        live = (torch.arange(x.size(1))[None, :] < lens[:, None]).unsqueeze(-1).to(x.dtype)
        return F.layer_norm(x + y, (x.size(-1),), gamma, beta, eps) * live

    def _live(t, lens):
        return (torch.arange(t.size(1))[None, :] < lens[:, None]).unsqueeze(-1).to(t.dtype)

    def wn_residual(x, rs, acc, lens):
        H = rs.size(-1) // 2
        skip = rs[..., H:]
        return (x + rs[..., :H]) * _live(x, lens), (skip if acc is None else acc + skip)

    def wn_residual_last(rs, acc, lens):
        return (rs if acc is None else acc + rs) * _live(rs, lens)

    saved_wn = (PMod.wn_residual, PMod.wn_residual_last)
    PMod.wn_residual, PMod.wn_residual_last = wn_residual, wn_residual_last
    PA.res_drop_ln, PE.bump_rng = res_drop_ln, (lambda device: None)

    saved = (HC.EvtConv1d.forward, PMod.res_unit, PMod.Add3ScaleFn, PMod.GatedActFn, PL.feature_loss,
             PL.discriminator_loss, PL.generator_loss, PM.mel_spectrogram_torch, PM.spectrogram_torch)

    class _Add3:
        @staticmethod
        def apply(a, b, c, scale, owner=None):
            s = a
            if b is not None:
                s = s + b
            if c is not None:
                s = s + c
            return s * scale

    class _Gated:
        @staticmethod
        def apply(xin, g):
            h = xin.size(-1) // 2
            a = xin + g.unsqueeze(1) if g is not None else xin
            return torch.tanh(a[..., :h]) * torch.sigmoid(a[..., h:])

    def res_unit(x, c1, c2, slope):
        mid = _conv_forward(c1, x, None, slope)
        return _conv_forward(c2, mid, x, slope)

    def feature_loss(fr, fg):
        loss = 0
        for dr, dg in zip(fr, fg):
            for rl, gl in zip(dr, dg):
                loss = loss + torch.mean(torch.abs(rl.float().detach() - gl.float()))
        return loss * 2

    def discriminator_loss(rs, gs):
        loss = 0
        for dr, dg in zip(rs, gs):
            loss = loss + torch.mean((1 - dr.float()) ** 2) + torch.mean(dg.float() ** 2)
        return loss

    def generator_loss(gs):
        loss = 0
        for dg in gs:
            loss = loss + torch.mean((1 - dg.float()) ** 2)
        return loss

    def _stft_mag(y, n_fft, hop):
        p = (n_fft - hop) // 2
        yp = F.pad(y.unsqueeze(1), (p, p), mode="reflect").squeeze(1)
        s = torch.stft(yp, n_fft, hop_length=hop, win_length=n_fft, window=torch.hann_window(n_fft), center=False,
                       onesided=True, return_complex=True)
        return torch.sqrt(s.real ** 2 + s.imag ** 2 + 1e-6)

    def mel_spectrogram_torch(y, n_fft, num_mels, sr, hop, win, fmin, fmax, center=False):
        basis = torch.from_numpy(PM.mel_filterbank(sr, n_fft, num_mels, fmin, fmax))
        return torch.log(torch.clamp(torch.matmul(basis, _stft_mag(y.float(), n_fft, hop)), min=1e-5))

    def spectrogram_torch(y, n_fft, sr, hop, win, center=False):
        return _stft_mag(y.float(), n_fft, hop)

    def spec_to_mel_torch(spec, n_fft, num_mels, sr, fmin, fmax):
        basis = torch.from_numpy(PM.mel_filterbank(sr, n_fft, num_mels, fmin, fmax))
        return torch.log(torch.clamp(torch.matmul(basis, spec.float()), min=1e-5))

    def spec_to_mel_slices(spec, ids_slice, nfr, n_fft, num_mels, sr, fmin, fmax):
        from easevoice_trainer_amd.module import commons
        mel = spec_to_mel_torch(spec, n_fft, num_mels, sr, fmin, fmax)
        return commons.slice_segments(mel.transpose(1, 2), ids_slice, nfr).transpose(1, 2)

    PM.spec_to_mel_torch, PM.spec_to_mel_slices, PE.spec_to_mel_slices = spec_to_mel_torch, spec_to_mel_slices, spec_to_mel_slices

    def kl_loss(z_p, logs_q, m_p, logs_p, z_mask, lens=None):
        z_p, logs_q, m_p, logs_p, z_mask = z_p.float(), logs_q.float(), m_p.float(), logs_p.float(), z_mask.float()
        kl = logs_p - logs_q - 0.5 + 0.5 * ((z_p - m_p) ** 2) * torch.exp(-2.0 * logs_p)
        return torch.sum(kl * z_mask) / torch.sum(z_mask)

    saved_kl = (PL.kl_loss, PE.kl_loss)
    PL.kl_loss = PE.kl_loss = kl_loss
    HC.EvtConv1d.forward = _conv_forward
    PMod.res_unit, PMod.Add3ScaleFn, PMod.GatedActFn = res_unit, _Add3, _Gated
    PL.feature_loss, PL.discriminator_loss, PL.generator_loss = feature_loss, discriminator_loss, generator_loss
    PM.mel_spectrogram_torch, PM.spectrogram_torch = mel_spectrogram_torch, spectrogram_torch
    try:
        yield
    finally:
        (HC.EvtConv1d.forward, PMod.res_unit, PMod.Add3ScaleFn, PMod.GatedActFn, PL.feature_loss,
         PL.discriminator_loss, PL.generator_loss, PM.mel_spectrogram_torch, PM.spectrogram_torch) = saved
        PA.res_drop_ln, PE.bump_rng = saved_enc
        (PA.rel_self_attention, PA.mha_core, PMod.rel_self_attention, PMod.ncl_to_nlc, PMod.rvq_encode,
         PM.spec_to_mel_torch, PM.spec_to_mel_slices, PE.spec_to_mel_slices, PA.PointwiseEvtConv.forward) = saved_attn
        PMod.SynthesizerTrn._rvq = saved_rvq
        PL.kl_loss, PE.kl_loss = saved_kl
        PMod.wn_residual, PMod.wn_residual_last = saved_wn


@contextlib.contextmanager
def cpu_emulation_s1():
    """substitutes the three s1 autograd Functions and ScaledAdam's two kernel calls by torch CPU equivalents"""
    from easevoice_trainer_amd.auto_reg import optim as OPT, t2s_model as TM
    from oracle import s1_step as OS

    saved = (TM.PrefixLMAttentionFn, TM.AddLayerNormFn, TM.CrossEntropySumFn, OPT.ScaledAdam._k_stats,
             OPT.ScaledAdam._k_apply)

    class _Attn:
        @staticmethod
        def apply(qkv, x_lens, y_lens, x_len, n_head, dropout_p, seed):
            mask = OS.prefix_lm_mask(x_lens.long(), y_lens.long(), x_len, qkv.size(1) - x_len)
            return OS.attention(qkv, mask, n_head)

    class _LN:
        @staticmethod
        def apply(x, r, gamma, beta, eps):
            return F.layer_norm(x + r if r is not None else x, (x.size(-1),), gamma, beta, eps)

    def lin(x, weight, bias=None, relu=False):
        y = F.linear(x, weight.to(x.dtype), bias.to(x.dtype) if bias is not None else None)
        return F.relu(y) if relu else y

    class _CE:
        @staticmethod
        def apply(logits, targets, topk, ignore_index, V=None):
            logits = logits if V is None else logits[:, :V]
            loss = F.cross_entropy(logits.float(), targets, reduction="sum")
            lt = logits.detach().gather(1, targets[:, None])
            gt = (logits.detach() > lt).sum(dim=1)
            keep = targets != ignore_index
            hits = torch.stack([((gt < topk) & keep).sum(), keep.sum()]).to(torch.int32)
            return loss, hits

    class _CERows:
        @staticmethod
        def apply(logits, targets, topk, ignore_index, V=None):
            logits = logits if V is None else logits[:, :V]
            row = F.cross_entropy(logits.float(), targets, reduction="none")
            return row, _CE.apply(logits, targets, topk, ignore_index)[1]

    def k_stats(self):
        a = self.arena
        for b, e, t in self._chunk_list:
            p, g = a.param[b:e], a.grad[b:e]
            self._stats[t, 0] += (p * g).sum()
            self._stats[t, 1] += (p * p).sum()
            self._stats[t, 2] += (g * g).sum()

    def k_apply(self, hp):
        a = self.arena
        bc2 = 1.0 - hp.beta2 ** (hp.step + 1)
        for b, e, t in self._chunk_list:
            p, g = a.param[b:e], a.grad[b:e]
            d, v = self.delta[b:e], self.exp_avg_sq[b:e]
            d.mul_(hp.beta1)
            v.mul_(hp.beta2).addcmul_(g, g, value=1 - hp.beta2)
            if float(self._coef[t, 2]) == 0.0:
                d.add_(p * self._coef[t, 0])
                vh = v / bc2 if bc2 < 0.99 else v
                d.add_(g / (vh.sqrt() + hp.eps) * self._coef[t, 1])
                p.add_(d)
            else:
                d.add_(g / ((v / bc2).sqrt() + hp.eps), alpha=-hp.lr * hp.scalar_lr_scale * (1 - hp.beta1))
                p.clamp_(min=-hp.scalar_max, max=hp.scalar_max)
                p.add_(d)

    TM.PrefixLMAttentionFn, TM.AddLayerNormFn, TM.CrossEntropySumFn = _Attn, _LN, _CE
    saved_lin, TM.linear = TM.linear, lin
    from easevoice_trainer_amd.hip import linear as HL
    saved_prep, HL.LinearBank.prepare = HL.LinearBank.prepare, (lambda self, force=False: None)   # no images on the CPU
    saved_rows, TM.CrossEntropyRowsFn = TM.CrossEntropyRowsFn, _CERows
    OPT.ScaledAdam._k_stats, OPT.ScaledAdam._k_apply = k_stats, k_apply
    try:
        yield
    finally:
        (TM.PrefixLMAttentionFn, TM.AddLayerNormFn, TM.CrossEntropySumFn, OPT.ScaledAdam._k_stats,
         OPT.ScaledAdam._k_apply) = saved
        TM.CrossEntropyRowsFn = saved_rows
        TM.linear = saved_lin
        HL.LinearBank.prepare = saved_prep


@contextlib.contextmanager
def cpu_emulation_decode():
    """the s1 decoding session (auto_reg/t2s_infer.py) with its HIP launches substituted by torch CPU arithmetic on the
    session's own buffers and counters: pins the host side -- prompt pass, counters, stop polling, per-row cuts, batch
    groups -- against the reference's token sequences without a GPU.  Graph capture is off (EVT_DECODE_GRAPH=0)."""
    import os
    from easevoice_trainer_amd.auto_reg import t2s_infer as TI
    from oracle import s1_step as OS

    DS = TI.DecodeSession
    saved = (DS._gemv, DS._sample_embed_advance, DS.step_launches, TI.PrefixLMAttentionFn, TI.AddLayerNormFn,
             os.environ.get("EVT_DECODE_GRAPH"))

    class _Attn:
        @staticmethod
        def apply(qkv, x_lens, y_lens, x_len, n_head, dropout_p, seed):
            mask = OS.prefix_lm_mask(x_lens.long(), y_lens.long(), x_len, qkv.size(1) - x_len)
            return OS.attention(qkv, mask, n_head)

    class _LN:
        @staticmethod
        def apply(x, r, gamma, beta, eps):
            return F.layer_norm(x + r if r is not None else x, (x.size(-1),), gamma, beta, eps)

    def gemv(self, w, bias, a, r, g, b, eps, x_out, y, relu=0):
        x = a if r is None else F.layer_norm(a + r, (a.size(-1),), g, b, eps)
        if r is not None and x_out is not None:
            x_out.copy_(x)
        o = x @ w.float().t() + (bias if bias is not None else 0.0)
        y.copy_(o.clamp(min=0) if relu else o)

    def sample_embed_advance(self, W, sp, noise, pe, dpos):
        pos, idx, ycount, ylen = self.ctr[:4].tolist()
        Ve = sp.V - 1 if idx < sp.no_eos_steps else sp.V
        for b in range(self.B):
            lg = self.logits[b:b + 1, :Ve].clone()
            prev = self.y[b:b + 1, :ycount]
            if sp.repetition_penalty != 1.0 and ycount > 0:       # in place in the reference: the EOS test sees it
                sc = torch.gather(lg, 1, prev)
                lg.scatter_(1, prev, torch.where(sc < 0, sc * sp.repetition_penalty, sc / sp.repetition_penalty))
            probs = OS.logits_to_probs(lg, None, sp.temperature, sp.top_k if sp.top_k > 0 else None, sp.top_p, 1.0)
            q = noise[idx, b if sp.noise_rows > 1 else 0, :Ve] if noise.dim() == 3 else noise[idx, :Ve]
            tok = int(torch.argmax(probs / q, dim=-1)[0])
            self.y[b, ycount] = tok
            if (int(torch.argmax(lg, dim=-1)[0]) == sp.eos or tok == sp.eos) and int(self.stop[b]) < 0:
                self.stop[b] = idx
            self.xa[b] = W.emb[tok] * self.model.ar_audio_position.x_scale + W.alpha * pe[ylen + idx]
        self.ctr[0] += dpos
        self.ctr[1] += 1
        self.ctr[2] += 1

    def step_launches(self, W, sp, noise, pe, fused_qkv=False):
        pos = int(self.ctr[0])
        E, H = self.E, self.H
        d = E // H
        x = self.xa.clone()
        keep = torch.ones(self.B, pos + 1, dtype=torch.bool)
        if self.x_lens is not None:
            for b in range(self.B):
                keep[b, int(self.x_lens[b]):self.x_len] = False
        for i, w in enumerate(W.layers):
            qkv = x @ w["wqkv"].float().t() + w["bqkv"]
            self.kc[i, :, pos] = qkv[:, E:2 * E]
            self.vc[i, :, pos] = qkv[:, 2 * E:]
            K = self.kc[i, :, :pos + 1].float().view(self.B, pos + 1, H, d).transpose(1, 2)
            V = self.vc[i, :, :pos + 1].float().view(self.B, pos + 1, H, d).transpose(1, 2)
            q = qkv[:, :E].view(self.B, 1, H, d).transpose(1, 2)
            s = (q @ K.transpose(-1, -2) / d ** 0.5).masked_fill(~keep[:, None, None, :], float("-inf"))
            att = (F.softmax(s, -1) @ V).transpose(1, 2).reshape(self.B, E)
            t = att @ w["wo"].float().t() + w["bo"]
            x1 = F.layer_norm(x + t, (E,), w["g1"], w["be1"], w["eps1"])
            u = F.relu(x1 @ w["w1"].float().t() + w["b1"]) @ w["w2"].float().t() + w["b2"]
            x = F.layer_norm(x1 + u, (E,), w["g2"], w["be2"], w["eps2"])
        self.logits.copy_(x @ W.wpred.float().t())
        sample_embed_advance(self, W, sp, noise, pe, 1)

    def dense(self, dtype, device):
        bias = {id(w): b for _n, w, b in self.model.dense_specs()}

        def run(x, weight, relu=False):
            b = bias[id(weight)]
            o = x.float() @ weight.float().t() + (b.float() if b is not None else 0.0)
            return (o.clamp(min=0) if relu else o).to(x.dtype)

        return run

    saved_dense = TI.T2SInfer.dense
    TI.T2SInfer.dense = dense
    DS._gemv, DS._sample_embed_advance, DS.step_launches = gemv, sample_embed_advance, step_launches
    TI.PrefixLMAttentionFn, TI.AddLayerNormFn = _Attn, _LN
    os.environ["EVT_DECODE_GRAPH"] = "0"
    try:
        yield
    finally:
        TI.T2SInfer.dense = saved_dense
        DS._gemv, DS._sample_embed_advance, DS.step_launches, TI.PrefixLMAttentionFn, TI.AddLayerNormFn = saved[:5]
        if saved[5] is None:
            os.environ.pop("EVT_DECODE_GRAPH", None)
        else:
            os.environ["EVT_DECODE_GRAPH"] = saved[5]
