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
    return m.weight.squeeze(-1) if m.kdims == 2 else m.weight


def _conv_forward(self, x, res=None, in_slope=1.0, out_act=0, out_slope=1.0):
    y = O.conv_block(x.transpose(1, 2), _w(self), self.bias, res.transpose(1, 2) if res is not None else None,
                     stride=self.stride, pad=self.pad, dil=self.dil, groups=self.groups, transposed=self.transposed,
                     in_slope=in_slope, out_act=out_act, out_slope=out_slope)
    return y.transpose(1, 2).contiguous()


@contextlib.contextmanager
def cpu_emulation():
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.module import losses as PL, mel_processing as PM, models as PMod

    saved = (HC.EvtConv1d.forward, PMod.res_unit, PMod.Add3ScaleFn, PMod.GatedActFn, PL.feature_loss,
             PL.discriminator_loss, PL.generator_loss, PM.mel_spectrogram_torch, PM.spectrogram_torch)

    class _Add3:
        @staticmethod
        def apply(a, b, c, scale):
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

    HC.EvtConv1d.forward = _conv_forward
    PMod.res_unit, PMod.Add3ScaleFn, PMod.GatedActFn = res_unit, _Add3, _Gated
    PL.feature_loss, PL.discriminator_loss, PL.generator_loss = feature_loss, discriminator_loss, generator_loss
    PM.mel_spectrogram_torch, PM.spectrogram_torch = mel_spectrogram_torch, spectrogram_torch
    try:
        yield
    finally:
        (HC.EvtConv1d.forward, PMod.res_unit, PMod.Add3ScaleFn, PMod.GatedActFn, PL.feature_loss,
         PL.discriminator_loss, PL.generator_loss, PM.mel_spectrogram_torch, PM.spectrogram_torch) = saved
