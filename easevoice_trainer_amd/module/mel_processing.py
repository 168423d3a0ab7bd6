"""Mel / spectrogram front-end of the s2 step, src/easevoice/module/mel_processing.py:40-142.

`mel_spectrogram_torch` (needs a backward: it is applied to the generated waveform) is the fused HIP
STFT->magnitude->mel->log kernel with an analytic backward; `spec_to_mel_torch` (target side, no grad) is the
LDS-tiled projection kernel of csrc/frontend.hip (`spec_to_mel_slices`: only the frames of the training segment);
`spectrogram_torch` (dataset side) reuses the HIP kernel's magnitude output."""
import numpy as np
import torch

from ..hip import lib as L
from ..hip.frontend import spec_to_mel

_mel_basis = {}
_hann_window = {}


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """Slaney-scale, slaney-normalised triangular filterbank == librosa.filters.mel(htk=False, norm="slaney")
    (librosa 0.9.2 is the reference's pinned dependency, uv.lock:1805-1806).  float32 [n_mels, n_fft//2+1]."""
    fmax = sr / 2.0 if fmax is None else fmax
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz2mel(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def mel2hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    freqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    edges = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    lower = -ramps[:-2] / width[:-1, None]
    upper = ramps[2:] / width[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return w.astype(np.float32)


def _basis(n_fft, num_mels, sampling_rate, fmin, fmax, device):
    key = (n_fft, num_mels, sampling_rate, fmin, fmax, str(device))
    if key not in _mel_basis:
        _mel_basis[key] = torch.from_numpy(mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)).to(device)
    return _mel_basis[key]


def _window(win_size, device):
    key = (win_size, str(device))
    if key not in _hann_window:
        _hann_window[key] = torch.hann_window(win_size, device=device, dtype=torch.float32)
    return _hann_window[key]


class _MelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wav, window, basis, n_fft, hop, want_spec):
        nseq, wav_len = wav.shape
        n_mels = basis.size(0)
        pad = (n_fft - hop) // 2
        frames = (wav_len + 2 * pad - n_fft) // hop + 1
        lib = L.lib()
        nws = lib.evt_mel_workspace_floats(nseq, wav_len, n_fft, hop, n_mels)
        ws = torch.empty(nws, dtype=torch.float32, device=wav.device)
        mel = torch.empty((nseq, n_mels, frames), dtype=torch.float32, device=wav.device)
        spec = torch.empty((nseq, n_fft // 2 + 1, frames), dtype=torch.float32, device=wav.device) if want_spec else None
        L.check(lib.evt_mel_fwd(L.ptr(wav), L.ptr(window), L.ptr(basis), L.ptr(spec), L.ptr(mel), L.ptr(ws), nseq,
                                wav_len, n_fft, hop, n_mels, L.stream_ptr()), "evt_mel_fwd")
        ctx.save_for_backward(window, basis, ws)
        ctx.dims = (nseq, wav_len, n_fft, hop, n_mels)
        ctx.mark_non_differentiable(*( [spec] if spec is not None else []))
        return (mel, spec) if want_spec else mel

    @staticmethod
    def backward(ctx, dmel, *_):
        window, basis, ws = ctx.saved_tensors
        nseq, wav_len, n_fft, hop, n_mels = ctx.dims
        dwav = torch.empty((nseq, wav_len), dtype=torch.float32, device=dmel.device)
        L.check(L.lib().evt_mel_bwd(L.ptr(dmel.float().contiguous()), L.ptr(window), L.ptr(basis), L.ptr(ws),
                                    L.ptr(dwav), nseq, wav_len, n_fft, hop, n_mels, L.stream_ptr()), "evt_mel_bwd")
        return dwav, None, None, None, None, None


def mel_spectrogram_torch(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    """y [B, T] waveform -> log-mel [B, num_mels, frames] (mel_processing.py:93-142)."""
    if win_size != n_fft or center:
        raise L.EvtError("only win_length == n_fft, center=False (configs/s2.json) is implemented")
    y = y.float().contiguous()
    return _MelFn.apply(y, _window(win_size, y.device), _basis(n_fft, num_mels, sampling_rate, fmin, fmax, y.device),
                        n_fft, hop_size, False)


def spectrogram_torch(y, n_fft, sampling_rate, hop_size, win_size, center=False):
    """y [B, T] -> linear magnitude spectrogram [B, n_fft//2+1, frames] (mel_processing.py:40-74)."""
    if win_size != n_fft or center:
        raise L.EvtError("only win_length == n_fft, center=False is implemented")
    y = y.float().contiguous()
    with torch.no_grad():
        _, spec = _MelFn.apply(y, _window(win_size, y.device), _basis(n_fft, 128, sampling_rate, 0.0, None, y.device),
                               n_fft, hop_size, True)
    return spec


def spec_to_mel_torch(spec, n_fft, num_mels, sampling_rate, fmin, fmax):
    """[B, n_fft//2+1, T] -> log-mel [B, num_mels, T] (mel_processing.py:77-90); no grad path."""
    with torch.no_grad():
        return spec_to_mel(spec, _basis(n_fft, num_mels, sampling_rate, fmin, fmax, spec.device))


def spec_to_mel_slices(spec, ids_slice, segment_frames, n_fft, num_mels, sampling_rate, fmin, fmax):
    """the mel of the training segments only: spec_to_mel_torch followed by commons.slice_segments(mel, ids_slice,
    segment_frames) (sovits.py:470-480) in one launch -> [B, num_mels, segment_frames]"""
    with torch.no_grad():
        return spec_to_mel(spec, _basis(n_fft, num_mels, sampling_rate, fmin, fmax, spec.device), ids_slice,
                           segment_frames)
