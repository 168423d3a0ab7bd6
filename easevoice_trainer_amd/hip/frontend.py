"""Host side of the s2 input front-end (csrc/frontend.hip): the layout change from the reference's [B, C, T] tensors to
channels-last rows, the frozen quantizer look-up and the target-side mel projection.  Nothing here has a backward: the
reference computes all of it without gradients (models.py:912-921, sovits.py:470-480)."""
import ctypes as C

import torch

from . import conv as HC
from . import lib as L


def ncl_to_nlc(x, cpad=None, dtype=torch.float32):
    """x fp32 [B, C, T] (reference layout) -> [B, T, cpad] in `dtype`, channels >= C zero: ONE launch for the transpose,
    the cast and the padding (through torch: a transposed view, a contiguous copy, a cast and -- for the 1025-bin
    spectrogram -- a vendor GEMM on unaligned rows)."""
    if x.dim() != 3 or x.dtype != torch.float32:
        raise L.EvtError(f"ncl_to_nlc: fp32 [B, C, T] expected, got {tuple(x.shape)} {x.dtype}")
    x = x.contiguous()
    B, Cc, T = x.shape
    cpad = Cc if cpad is None else int(cpad)
    out = torch.empty((B, T, cpad), dtype=dtype, device=x.device)
    L.check(L.lib().evt_ncl_to_nlc(L.dt_code(dtype), L.ptr(x), L.ptr(out), B, Cc, T, cpad,
                                   L.stream_ptr()), "evt_ncl_to_nlc")
    return out


class RvqEncoder:
    """ssl_proj (Conv1d 768 -> 768, k = 2 / stride 2 at 25 Hz) + nearest code of the frozen codebook + code vectors, in
    fp32: two launches of the library's convolution family on fp32 images (the projection, and x . embed^T as a 1x1 layer
    whose weight is the codebook), then the select kernel.  `weight_fn` / `bias_fn` / `embed_fn` return the live
    tensors (parameters move into the runtime's arena after construction)."""

    def __init__(self, weight_fn, bias_fn, embed_fn, dim, bins, k, stride, device):
        self.embed_fn, self.dim, self.bins = embed_fn, dim, bins
        self.proj = HC.FrozenConv(weight_fn, bias_fn, dim, dim, k=k, stride=stride)
        self.book = HC.FrozenConv(embed_fn, None, dim, bins)
        self.bank = HC.FrozenBank([self.proj, self.book], device)

    def project(self, ssl_ncl):
        """ssl fp32 [B, dim, T] -> h fp32 [B, T', dim]"""
        self.bank.prepare()
        x = ncl_to_nlc(ssl_ncl.float(), None, torch.float32)
        return HC._fwd(self.proj._slot, x, None, 1.0, L.ACT_NONE, 1.0)

    def lookup(self, h, rep):
        """h fp32 [B, T', dim] -> (quantized fp32 [B, T'*rep, dim], codes int64 [B, T'])"""
        self.bank.prepare()
        B, T2, D = h.shape
        N = B * T2
        embed = self.embed_fn()
        dots = HC._fwd(self.book._slot, h.view(1, N, D), None, 1.0, L.ACT_NONE, 1.0)
        ee = torch.empty(self.bins, dtype=torch.float32, device=h.device)
        codes = torch.empty(N, dtype=torch.int64, device=h.device)
        q = torch.empty((B, T2 * rep, D), dtype=torch.float32, device=h.device)
        lib = L.lib()
        L.check(lib.evt_rvq_norms(L.ptr(embed), L.ptr(ee), self.bins, D, L.stream_ptr()), "evt_rvq_norms")
        L.check(lib.evt_rvq_select(L.ptr(h), L.ptr(dots), L.ptr(embed), L.ptr(ee), L.ptr(codes), L.ptr(q), N, D, self.bins,
                                   rep, L.stream_ptr()), "evt_rvq_select")
        return q, codes.view(B, T2)


def spec_to_mel(spec, basis, starts=None, nfr=None):
    """spec fp32 [B, F, T], basis fp32 [M, F] -> log-mel fp32 [B, M, nfr] of the frames starts[b] .. starts[b] + nfr
    (starts None: from frame 0; nfr None: all T frames)"""
    spec = spec.float().contiguous()
    B, F, T = spec.shape
    M = basis.size(0)
    nfr = T if nfr is None else int(nfr)
    if starts is not None:
        starts = starts.to(torch.int64).contiguous()
    out = torch.empty((B, M, nfr), dtype=torch.float32, device=spec.device)
    L.check(L.lib().evt_spec_to_mel(L.ptr(spec), L.ptr(basis.contiguous()), L.ptr(starts), L.ptr(out), B, F, T, M, nfr,
                                    L.stream_ptr()), "evt_spec_to_mel")
    return out
