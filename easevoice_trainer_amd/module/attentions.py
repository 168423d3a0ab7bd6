"""Relative-position transformer encoder of the s2 text/ssl encoders, channels-last.

Mirrors (names, parameter keys and arithmetic) src/easevoice/module/attentions.py:12-90 (Encoder),
:179-292 (MultiHeadAttention with window-4 relative position embeddings), :379-435 (FFN) of the
reference.  Tensors are [B, T, C] here ([B, C, T] in the reference); masks are [B, T, 1].

T <= a few hundred frames, so the attention itself stays on rocBLAS GEMMs through torch; the two
k=3 FFN convolutions (most of this block's MACs) run on the fused HIP conv kernel.
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from ..hip import lib as L
from ..hip.conv import EvtConv1d
from ..hip.enc import new_site, rel_self_attention, relu_dropout, res_drop_ln


class LayerNorm(nn.Module):
    """modules.py:19-31 — LayerNorm over the channel axis (keys gamma/beta)."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x):
        return F.layer_norm(x, (self.channels,), self.gamma, self.beta, self.eps)


class PointwiseConv(nn.Module):
    """nn.Conv1d(cin, cout, 1) parameters ([cout, cin, 1] weight) applied as a GEMM on [B, T, C]."""

    def __init__(self, cin, cout, bias=True):
        super().__init__()
        c = nn.Conv1d(cin, cout, 1, bias=bias)
        self.weight = c.weight
        self.bias = c.bias

    def forward(self, x):
        return F.linear(x, self.weight.squeeze(-1), self.bias)


class PointwiseEvtConv(EvtConv1d):
    """the same nn.Conv1d(cin, cout, 1) parameters (`weight` [cout, cin, 1], `bias`) on the fused HIP conv path: a
    3200-row x 192..768-column GEMM is launch/latency-bound, the ring-pipelined k = 1 conv does it in ~10 us and its
    weight-gradient launch also produces the bias gradient (F.linear needs dgrad + wgrad + a column-sum launch)."""

    def __init__(self, cin, cout, kdims=1, weight_norm=False):
        super().__init__(cin, cout, 1, kdims=kdims, weight_norm=weight_norm)

    def forward(self, x):
        """x [..., cin] -> [..., cout]: any leading shape (a [B, cin] vector is one sequence of B rows)"""
        if self._slot is not None:
            cd = self._slot.bank.dtype
            if x.dtype != cd or not x.is_contiguous():
                x = x.to(cd).contiguous()
        if x.dim() == 3:
            return super().forward(x)
        lead = x.shape[:-1]
        return super().forward(x.reshape(1, -1, x.size(-1))).reshape(*lead, self.cout)


def pointwise(cin, cout):
    """1x1 conv on [B, T, C]: the library's conv kernels whenever the rows are 16-byte aligned (both widths multiples
    of 8); only the 1025-bin spectrogram projection of the posterior encoder stays on a vendor GEMM"""
    return PointwiseEvtConv(cin, cout) if cin % 8 == 0 and cout % 8 == 0 else PointwiseConv(cin, cout)


def linear_rows(cin, cout):
    """an nn.Linear (state_dict keys `weight` [cout, cin], `bias`) on the same kernels"""
    return PointwiseEvtConv(cin, cout, kdims=0) if cin % 8 == 0 and cout % 8 == 0 else nn.Linear(cin, cout)


class MultiHeadAttention(nn.Module):
    def __init__(self, channels, out_channels, n_heads, p_dropout=0.0, window_size=None, heads_share=True):
        super().__init__()
        assert channels % n_heads == 0
        self.channels, self.out_channels, self.n_heads = channels, out_channels, n_heads
        self.p_dropout, self.window_size = p_dropout, window_size
        self.k_channels = channels // n_heads
        self.conv_q = pointwise(channels, channels)
        self.conv_k = pointwise(channels, channels)
        self.conv_v = pointwise(channels, channels)
        self.conv_o = pointwise(channels, out_channels)
        self.drop = nn.Dropout(p_dropout)
        if window_size is not None:
            n_heads_rel = 1 if heads_share else n_heads
            std = self.k_channels ** -0.5
            self.emb_rel_k = nn.Parameter(torch.randn(n_heads_rel, window_size * 2 + 1, self.k_channels) * std)
            self.emb_rel_v = nn.Parameter(torch.randn(n_heads_rel, window_size * 2 + 1, self.k_channels) * std)
        nn.init.xavier_uniform_(self.conv_q.weight)
        nn.init.xavier_uniform_(self.conv_k.weight)
        nn.init.xavier_uniform_(self.conv_v.weight)
        self._site = new_site()      # dropout stream id of the fused attention launch
        self._qkv_packed = None      # hip/conv.py::PackedConv of the three projections, set by the bf16 WeightBank

    def qkv_pack_modules(self):
        """the three projections when they can run as ONE [3C, C] GEMM: windowed self-attention layers (the fused path)"""
        qkv = (self.conv_q, self.conv_k, self.conv_v)
        if self.window_size is None or not all(isinstance(c, PointwiseEvtConv) for c in qkv):
            return None
        return qkv

    def arena_adjacent(self):
        """parameter groups the runtime's arena lays out back to back (runtime.ParamArena), so that the packed projection is
        a dense matrix over the members' own storage"""
        if self.qkv_pack_modules() is None:
            return []
        return [["conv_q.weight", "conv_k.weight", "conv_v.weight"], ["conv_q.bias", "conv_k.bias", "conv_v.bias"]]

    @staticmethod
    def _rel_to_abs(x):
        """[b, h, l, 2l-1] relative logits -> [b, h, l, l] (entry (i, j) = rel[i, j - i + l - 1]) by the
        pad / reshape skew: pure copies forward and backward, no gather/scatter atomics."""
        b, h, l, _ = x.shape
        x = F.pad(x, (0, 1)).reshape(b, h, l * 2 * l)
        x = F.pad(x, (0, l - 1)).reshape(b, h, l + 1, 2 * l - 1)
        return x[:, :, :l, l - 1:]

    @staticmethod
    def _abs_to_rel(x):
        """[b, h, l, l] -> [b, h, l, 2l-1] (entry (i, r) = abs[i, i + r - (l-1)], zero outside)"""
        b, h, l, _ = x.shape
        x = F.pad(x, (0, l - 1)).reshape(b, h, l * (2 * l - 1))
        x = F.pad(x, (l, 0)).reshape(b, h, l, 2 * l)
        return x[:, :, :, 1:]

    def _band_to_full(self, band, length):
        """[.., 2w+1] band around offset 0 -> [.., 2l-1] full relative axis (zeros outside the window)"""
        w = self.window_size
        extra = length - 1 - w
        if extra >= 0:
            return F.pad(band, (extra, extra))
        return band[..., -extra: band.size(-1) + extra]

    def _full_to_band(self, full, length):
        w = self.window_size
        extra = length - 1 - w
        if extra >= 0:
            return full[..., extra: extra + 2 * w + 1]
        return F.pad(full, (-extra, -extra))

    def fused_ok(self, x, c):
        """self-attention with relative window in bf16 on the GPU -> csrc/relattn.hip"""
        return (x is c and self.window_size is not None and x.is_cuda and x.dtype == torch.bfloat16
                and x.is_contiguous() and isinstance(self.conv_q, PointwiseEvtConv)
                and self.k_channels % 32 == 0 and self.k_channels <= 128 and 2 * self.window_size + 1 <= 16)

    def forward(self, x, c, attn_mask=None, lens=None):
        """x [B, Tt, C] queries, c [B, Ts, C] keys/values, attn_mask [B, 1, Tt, Ts] (1 = attend).
        With `lens` [B] int32 (live frames, the mask being lens x lens) and bf16 self-attention, the whole core --
        scores, relative logits, mask, softmax, dropout, values, relative values -- is one fused launch behind the three
        1x1 projections, all of it one autograd node (hip/enc.py::RelSelfAttnFn)."""
        if lens is not None and self.fused_ok(x, c):
            p = self.drop.p if self.training else 0.0
            out = rel_self_attention(x, self.conv_q, self.conv_k, self.conv_v, self.emb_rel_k, self.emb_rel_v, lens,
                                     self.n_heads, self.window_size, p, self._site, packed=self._qkv_packed)
            return self.conv_o(out)
        b, t_t, _ = x.shape
        t_s = c.size(1)
        h, d = self.n_heads, self.k_channels
        q = self.conv_q(x).view(b, t_t, h, d).transpose(1, 2)    # [b, h, t, d]
        k = self.conv_k(c).view(b, t_s, h, d).transpose(1, 2)
        v = self.conv_v(c).view(b, t_s, h, d).transpose(1, 2)
        qs = q / math.sqrt(d)
        scores = torch.matmul(qs, k.transpose(-2, -1))
        if self.window_size is not None:
            assert t_s == t_t, "relative attention is only available for self-attention"
            # logits against the 2w+1 relative key embeddings, skewed onto the |i-j| <= w band
            qe = torch.matmul(qs, self.emb_rel_k.unsqueeze(0).transpose(-2, -1))       # [b, h, l, 2w+1]
            scores = scores + self._rel_to_abs(self._band_to_full(qe, t_s))
        if attn_mask is not None:
            scores = scores.masked_fill(attn_mask == 0, -1e4)
        p_attn = self.drop(F.softmax(scores, dim=-1))
        out = torch.matmul(p_attn, v)
        if self.window_size is not None:
            relw = self._full_to_band(self._abs_to_rel(p_attn), t_s)                    # [b, h, l, 2w+1]
            out = out + torch.matmul(relw, self.emb_rel_v.unsqueeze(0))
        out = out.transpose(1, 2).reshape(b, t_t, h * d)
        return self.conv_o(out)


class FFN(nn.Module):
    """conv(k) -> relu -> dropout -> conv(k), 'same' padding, masked (attentions.py:408-416); both convs
    run on the fused HIP kernel, the relu is the first conv's epilogue."""

    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0.0):
        super().__init__()
        assert kernel_size % 2 == 1
        self.conv_1 = EvtConv1d(in_channels, filter_channels, kernel_size, padding=(kernel_size - 1) // 2)
        self.conv_2 = EvtConv1d(filter_channels, out_channels, kernel_size, padding=(kernel_size - 1) // 2)
        self.drop = nn.Dropout(p_dropout)
        self._site = new_site()

    def forward(self, x, x_mask, cd, premasked=False, lens=None):
        """premasked: x is already x * x_mask in the compute dtype and the caller masks the result (Encoder).
        With `lens` on the GPU the relu, the dropout and the `* x_mask` between the convs are ONE launch."""
        if premasked and lens is not None and x.is_cuda:
            p = self.drop.p if self.training else 0.0
            h = relu_dropout(self.conv_1(x), p, self._site, lens)
            return self.conv_2(h)
        if not premasked:
            x = (x * x_mask).to(cd).contiguous()
        x = self.conv_1(x, out_act=L.ACT_LRELU, out_slope=0.0)
        x = self.drop(x)
        x = self.conv_2((x * x_mask).to(cd).contiguous())
        return x if premasked else x * x_mask


class Encoder(nn.Module):
    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0.0,
                 window_size=4):
        super().__init__()
        self.hidden_channels, self.n_layers = hidden_channels, n_layers
        self.drop = nn.Dropout(p_dropout)
        self.attn_layers = nn.ModuleList()
        self.norm_layers_1 = nn.ModuleList()
        self.ffn_layers = nn.ModuleList()
        self.norm_layers_2 = nn.ModuleList()
        for _ in range(n_layers):
            self.attn_layers.append(MultiHeadAttention(hidden_channels, hidden_channels, n_heads, p_dropout=p_dropout,
                                                       window_size=window_size))
            self.norm_layers_1.append(LayerNorm(hidden_channels))
            self.ffn_layers.append(FFN(hidden_channels, hidden_channels, filter_channels, kernel_size,
                                       p_dropout=p_dropout))
            self.norm_layers_2.append(LayerNorm(hidden_channels))
        self._sites = [new_site() for _ in range(2 * n_layers)]     # dropout stream ids of the fused drop+add+norm launches

    def forward(self, x, x_mask, cd, lengths=None):
        """x [B, T, C], x_mask [B, T, 1] (1 = live frame), lengths [B] (optional, = x_mask.sum(1)).
        Returns x * x_mask in the compute dtype.  Per layer: attention, then ONE fused launch for
        drop -> add -> LayerNorm -> mask (hip/enc.py), FFN, and the same fused launch again."""
        lens = (lengths if lengths is not None else x_mask.sum(dim=(1, 2))).to(torch.int32)
        mask_cd = x_mask.to(cd)
        x = (x * x_mask).to(cd).contiguous()
        fused = self.n_layers > 0 and self.attn_layers[0].fused_ok(x, x)
        attn_mask = None if fused else (x_mask.transpose(1, 2).unsqueeze(2) * x_mask.unsqueeze(1))   # [B, 1, T, T]
        p = self.drop.p if self.training else 0.0
        for i in range(self.n_layers):
            n1, n2 = self.norm_layers_1[i], self.norm_layers_2[i]
            y = self.attn_layers[i](x, x, attn_mask, lens=lens).to(cd).contiguous()
            x = res_drop_ln(x, y, n1.gamma, n1.beta, lens, p, self._sites[2 * i], n1.eps)
            y = self.ffn_layers[i](x, mask_cd, cd, premasked=True, lens=lens)
            x = res_drop_ln(x, y, n2.gamma, n2.beta, lens, p, self._sites[2 * i + 1], n2.eps)
        return x
