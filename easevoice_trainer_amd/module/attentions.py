"""Relative-position transformer encoder of the s2 text/ssl encoders, channels-last.

Mirrors (names, parameter keys and arithmetic) src/easevoice/module/attentions.py:12-90 (Encoder),
:179-292 (MultiHeadAttention with window-4 relative position embeddings), :379-435 (FFN) of the
reference.  Tensors are [B, T, C] here ([B, C, T] in the reference); padding is described by per-item lengths.

Everything runs on the library: the projections and FFN convolutions on the fused conv kernels, the attention core
(scores, relative logits, key mask, softmax, dropout, values, relative values) as one launch of csrc/mha.hip each way.
There is no torch attention path: a shape the kernels do not cover raises.
"""
import torch
from torch import nn
from torch.nn import functional as F

from ..hip import lib as L
from ..hip.conv import EvtConv1d
from ..hip.enc import mha_core, new_site, rel_self_attention, relu_dropout, res_drop_ln


class LayerNorm(nn.Module):
    """modules.py:19-31 — LayerNorm over the channel axis (keys gamma/beta)."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x):
        return F.layer_norm(x, (self.channels,), self.gamma, self.beta, self.eps)


class PointwiseEvtConv(EvtConv1d):
    """the same nn.Conv1d(cin, cout, 1) parameters (`weight` [cout, cin, 1], `bias`) on the fused HIP conv path: a
    3200-row x 192..768-column GEMM is launch/latency-bound, the ring-pipelined k = 1 conv does it in ~10 us and its
    weight-gradient launch also produces the bias gradient (a dense-layer call through torch takes three launches for
    that: data gradient, weight gradient, column sum)."""

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


class PaddedInPointwise(PointwiseEvtConv):
    """nn.Conv1d(cin, cout, 1) whose input width is not a multiple of 8 (enc_q.pre: the 1025 spectrogram bins,
    models.py:338): the parameter keeps the reference's shape [cout, cin, 1]; the prepared GEMM image is padded with zero
    columns to `cin_pad` (a multiple of 64) and the module takes rows of cin_pad values whose tail is zero
    (hip/frontend.py::ncl_to_nlc writes them), so the projection runs on the LDS-DMA GEMM kernels like every other 1x1
    layer instead of a vendor GEMM on unaligned rows."""

    def __init__(self, cin, cout):
        super().__init__(cin, cout)
        self.src_d1 = cin                       # the parameter's own row length (evt_wprep_item.src_d1)
        self.cin = (cin + 63) // 64 * 64        # what the kernels see


def pointwise(cin, cout):
    """1x1 conv on [B, T, C] rows on the library's conv kernels; an input width that is not a multiple of 8 (only the
    1025-bin spectrogram projection of the posterior encoder) gets a zero-padded image"""
    if cout % 8:
        raise L.EvtError(f"pointwise({cin}, {cout}): output rows must be 16-byte multiples")
    return PointwiseEvtConv(cin, cout) if cin % 8 == 0 else PaddedInPointwise(cin, cout)


def linear_rows(cin, cout):
    """an nn.Linear (state_dict keys `weight` [cout, cin], `bias`) on the same kernels"""
    if cin % 8 or cout % 8:
        raise L.EvtError(f"linear_rows({cin}, {cout}): rows must be 16-byte multiples")
    return PointwiseEvtConv(cin, cout, kdims=0)


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
        """the three projections when they can run as ONE [3C, C] GEMM: the windowed self-attention layers of the
        encoders (forward() takes the fused node for every self-attention call, so packing and use cannot disagree)"""
        if self.window_size is None:
            return None
        return (self.conv_q, self.conv_k, self.conv_v)

    def arena_adjacent(self):
        """parameter groups the runtime's arena lays out back to back (runtime.ParamArena), so that the packed projection is
        a dense matrix over the members' own storage"""
        if self.qkv_pack_modules() is None:
            return []
        return [["conv_q.weight", "conv_k.weight", "conv_v.weight"], ["conv_q.bias", "conv_k.bias", "conv_v.bias"]]

    def forward(self, x, c, lens, lens_c=None):
        """x [B, Tt, C] queries, c [B, Ts, C] keys/values (`c is x`: self-attention), lens [B] int32 live query rows,
        lens_c [B] int32 live key rows (cross-attention).  The reference's attn_mask (attentions.py:268-269) is the outer
        product of the two length masks: padded keys are excluded, padded query rows come out as zeros (their value is
        discarded by every caller's own mask).
        Self-attention: the three projections + the core are ONE autograd node (hip/enc.py::RelSelfAttnFn).
        Cross-attention (MRTE, mrte_model.py:25-61): three projection launches + the core node."""
        p = self.drop.p if self.training else 0.0
        cd = self.conv_q._slot.bank.dtype if self.conv_q._slot is not None else x.dtype
        self_attn = c is x
        x = x.to(cd).contiguous()
        if self_attn:
            out = rel_self_attention(x, self.conv_q, self.conv_k, self.conv_v, getattr(self, "emb_rel_k", None),
                                     getattr(self, "emb_rel_v", None), lens, self.n_heads, self.window_size, p,
                                     self._site, packed=self._qkv_packed)
            return self.conv_o(out)
        if self.window_size is not None:
            raise L.EvtError("relative attention is only available for self-attention")       # attentions.py:259-261
        c = c.to(cd).contiguous()
        q, k, v = self.conv_q(x), self.conv_k(c), self.conv_v(c)
        out = mha_core(q, k, v, lens, lens_c, self.n_heads, p, self._site, self.k_channels ** -0.5)
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
        p = self.drop.p if self.training else 0.0
        for i in range(self.n_layers):
            n1, n2 = self.norm_layers_1[i], self.norm_layers_2[i]
            y = self.attn_layers[i](x, x, lens)
            x = res_drop_ln(x, y, n1.gamma, n1.beta, lens, p, self._sites[2 * i], n1.eps)
            y = self.ffn_layers[i](x, mask_cd, cd, premasked=True, lens=lens)
            x = res_drop_ln(x, y, n2.gamma, n2.beta, lens, p, self._sites[2 * i + 1], n2.eps)
        return x
