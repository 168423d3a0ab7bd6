"""s2 SoVITS generator (SynthesizerTrn) and discriminators (MultiPeriodDiscriminator), MI355X-native.

Same class names, constructor arguments, parameter/buffer keys and arithmetic as the reference's
src/easevoice/module/models.py:174-471,481-614,803-946 (+ modules.py:135-317,404-458,685-763,
mrte_model.py:9-61, core_vq.py:172-228), re-laid-out for the hardware:

  * every activation is channels-last [B, T, C] so the conv stacks are implicit GEMMs whose K index
    (tap, channel) is contiguous in HBM/LDS; the reference's [B, C, T] only exists at the module boundary;
  * every Conv1d / ConvTranspose1d / Conv2d((k,1)) is one fused HIP launch (hip/conv.py): leaky-relu
    prologue, bias / activation / residual epilogues, weight-norm folded once per step for the whole model;
  * 1x1 convolutions / nn.Linear layers on [B, T, C] rows are k = 1 members of the same conv family; the attention cores
    (relative-position self-attention, MRTE cross-attention, style self-attention) are csrc/mha.hip; the frozen
    quantizer, the spectrogram's layout change and the target mel are csrc/frontend.hip -- no vendor GEMM, no torch
    matmul / softmax on the training path;
  * DiscriminatorP's period axis is laid out as extra sequences ([B*p, T/p, C]) instead of a 2-D image.

There is no CPU / eager fallback: the modules need a WeightBank (hip/conv.py) and a GPU.
"""

import os

import torch
from torch import nn
from torch.nn import functional as F

from ..hip import lib as L
from ..hip.conv import Add3ScaleFn, EvtConv1d, GatedActFn, res_stage, res_unit
from ..hip.enc import new_site, rel_self_attention, unbind_rows, wn_residual, wn_residual_last
from ..hip.frontend import RvqEncoder, ncl_to_nlc
from ..hip.wn import wn_stack
from . import commons
from .attentions import Encoder, MultiHeadAttention, PointwiseEvtConv, linear_rows, pointwise

LRELU_SLOPE = 0.1
N_SYMBOLS = 732  # len(SYMBOLS), src/easevoice/text/symbols.py:410-412 (pinned by tests/easevoice/text_test.py)


def get_padding(kernel_size, dilation=1):
    return (kernel_size * dilation - dilation) // 2


class _ComputeDtype:
    """mix-in: `self.cd` is the compute dtype of the fused conv kernels (set by runtime.attach)"""
    cd = torch.float32


# --------------------------------------------------------------------------------------------------
# WN / posterior encoder / flow
# --------------------------------------------------------------------------------------------------
class WeightNormPointwise(PointwiseEvtConv):
    """weight-normed nn.Conv1d(cin, cout, 1) applied to a [B, C] vector (the WN cond_layer on ge): keys bias, weight_g,
    weight_v like the reference; the weight-norm fold and its gradient are the bank's multi-tensor launches, the product
    a 1x1 conv over the B rows (through torch this was a norm / divide / multiply chain, a vendor GEMM and their
    backward per call)."""

    def __init__(self, cin, cout):
        super().__init__(cin, cout, weight_norm=True)


class WN(nn.Module, _ComputeDtype):
    """modules.py:135-212.  in_layers (k=5) and res_skip_layers (1x1) are fused HIP convs, the gated
    tanh*sigmoid with the conditioning add is one HIP element-wise launch."""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert kernel_size % 2 == 1
        self.hidden_channels, self.n_layers, self.gin_channels = hidden_channels, n_layers, gin_channels
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        self.drop = nn.Dropout(p_dropout)
        if gin_channels != 0:
            self.cond_layer = WeightNormPointwise(gin_channels, 2 * hidden_channels * n_layers)
        for i in range(n_layers):
            dilation = dilation_rate ** i
            self.in_layers.append(EvtConv1d(hidden_channels, 2 * hidden_channels, kernel_size, dilation=dilation,
                                            padding=(kernel_size * dilation - dilation) // 2, weight_norm=True))
            rs = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(EvtConv1d(hidden_channels, rs, 1, weight_norm=True))

    def forward(self, x, x_mask, g=None, lens=None):
        """x [B, T, H] (compute dtype), x_mask [B, T, 1], g [B, gin] or None, lens [B] int32 (= x_mask.sum(1)).
        Per layer: in_layer conv, gated activation, res_skip conv (HIP) and ONE fused launch for the residual / skip
        bookkeeping  x <- (x + rs[:H]) * mask, out <- out + rs[H:]  (hip/enc.py)."""
        H = self.hidden_channels
        if lens is None:
            lens = x_mask.sum(dim=(1, 2)).to(torch.int32)
        x = x.contiguous()
        if x.is_cuda and not (self.training and self.drop.p > 0):
            # the whole stack as one autograd node (hip/wn.py): same launches, no torch glue between them
            g_lbh = None
            if g is not None:
                g_lbh = self.cond_layer(g).to(x.dtype).view(g.size(0), self.n_layers, 2 * H).transpose(0, 1).contiguous()
            return wn_stack(x, g_lbh, lens, self.in_layers, self.res_skip_layers, H)
        gs = None
        if g is not None:
            g = self.cond_layer(g).to(x.dtype)     # [B, 2*H*n_layers]
            gs = unbind_rows(g.view(g.size(0), self.n_layers, 2 * H).transpose(0, 1).contiguous())
        output = None
        for i in range(self.n_layers):
            x_in = self.in_layers[i](x)
            acts = self.drop(GatedActFn.apply(x_in, gs[i] if gs is not None else None))
            rs = self.res_skip_layers[i](acts)
            if i < self.n_layers - 1:
                x, output = wn_residual(x, rs, output, lens)
            else:
                output = wn_residual_last(rs, output, lens)
        return output


class PosteriorEncoder(nn.Module, _ComputeDtype):
    """models.py:318-359."""

    def __init__(self, in_channels, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                 gin_channels=0):
        super().__init__()
        self.out_channels = out_channels
        self.pre = pointwise(in_channels, hidden_channels)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels)
        self.proj = pointwise(hidden_channels, out_channels * 2)

    def forward(self, x, x_mask, g=None, eps=None, lens=None):
        """x [B, T, self.pre.cin] channels-last in the compute dtype (the spectrogram after hip/frontend.py::ncl_to_nlc:
        transposed, cast and zero-padded to the projection's image width in one launch) -> z, m, logs [B, T, out];
        eps (the randn_like draw of models.py:358) may be injected"""
        if g is not None:
            g = g.detach()
        h = (self.pre(x) * x_mask).to(self.cd).contiguous()
        h = self.enc(h, x_mask, g=g, lens=lens)
        if h.is_cuda and lens is not None:
            # mask, split, cast, exp, scale, add, mask: one launch (hip/enc.py::ReparamFn)
            from ..hip.enc import reparam

            stats = self.proj(h)
            if eps is None:
                eps = torch.randn((stats.size(0), stats.size(1), self.out_channels), dtype=torch.float32, device=h.device)
            return reparam(stats, eps, lens)
        stats = self.proj(h) * x_mask
        m, logs = torch.split(stats.float(), self.out_channels, dim=-1)
        if eps is None:
            eps = torch.randn_like(m)
        z = (m + eps * torch.exp(logs)) * x_mask
        return z, m, logs


class ResidualCouplingLayer(nn.Module, _ComputeDtype):
    """modules.py:404-458 (mean_only coupling)."""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0, gin_channels=0,
                 mean_only=False):
        super().__init__()
        assert channels % 2 == 0
        self.half_channels, self.mean_only = channels // 2, mean_only
        self.pre = pointwise(self.half_channels, hidden_channels)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=p_dropout,
                      gin_channels=gin_channels)
        self.post = pointwise(hidden_channels, self.half_channels * (2 - mean_only))
        self.post.weight.data.zero_()
        self.post.bias.data.zero_()

    def forward(self, x, x_mask, g=None, reverse=False, lens=None):
        x0, x1 = torch.split(x, [self.half_channels] * 2, dim=-1)
        h = (self.pre(x0) * x_mask).to(self.cd).contiguous()
        h = self.enc(h, x_mask, g=g, lens=lens)
        stats = (self.post(h) * x_mask).float()
        if self.mean_only:
            # logs == 0 (modules.py:449-452): exp(logs) is 1, the affine step is a masked shift -- no zero tensor, no exp,
            # no multiply by one (and none of their backward launches)
            x1 = stats + x1 * x_mask if not reverse else (x1 - stats) * x_mask
            return torch.cat([x0, x1], dim=-1)
        m, logs = torch.split(stats, [self.half_channels] * 2, dim=-1)
        if not reverse:
            x1 = m + x1 * torch.exp(logs) * x_mask
        else:
            x1 = (x1 - m) * torch.exp(-logs) * x_mask
        return torch.cat([x0, x1], dim=-1)

    def forward_flip(self, x, x0c, x_mask, mask_cd, g, lens, want_next):
        """this layer followed by its Flip, training direction, mean_only, on the GPU: (flipped output [B, T, C] fp32, its
        first half in the compute dtype for the next layer's `pre` or None).  x0c: x[..., :half] in the compute dtype as
        the previous layer left it (None: taken from x)."""
        from ..hip.enc import coupling_flip

        if x0c is None:
            x0c = x[..., :self.half_channels]
        h = (self.pre(x0c) * mask_cd).contiguous()
        h = self.enc(h, x_mask, g=g, lens=lens)
        return coupling_flip(x, self.post(h), lens, want_next)


class Flip(nn.Module):
    def forward(self, x, *args, **kwargs):
        return torch.flip(x, [-1])


class ResidualCouplingBlock(nn.Module):
    """models.py:273-315."""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, n_flows=4, gin_channels=0):
        super().__init__()
        self.flows = nn.ModuleList()
        for _ in range(n_flows):
            self.flows.append(ResidualCouplingLayer(channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                                                    gin_channels=gin_channels, mean_only=True))
            self.flows.append(Flip())

    def forward(self, x, x_mask, g=None, reverse=False, lens=None):
        if (not reverse and x.is_cuda and lens is not None and x.dtype == torch.float32
                and all(f.mean_only for f in self.flows[0::2])):
            # every coupling layer with its Flip: the element-wise tail of a layer and the head of the next in one launch
            x0c = None
            n = len(self.flows) // 2
            mask_cd = x_mask.to(self.flows[0].cd)
            for i in range(n):
                x, x0c = self.flows[2 * i].forward_flip(x, x0c, x_mask, mask_cd, g, lens, want_next=i + 1 < n)
            return x
        flows = self.flows if not reverse else reversed(self.flows)
        for flow in flows:
            x = flow(x, x_mask, g=g, reverse=reverse, lens=lens)
        return x


# --------------------------------------------------------------------------------------------------
# text / ssl encoder
# --------------------------------------------------------------------------------------------------
class MRTE(nn.Module):
    """mrte_model.py:9-61."""

    def __init__(self, content_enc_channels=192, hidden_size=512, out_channels=192, n_heads=4):
        super().__init__()
        self.cross_attention = MultiHeadAttention(hidden_size, hidden_size, n_heads)
        self.c_pre = pointwise(content_enc_channels, hidden_size)
        self.text_pre = pointwise(content_enc_channels, hidden_size)
        self.c_post = pointwise(hidden_size, out_channels)

    def forward(self, ssl_enc, ssl_mask, text, text_mask, ge, ssl_lens, text_lens):
        """ssl_enc [B, T, C], text [B, Tt, C], ge [B, 512] or None; ssl_lens / text_lens [B] int32 (the masks as lengths:
        the reference's attn_mask = text_mask x ssl_mask, mrte_model.py:28)"""
        ssl_enc = self.c_pre(ssl_enc * ssl_mask)
        text_enc = self.text_pre(text * text_mask)
        x = self.cross_attention(ssl_enc * ssl_mask, text_enc * text_mask, ssl_lens, text_lens) + ssl_enc
        if ge is not None:
            x = x + ge.unsqueeze(1)
        return self.c_post(x * ssl_mask)


class TextEncoder(nn.Module, _ComputeDtype):
    """models.py:174-251."""

    def __init__(self, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout,
                 latent_channels=192, version="v2"):
        super().__init__()
        self.out_channels = out_channels
        self.ssl_proj = pointwise(768, hidden_channels)
        self.encoder_ssl = Encoder(hidden_channels, filter_channels, n_heads, n_layers // 2, kernel_size, p_dropout)
        self.encoder_text = Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.text_embedding = nn.Embedding(N_SYMBOLS, hidden_channels)
        self.mrte = MRTE()
        self.encoder2 = Encoder(hidden_channels, filter_channels, n_heads, n_layers // 2, kernel_size, p_dropout)
        self.proj = pointwise(hidden_channels, out_channels * 2)

    def forward(self, y, y_mask, text, text_mask, ge, y_lengths=None, text_lengths=None, speed=1):
        """y [B, T, 768] (quantized ssl), text [B, Tt] ids, ge [B, 512]; the encoders mask their own input / output.
        speed != 1 (inference only, models.py:246-248) resamples the encoded sequence to int(T/speed)+1 frames before
        the projection and returns the resampled mask as a fourth value."""
        yl = (y_lengths if y_lengths is not None else y_mask.sum(dim=(1, 2))).to(torch.int32)
        tl = (text_lengths if text_lengths is not None else text_mask.sum(dim=(1, 2))).to(torch.int32)
        y = self.ssl_proj(y * y_mask)
        y = self.encoder_ssl(y, y_mask, self.cd, lengths=yl)
        t = self.text_embedding(text)
        t = self.encoder_text(t, text_mask, self.cd, lengths=tl)
        y = self.mrte(y, y_mask, t, text_mask, ge, yl, tl)
        y = self.encoder2(y, y_mask, self.cd, lengths=yl)
        if speed != 1:
            n = int(y.size(1) / speed) + 1
            y = F.interpolate(y.transpose(1, 2).float(), size=n, mode="linear").transpose(1, 2).to(y.dtype).contiguous()
            y_mask = F.interpolate(y_mask.transpose(1, 2), size=n, mode="nearest").transpose(1, 2).contiguous()
        stats = (self.proj(y) * y_mask).float()
        m, logs = torch.split(stats, self.out_channels, dim=-1)
        return (y, m, logs) if speed == 1 else (y, m, logs, y_mask)


# --------------------------------------------------------------------------------------------------
# style encoder
# --------------------------------------------------------------------------------------------------
class LinearNorm(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        if not bias:
            raise L.EvtError("LinearNorm without a bias is not used by the style encoder (modules.py:521-545)")
        self.fc = linear_rows(cin, cout)

    def forward(self, x):
        return self.fc(x)


class Mish(nn.Module):
    def forward(self, x):
        return x * torch.tanh(F.softplus(x))


class ConvNorm(nn.Module):
    def __init__(self, cin, cout, kernel_size):
        super().__init__()
        self.conv = EvtConv1d(cin, cout, kernel_size, padding=(kernel_size - 1) // 2)

    def forward(self, x):
        return self.conv(x)


class Conv1dGLU(nn.Module, _ComputeDtype):
    """modules.py:548-566."""

    def __init__(self, in_channels, out_channels, kernel_size, dropout):
        super().__init__()
        self.out_channels = out_channels
        self.conv1 = ConvNorm(in_channels, 2 * out_channels, kernel_size)
        self.dropout = nn.Dropout(dropout)
        self._site = new_site()

    def forward(self, x):
        residual = x
        h = self.conv1(x.to(self.cd).contiguous())
        if h.is_cuda:
            # sigmoid, product, dropout and the residual add in one launch (hip/enc.py)
            from ..hip.enc import glu_dropout_res

            return glu_dropout_res(h, residual, self.dropout.p if self.training else 0.0, self._site)
        x1, x2 = torch.split(h, self.out_channels, dim=-1)
        return residual + self.dropout(x1 * torch.sigmoid(x2))


class StyleAttention(nn.Module):
    """modules.py:605-682 (MultiHeadAttention + ScaledDotProductAttention of the style encoder)."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.0):
        super().__init__()
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_qs = linear_rows(d_model, n_head * d_k)
        self.w_ks = linear_rows(d_model, n_head * d_k)
        self.w_vs = linear_rows(d_model, n_head * d_v)
        if d_k != d_v:
            raise L.EvtError("the fused attention core takes one head width (d_k == d_v, as the style encoder has it)")
        self.temperature = float(d_model) ** 0.5
        self.fc = linear_rows(n_head * d_v, d_model)
        self.dropout = nn.Dropout(dropout)
        self.attn_dropout = nn.Dropout(dropout)
        self._site = new_site()

    def forward(self, x, lens):
        """x [B, T, d_model]; lens [B] int32 live frames (the reference's [B, T, T] mask blocks padded KEYS with -inf,
        modules.py:672-673).  The three projections + softmax(q k^T / sqrt(d_model)) v are one autograd node on the
        library (hip/enc.py::RelSelfAttnFn, no relative positions); padded query rows come out as fc(0) + x and are
        dropped by the encoder's masked mean."""
        p = self.attn_dropout.p if self.training else 0.0
        cd = self.w_qs._slot.bank.dtype if self.w_qs._slot is not None else x.dtype
        x = x.to(cd).contiguous()
        out = rel_self_attention(x, self.w_qs, self.w_ks, self.w_vs, None, None, lens, self.n_head, None, p, self._site,
                                 scale=1.0 / self.temperature)
        return self.dropout(self.fc(out)) + x


class MelStyleEncoder(nn.Module):
    """modules.py:685-763."""

    def __init__(self, n_mel_channels=80, style_hidden=128, style_vector_dim=256, style_kernel_size=5, style_head=2,
                 dropout=0.1):
        super().__init__()
        self.spectral = nn.Sequential(LinearNorm(n_mel_channels, style_hidden), Mish(), nn.Dropout(dropout),
                                      LinearNorm(style_hidden, style_hidden), Mish(), nn.Dropout(dropout))
        self.temporal = nn.Sequential(Conv1dGLU(style_hidden, style_hidden, style_kernel_size, dropout),
                                      Conv1dGLU(style_hidden, style_hidden, style_kernel_size, dropout))
        self.slf_attn = StyleAttention(style_head, style_hidden, style_hidden // style_head,
                                       style_hidden // style_head, dropout)
        self.fc = LinearNorm(style_hidden, style_vector_dim)
        self._sites = (new_site(), new_site())
        # set by a caller that pads the time axis beyond the reference's collate length (train/data.py, EVT_PAD_FRAMES): see forward
        self.mask_beyond_collate = False

    def _spectral(self, x):
        """self.spectral with Mish + Dropout as one launch per pair on the GPU (the Sequential keeps the reference's
        state_dict keys spectral.0 / spectral.3)"""
        sp = self.spectral
        if not x.is_cuda:
            return sp(x)
        from ..hip.enc import mish_dropout

        x = mish_dropout(sp[0](x), sp[2].p if self.training else 0.0, self._sites[0])        # feeds a projection: its dtype
        # the second activation starts the fp32 residual stream of `temporal` (as under the reference's autocast)
        return mish_dropout(sp[3](x), sp[5].p if self.training else 0.0, self._sites[1], torch.float32)

    def forward(self, x, x_mask, lens=None):
        """x [B, T, n_mel], x_mask [B, T, 1], lens [B] (= x_mask.sum(1)) -> [B, style_vector_dim]"""
        pad = x_mask.squeeze(-1) == 0                     # [B, T] True = padding
        if lens is None:
            lens = x_mask.sum(dim=(1, 2))
        x = self._spectral(x)
        if self.mask_beyond_collate:
            # The reference's style encoder is the one consumer that does NOT mask by the lengths before a convolution over
            # time (modules.py:748-756): `temporal` (two k = 5 convolutions) runs over spectral(zero frames) != 0 behind an
            # item's end and over the convolution's own zero padding behind the TENSOR's end, which the reference's collate
            # puts at 2 * (longest // 2 + 1) frames (data_utils.py:189-193).  A batch whose time axis was padded further
            # (EVT_PAD_FRAMES) must look the same to those convolutions: frames the reference's tensor would not have had
            # contribute zeros.  Device-side arithmetic on the lengths: no host read, replayable.
            t_ref = 2 * (torch.div(lens.max(), 2, rounding_mode="floor") + 1)
            live = (torch.arange(x.size(1), device=x.device) < t_ref).to(x.dtype).view(1, -1, 1)
            x = x * live
        x = self.temporal(x)
        x = x.masked_fill(pad.unsqueeze(-1), 0)
        x = self.slf_attn(x, lens.to(torch.int32))
        x = self.fc(x)
        n = (~pad).sum(dim=1, keepdim=True)
        return x.masked_fill(pad.unsqueeze(-1), 0).sum(dim=1) / n


# --------------------------------------------------------------------------------------------------
# frozen residual vector quantizer (n_q = 1), eval semantics only (models.py:912-926)
# --------------------------------------------------------------------------------------------------
class _Codebook(nn.Module):
    KMEANS_ITERS = 50          # ResidualVectorQuantizer default, quantize.py:48
    KMEANS_SAMPLES = 500       # core_vq.py:72

    def __init__(self, dim, codebook_size):
        super().__init__()
        self.codebook_size = codebook_size
        self.register_buffer("inited", torch.Tensor([False]))   # kmeans_init=True in the reference
        self.register_buffer("cluster_size", torch.zeros(codebook_size))
        self.register_buffer("embed", torch.zeros(codebook_size, dim))
        self.register_buffer("embed_avg", torch.zeros(codebook_size, dim))

    @torch.no_grad()
    def init_embed_(self, data):
        """k-means initialisation from the first batch, core_vq.py:61-92,140-149 (a run without a pretrained generator):
        the first 500 vectors, centres drawn with randperm (randint when there are fewer vectors than codes), 50 Lloyd
        iterations with the (s - m)^2 distance of the reference, empty clusters keep their centre.  The reference
        materialises the [500, K, D] difference tensor (1.5 GB); here the samples go through in slices of 50."""
        samples = data[:self.KMEANS_SAMPLES].float()
        n, K = samples.shape[0], self.codebook_size
        if n >= K:
            idx = torch.randperm(n, device=samples.device)[:K]
        else:
            idx = torch.randint(0, n, (K,), device=samples.device)
        means = samples[idx]
        bins = None
        for _ in range(self.KMEANS_ITERS):
            buckets = torch.cat([(-((samples[i:i + 50, None, :] - means[None]) ** 2).sum(dim=-1)).max(dim=-1).indices
                                 for i in range(0, n, 50)])
            bins = torch.bincount(buckets, minlength=K)
            zero = bins == 0
            new_means = torch.zeros_like(means)
            new_means.scatter_add_(0, buckets[:, None].expand(-1, samples.shape[1]), samples)
            new_means = new_means / bins.masked_fill(zero, 1)[:, None]
            means = torch.where(zero[:, None], means, new_means)
        self.embed.copy_(means)
        self.embed_avg.copy_(means)
        self.cluster_size.copy_(bins.to(self.cluster_size.dtype))
        self.inited.fill_(1.0)


class _VQLayer(nn.Module):
    def __init__(self, dim, codebook_size):
        super().__init__()
        self._codebook = _Codebook(dim, codebook_size)


class _RVQ(nn.Module):
    def __init__(self, dim, codebook_size, n_q):
        super().__init__()
        self.layers = nn.ModuleList([_VQLayer(dim, codebook_size) for _ in range(n_q)])


class ResidualVectorQuantizer(nn.Module):
    """quantize.py:29-94 / core_vq.py:95-357, inference path of the frozen quantizer: look-up only.
    (The reference runs it in eval mode inside the training forward, so there is no EMA update, no
    commitment loss and no gradient: models.py:912-921.)"""

    def __init__(self, dimension=256, n_q=8, bins=1024):
        super().__init__()
        self.n_q, self.dimension, self.bins = n_q, dimension, bins
        self.vq = _RVQ(dimension, bins, n_q)

    def ensure_init(self, h):
        """h [B, T, D] fp32 projected features of the first batch.  Checked once (a host sync), not every step: without
        a pretrained codebook, k-means on the first batch like the reference (core_vq.py:140-149); with several ranks
        every rank takes rank 0's result (DDP's buffer broadcast does that in the reference)"""
        if getattr(self, "_inited_ok", False):
            return
        cb = self.vq.layers[0]._codebook
        if not bool(cb.inited.cpu().item()):
            cb.init_embed_(h.reshape(-1, h.size(-1)))
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                for buf in (cb.embed, cb.embed_avg, cb.cluster_size, cb.inited):
                    dist.broadcast(buf, src=0)
        self._inited_ok = True

    @torch.no_grad()
    def decode(self, codes):
        """codes [n_q, B, T] -> quantized [B, T, D] channels-last: sum of the layers' code vectors (quantize.py:112-119,
        core_vq.py:367-373)"""
        q = None
        for layer, ind in zip(self.vq.layers, codes):
            e = F.embedding(ind, layer._codebook.embed)
            q = e if q is None else q + e
        return q


def rvq_encode(enc, quantizer, ssl, rep):
    """ssl fp32 [B, D, T] -> (quantized fp32 [B, T' * rep, D], codes [1, B, T']): the projection, the k-means
    initialisation of an empty codebook on the first call, the nearest-code look-up (hip/frontend.py::RvqEncoder)"""
    h = enc.project(ssl)
    quantizer.ensure_init(h)
    q, codes = enc.lookup(h, rep)
    return q, codes.unsqueeze(0)


# --------------------------------------------------------------------------------------------------
# HiFi-GAN generator
# --------------------------------------------------------------------------------------------------
class ResBlock1(nn.Module):
    """modules.py:223-317: three (dilated conv, conv) pairs with residuals, each pair = one ResUnitFn."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.convs1 = nn.ModuleList([
            EvtConv1d(channels, channels, kernel_size, dilation=d, padding=get_padding(kernel_size, d), weight_norm=True)
            for d in dilation])
        self.convs2 = nn.ModuleList([
            EvtConv1d(channels, channels, kernel_size, dilation=1, padding=get_padding(kernel_size, 1), weight_norm=True)
            for _ in dilation])

    def forward(self, x):
        for c1, c2 in zip(self.convs1, self.convs2):
            x = res_unit(x, c1, c2, LRELU_SLOPE)
        return x


class Generator(nn.Module, _ComputeDtype):
    """models.py:404-471."""

    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0):
        super().__init__()
        if str(resblock) != "1":
            raise L.EvtError("only ResBlock1 (configs/s2.json resblock='1') is implemented")
        self.num_kernels, self.num_upsamples = len(resblock_kernel_sizes), len(upsample_rates)
        self.conv_pre = EvtConv1d(initial_channel, upsample_initial_channel, 7, padding=3)
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            self.ups.append(EvtConv1d(upsample_initial_channel // (2 ** i), upsample_initial_channel // (2 ** (i + 1)),
                                      k, stride=u, padding=(k - u) // 2, transposed=True, weight_norm=True))
        self.resblocks = nn.ModuleList()
        ch = upsample_initial_channel
        for i in range(len(self.ups)):
            ch = upsample_initial_channel // (2 ** (i + 1))
            for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(ResBlock1(ch, k, tuple(d)))
        self.conv_post = EvtConv1d(ch, 1, 7, padding=3, bias=False)
        if gin_channels != 0:
            self.cond = pointwise(gin_channels, upsample_initial_channel)

    def forward(self, x, g=None):
        """x [B, T, inter] -> waveform [B, T*prod(up), 1]"""
        x = self.conv_pre(x.to(self.cd).contiguous())
        if g is not None:
            x = (x + self.cond(g).unsqueeze(1)).to(self.cd)
        for i in range(self.num_upsamples):
            x = self.ups[i](x, in_slope=LRELU_SLOPE)          # leaky_relu(0.1) fused on load
            blocks = [self.resblocks[i * self.num_kernels + j] for j in range(self.num_kernels)]
            if x.is_cuda:
                # narrow stages: step j of the three blocks is one grouped launch each way (hip/conv.py::ResStageFn)
                xs = res_stage(x, blocks, LRELU_SLOPE, 1.0 / self.num_kernels)
                if xs is not None:
                    x = xs
                    continue
            lane = None
            if x.is_cuda and len(blocks) == 3:
                from ..hip.disc import _On, dec_lane

                lane = dec_lane(x.device)
            if lane is not None:
                # the k = 11 block on the current stream, the k = 3 and k = 7 blocks (about the same work together) beside it
                main = torch.cuda.current_stream(x.device)
                lane.wait_stream(main)
                with _On(lane):
                    r0, r1 = blocks[0](x), blocks[1](x)
                x.record_stream(lane)
                r2 = blocks[2](x)
                main.wait_stream(lane)
                r0.record_stream(main)
                r1.record_stream(main)
                rs = [r0, r1, r2]
            else:
                rs = [b(x) for b in blocks]
            while len(rs) < 3:
                rs.append(None)
            x = Add3ScaleFn.apply(rs[0], rs[1], rs[2], 1.0 / self.num_kernels, self)
        # F.leaky_relu(x) at models.py:467 uses the DEFAULT slope 0.01, then conv_post, then tanh
        return self.conv_post(x, in_slope=0.01, out_act=L.ACT_TANH)


# --------------------------------------------------------------------------------------------------
# discriminators
# --------------------------------------------------------------------------------------------------
class DiscriminatorP(nn.Module, _ComputeDtype):
    """models.py:481-557.  [B, T] -> reflect-pad to a multiple of p -> p interleaved sequences of
    length T/p each; Conv2d((k,1),(s,1)) == Conv1d over each of the B*p sequences."""

    def __init__(self, period, kernel_size=5, stride=3, use_spectral_norm=False):
        super().__init__()
        if use_spectral_norm:
            raise L.EvtError("spectral_norm discriminators are not used by configs/s2.json")
        self.period = period
        chans = [1, 32, 128, 512, 1024, 1024]
        self.convs = nn.ModuleList([
            EvtConv1d(chans[i], chans[i + 1], kernel_size, stride=(stride if i < 4 else 1),
                      padding=get_padding(kernel_size, 1), weight_norm=True, kdims=2) for i in range(5)])
        self.conv_post = EvtConv1d(1024, 1, 3, padding=1, weight_norm=True, kdims=2)

    def prepare(self, x):
        """x [N, T] fp32 waveform -> the p interleaved sequences [N*p, T'/p, 1] in the compute dtype"""
        n, t = x.shape
        p = self.period
        if t % p != 0:
            x = F.pad(x.unsqueeze(1), (0, p - (t % p)), "reflect").squeeze(1)
            t = x.size(1)
        return x.view(n, t // p, p).transpose(1, 2).reshape(n * p, t // p, 1).to(self.cd).contiguous()

    def plan(self):
        """(conv slots, (output activation, slope) per conv) in forward order -- what hip/disc.py runs"""
        convs = list(self.convs) + [self.conv_post]
        return (tuple(c._slot for c in convs),
                tuple([(L.ACT_LRELU, LRELU_SLOPE)] * len(self.convs) + [(L.ACT_NONE, 1.0)]))

    def forward(self, x):
        """x [N, T] fp32 waveform -> (logits [N, p*H'] , fmaps list of [N*p, H_i, C_i])"""
        return self.forward_prepared(self.prepare(x))

    def forward_prepared(self, x):
        n = x.size(0) // self.period
        fmap = []
        for l in self.convs:
            x = l(x, out_act=L.ACT_LRELU, out_slope=LRELU_SLOPE)
            fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return x.view(n, -1), fmap


class DiscriminatorS(nn.Module, _ComputeDtype):
    """models.py:560-587."""

    def __init__(self, use_spectral_norm=False):
        super().__init__()
        if use_spectral_norm:
            raise L.EvtError("spectral_norm discriminators are not used by configs/s2.json")
        spec = [(1, 16, 15, 1, 1, 7), (16, 64, 41, 4, 4, 20), (64, 256, 41, 4, 16, 20), (256, 1024, 41, 4, 64, 20),
                (1024, 1024, 41, 4, 256, 20), (1024, 1024, 5, 1, 1, 2)]
        self.convs = nn.ModuleList([EvtConv1d(ci, co, k, stride=s, groups=g, padding=p, weight_norm=True)
                                    for ci, co, k, s, g, p in spec])
        self.conv_post = EvtConv1d(1024, 1, 3, padding=1, weight_norm=True)

    def prepare(self, x):
        return x.unsqueeze(-1).to(self.cd).contiguous()

    def plan(self):
        convs = list(self.convs) + [self.conv_post]
        return (tuple(c._slot for c in convs),
                tuple([(L.ACT_LRELU, LRELU_SLOPE)] * len(self.convs) + [(L.ACT_NONE, 1.0)]))

    def forward(self, x):
        return self.forward_prepared(self.prepare(x))

    def forward_prepared(self, x):
        fmap = []
        for l in self.convs:
            x = l(x, out_act=L.ACT_LRELU, out_slope=LRELU_SLOPE)
            fmap.append(x)
        x = self.conv_post(x)
        fmap.append(x)
        return x.view(x.size(0), -1), fmap


class MultiPeriodDiscriminator(nn.Module):
    """models.py:590-614.  `forward(y, y_hat)` keeps the reference's return structure
    (y_d_rs, y_d_gs, fmap_rs, fmap_gs); real and generated audio are batched through each
    sub-discriminator in ONE pass (same weights, twice the sequences)."""

    def __init__(self, use_spectral_norm=False):
        super().__init__()
        periods = [2, 3, 5, 7, 11]
        self.discriminators = nn.ModuleList([DiscriminatorS(use_spectral_norm)] +
                                            [DiscriminatorP(p, use_spectral_norm=use_spectral_norm) for p in periods])

    def periods(self):
        """input period of every sub-discriminator (1: DiscriminatorS takes the waveform itself)"""
        return tuple(getattr(d, "period", 1) for d in self.discriminators)

    def _prepared(self, y, y2=None):
        """the prepared input of every sub-discriminator for the waveform batch [y ; y2]: on the GPU one launch for all of
        them (hip/disc.py::mpd_fold), differentiable towards a single batch"""
        if y.is_cuda:
            from ..hip.disc import MPDFoldFn, mpd_fold

            cd = self.discriminators[0].cd
            if y2 is None and y.requires_grad:
                return list(MPDFoldFn.apply(y, self.periods(), cd))
            if y2 is None or not (y.requires_grad or y2.requires_grad):
                return mpd_fold(self.periods(), cd, y, y2)
        x = y if y2 is None else torch.cat([y.float(), y2.float()], dim=0)
        return [d.prepare(x.float()) for d in self.discriminators]

    def forward_single(self, y):
        """y [N, 1, T] or [N, T] -> (logits list, fmaps list-of-lists)"""
        y = y.reshape(y.size(0), -1)
        outs, fmaps = [], []
        for d, x in zip(self.discriminators, self._prepared(y)):
            o, f = d.forward_prepared(x)
            outs.append(o)
            fmaps.append(f)
        return outs, fmaps

    def generator_losses(self, y, y_hat):
        """The discriminators' part of the generator step (sovits.py:509-516): (generator_loss, feature_loss, generated
        logits per sub-discriminator) with gradients towards y_hat only -- one autograd node, real and generated audio
        batched through every convolution (hip/disc.py)."""
        from ..hip.disc import MPDGenLossFn

        n = y.size(0)
        plan = tuple(d.plan() for d in self.discriminators)
        if any(s is None for slots, _ in plan for s in slots):
            raise L.EvtError("MultiPeriodDiscriminator used before WeightBank.attach(); there is no eager fallback")
        anchor = plan[0][0][0].bank.anchor
        out = MPDGenLossFn.apply(anchor, plan, self.periods(), y_hat.reshape(n, -1).contiguous(),
                                 y.reshape(n, -1).detach().contiguous())
        return out[0], out[1], list(out[2:])

    def forward_batched(self, y, y_hat):
        """the D step's pass: logits of every sub-discriminator over [real ; generated] as ONE tensor each ([2n, -1]; the
        first n rows are the real audio) -- for losses.discriminator_loss_batched, which needs no slices"""
        n = y.size(0)
        xs = self._prepared(y.reshape(n, -1), y_hat.reshape(n, -1))
        if not y.is_cuda:
            return [d.forward_prepared(x)[0] for d, x in zip(self.discriminators, xs)]
        # the sub-discriminators are independent: dealt onto the current stream and a side stream (hip/disc.py, EVT_MPD_STREAMS);
        # autograd runs every node's backward on the stream of its forward and orders the streams itself
        from ..hip.disc import _On, _branches

        bank = self.discriminators[0].convs[0]._slot.bank if self.discriminators[0].convs[0]._slot is not None else None
        lanes, sides = _branches(y.device, len(self.discriminators)) if (bank is not None and bank.defer_n > 0) \
            else ([None] * len(self.discriminators), [])
        main = torch.cuda.current_stream(y.device) if sides else None
        for st in sides:
            st.wait_stream(main)
        outs = []
        for d, x, lane in zip(self.discriminators, xs, lanes):
            with _On(lane):
                o = d.forward_prepared(x)[0]
            if lane is not None:
                x.record_stream(lane)          # allocated on the current stream, read (and saved for backward) on the lane
                o.record_stream(main)
            outs.append(o)
        for st in sides:
            main.wait_stream(st)
        return outs

    def forward(self, y, y_hat):
        n = y.size(0)
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
        for d, x in zip(self.discriminators, self._prepared(y.reshape(n, -1), y_hat.reshape(n, -1))):
            o, f = d.forward_prepared(x)
            y_d_rs.append(o[:n])
            y_d_gs.append(o[n:])
            # sequences are ordered (item, period-phase): the first half of every fmap is the real audio
            fmap_rs.append([t[: t.size(0) // 2] for t in f])
            fmap_gs.append([t[t.size(0) // 2:] for t in f])
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs


# --------------------------------------------------------------------------------------------------
# SynthesizerTrn
# --------------------------------------------------------------------------------------------------
class SynthesizerTrn(nn.Module, _ComputeDtype):
    """Synthesizer for training, models.py:803-946.  Inputs/outputs use the reference's [B, C, T]
    layout; everything in between is channels-last."""

    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads,
                 n_layers, kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, n_speakers=0, gin_channels=0,
                 use_sdp=True, semantic_frame_rate=None, freeze_quantizer=None, version="v2", **kwargs):
        super().__init__()
        self.spec_channels, self.segment_size, self.inter_channels = spec_channels, segment_size, inter_channels
        self.gin_channels, self.version = gin_channels, version
        self.upsample_rates = upsample_rates           # read by the reference's TTS.run (tts.py:798)
        self.enc_p = TextEncoder(inter_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size,
                                 p_dropout, version=version)
        self.dec = Generator(inter_channels, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                             upsample_initial_channel, upsample_kernel_sizes, gin_channels=gin_channels)
        self.enc_q = PosteriorEncoder(spec_channels, inter_channels, hidden_channels, 5, 1, 16,
                                      gin_channels=gin_channels)
        self.flow = ResidualCouplingBlock(inter_channels, hidden_channels, 5, 1, 4, gin_channels=gin_channels)
        self.ref_enc = MelStyleEncoder(spec_channels if version == "v1" else 704, style_vector_dim=gin_channels)
        ssl_dim = 768
        assert semantic_frame_rate in ["25hz", "50hz"]
        self.semantic_frame_rate = semantic_frame_rate
        if semantic_frame_rate == "25hz":
            self.ssl_proj = nn.Conv1d(ssl_dim, ssl_dim, 2, stride=2)
        else:
            self.ssl_proj = nn.Conv1d(ssl_dim, ssl_dim, 1, stride=1)
        self.quantizer = ResidualVectorQuantizer(dimension=ssl_dim, n_q=1, bins=1024)
        self.freeze_quantizer = freeze_quantizer
        self.split_backward = False      # set by a data-parallel S2Engine (see forward)
        self._cut = None
        self._cut2 = None

    def _rvq(self):
        """the fp32 images + launches of the frozen quantizer path (hip/frontend.py::RvqEncoder), built on first use"""
        enc = getattr(self, "_rvq_enc", None)
        cb = self.quantizer.vq.layers[0]._codebook
        if enc is None or enc.bank.device != cb.embed.device:
            k = 2 if self.semantic_frame_rate == "25hz" else 1
            enc = RvqEncoder(lambda: self.ssl_proj.weight, lambda: self.ssl_proj.bias, lambda: cb.embed,
                             cb.embed.size(1), cb.codebook_size, k, k, cb.embed.device)
            object.__setattr__(self, "_rvq_enc", enc)
        return enc

    def _quantize(self, ssl):
        """ssl [B, 768, T] fp32 (reference layout): ssl_proj (fp32, no grad reaches it: models.py:912-921) + code look-up
        + x2 nearest upsample -> (quantized fp32 [B, T', 768] channels-last, codes [1, B, T'/2])"""
        with torch.no_grad(), torch.autocast("cuda", enabled=False):
            return rvq_encode(self._rvq(), self.quantizer, ssl, 2 if self.semantic_frame_rate == "25hz" else 1)

    @torch.no_grad()
    def extract_latent(self, x):
        """ssl features [B, 768, T] -> semantic codes [B, n_q=1, T'] (models.py:1015-1018), the tokens the s1 stage
        learns to predict (6-name2semantic.tsv)"""
        _q, codes = self._quantize(x)
        return codes.transpose(0, 1)

    @torch.no_grad()
    def decode(self, codes, text, refer, noise_scale=0.5, speed=1, noise=None):
        """Inference: semantic codes [1, B, Tc] + phoneme ids [B, Tt] + reference spectrogram(s) [B, spec, Tr] (a list
        averages the style vectors) -> waveform [B, 1, T*hop] (models.py:974-1013): prior from the text/ssl encoder,
        z_p = m_p + noise*exp(logs_p)*noise_scale, the flow run in reverse, the HiFi-GAN generator over the whole
        sequence.  `noise` ([B, inter, >=T]) injects the prior draw for deterministic parity runs."""
        amp = self.cd in (torch.bfloat16, torch.float16)
        with torch.autocast("cuda", dtype=self.cd if amp else torch.bfloat16, enabled=amp):
            ges = []
            for r in (refer if isinstance(refer, (list, tuple)) else [refer]):
                r_cl = r.transpose(1, 2)
                r_cl = r_cl if self.version == "v1" else r_cl[..., :704]
                ges.append(self.ref_enc(r_cl, torch.ones(r_cl.size(0), r_cl.size(1), 1, device=r.device)))
            ge = torch.stack(ges, 0).mean(0)
            B, Tc = codes.size(1), codes.size(2)
            T = Tc * 2 if self.semantic_frame_rate == "25hz" else Tc
            y_lengths = torch.full((B,), T, dtype=torch.long, device=codes.device)
            text_lengths = torch.full((B,), text.size(-1), dtype=torch.long, device=text.device)
            quantized = self.quantizer.decode(codes)
            if self.semantic_frame_rate == "25hz":
                quantized = quantized.repeat_interleave(2, dim=1)
            y_mask = torch.ones(B, T, 1, device=codes.device)
            text_mask = torch.ones(B, text.size(1), 1, device=text.device)
            enc = self.enc_p(quantized, y_mask, text, text_mask, ge, y_lengths, text_lengths, speed=speed)
            if speed != 1:
                _x, m_p, logs_p, y_mask = enc
                y_lengths = torch.full((B,), y_mask.size(1), dtype=torch.long, device=codes.device)
            else:
                _x, m_p, logs_p = enc
            if noise is None:
                noise = torch.randn(B, m_p.size(2), m_p.size(1), device=m_p.device)
            eps = noise[:, :, :m_p.size(1)].transpose(1, 2).to(m_p.dtype)
            z_p = m_p + eps * torch.exp(logs_p) * noise_scale
            ym = y_mask.to(self.cd)
            z = self.flow(z_p, ym, g=ge, reverse=True, lens=y_lengths.to(torch.int32))
            o = self.dec(z * ym, g=ge)
        return o.transpose(1, 2)

    def forward(self, ssl, y, y_lengths, text, text_lengths, eps=None, ids_slice=None):
        """ssl [B, 768, T], y [B, spec, T] linear spectrogram, text [B, Tt] -> same tuple as the reference:
        (o [B,1,seg*hop], commit_loss, ids_slice, y_mask, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q), quantized).
        `eps` ([B, inter, T]) and `ids_slice` ([B]) inject the two random draws (models.py:358, commons.py:55)
        for deterministic parity runs."""
        dev = y.device
        T = y.size(2)
        y_mask = commons.sequence_mask(y_lengths, T).unsqueeze(-1).to(torch.float32)        # [B, T, 1]
        text_mask = commons.sequence_mask(text_lengths, text.size(1)).unsqueeze(-1).to(torch.float32)
        amp = self.cd in (torch.bfloat16, torch.float16)
        with torch.autocast("cuda", dtype=self.cd if amp else torch.bfloat16, enabled=amp):
            lens32 = y_lengths.to(torch.int32)
            # the spectrogram once as channels-last rows in the compute dtype, zero-padded to enc_q.pre's image width
            y_cl = ncl_to_nlc(y.float(), self.enc_q.pre.cin, self.cd)                          # [B, T, 1088]
            ref_in = y_cl[..., :self.spec_channels] if self.version == "v1" else y_cl[..., :704]
            split = self.split_backward and torch.is_grad_enabled()
            lane = None
            if y.is_cuda and not split:
                from ..hip.disc import _On, enc_lane

                lane = enc_lane(dev)
            if lane is not None and os.environ.get("EVT_QUANT_LANE", "1") != "0":
                # the frozen ssl projection + quantizer (two fp32 convolutions, 0.2 ms) feed the prior encoder only: on its
                # lane from the start, beside the style encoder
                main = torch.cuda.current_stream(dev)
                lane.wait_stream(main)
                with _On(lane):
                    quantized, _codes = self._quantize(ssl)
                ssl.record_stream(lane)
                ge = self.ref_enc(ref_in * y_mask.to(self.cd), y_mask, lens32)                 # [B, gin]
            else:
                ge = self.ref_enc(ref_in * y_mask.to(self.cd), y_mask, lens32)                 # [B, gin]
                quantized, _codes = self._quantize(ssl)
            # the prior encoder meets the rest of the model again only in the KL term: on a side stream (hip/disc.py,
            # EVT_ENC_STREAM) its small-grid launches run beside posterior encoder / flow / vocoder, forward and -- autograd
            # runs a node's backward on the stream of its forward -- backward.  Not in the data-parallel cut program, which
            # sequences the sub-models' backward passes itself.
            if lane is not None:
                main = torch.cuda.current_stream(dev)
                lane.wait_stream(main)
                with _On(lane):
                    x, m_p, logs_p = self.enc_p(quantized, y_mask, text, text_mask, ge, y_lengths, text_lengths)
                for t in (quantized, y_mask, text, text_mask, ge, y_lengths, text_lengths):
                    t.record_stream(lane)
            else:
                x, m_p, logs_p = self.enc_p(quantized, y_mask, text, text_mask, ge, y_lengths, text_lengths)
            eps_cl = eps.transpose(1, 2) if eps is not None else None
            ym = y_mask.to(self.cd)      # 0/1 mask in the compute dtype: the WN stacks stay in one dtype (no cast kernels)
            ge_fq = ge
            if split:
                # second cut of the data-parallel step (see below): posterior encoder and flow hang on their own copy of
                # the style vector and on detached prior statistics, so their backward ends at these leaves -- their
                # gradient ranges are complete, and can be reduced, before the prior encoder's backward has started
                ge_fq = ge.detach().requires_grad_(True)
                m_p_cut, logs_p_cut = m_p.detach().requires_grad_(True), logs_p.detach().requires_grad_(True)
                self._cut2 = ((ge, ge_fq), (m_p, m_p_cut), (logs_p, logs_p_cut))
                m_p, logs_p = m_p_cut, logs_p_cut
            z, m_q, logs_q = self.enc_q(y_cl, ym, g=ge_fq, eps=eps_cl, lens=lens32)
            z_p = self.flow(z, ym, g=ge_fq, lens=lens32)
            if ids_slice is None:
                z_slice, ids_slice = commons.rand_slice_segments(z, y_lengths, self.segment_size)
            else:
                z_slice = commons.slice_segments(z, ids_slice, self.segment_size)
            if split:
                # data-parallel step: the autograd graph is cut at the vocoder's inputs, so that the engine can run the
                # backward of `dec` (and of the discriminators above it) first, start reducing those gradients, and
                # continue into flow / posterior encoder, then prior / style encoder, from the saved cut gradients
                # (train/s2_engine.py)
                z_cut, ge_cut = z_slice.detach().requires_grad_(True), ge.detach().requires_grad_(True)
                self._cut = ((z_slice, z_cut), (ge, ge_cut))
                o = self.dec(z_cut, g=ge_cut)
            else:
                o = self.dec(z_slice, g=ge)                                                    # [B, seg*hop, 1]
            if lane is not None:
                main.wait_stream(lane)
                for t in (x, m_p, logs_p, quantized):
                    t.record_stream(main)
        commit_loss = torch.zeros((), device=dev)   # quantizer in eval mode: core_vq.py:311-316 adds nothing
        tr = lambda t: t.transpose(1, 2)
        y_mask_ncl = y_mask.transpose(1, 2)
        return (o.transpose(1, 2), commit_loss, ids_slice, y_mask_ncl, y_mask_ncl,
                (tr(z), tr(z_p), tr(m_p), tr(logs_p), tr(m_q), tr(logs_q)), tr(quantized))
