"""CN-HuBERT content features on the MI355X library.

Reference: src/easevoice/feature_extractor/cnhubert.py:16-32 (`CNHubert`: transformers' HubertModel.from_pretrained) and
src/normalization/normalize.py:132-180 (`_name2go`: 32 kHz clip -> rescale -> 16 kHz -> `model.model(wav)["last_hidden_state"]`
-> transposed [1, 768, T] saved as 4-cnhubert/<name>.pt), which the reference pins to the CPU (normalize.py:58-60).

The model is transformers' HubertModel (chinese-hubert-base: "feat_extract_norm": "group", post-LN encoder,
do_stable_layer_norm = false) restated module for module with the SAME parameter names and shapes, so a checkpoint's
state_dict loads as is (`load_hf_state_dict`); channels-last [B, T, C] rows; every layer is a launch of the C ABI:
  feature extractor   seven strided convolutions (evt_conv1d_fwd: 1 -> 512 k10 s5, 512 -> 512 k3 s2 x4, k2 s2 x2, no bias),
                      GroupNorm(512, 512) + GELU after the first (evt_channel_norm_gelu_fwd), GELU after the others
  feature projection  LayerNorm(512) (evt_add_layernorm_fwd) -> Linear 512 -> 768 (1x1 convolution)
  positional conv     grouped convolution 768 -> 768, k = 128, 16 groups, padding 64 (weight-norm over the tap axis, folded
                      once at load time), bias + drop-last-frame + GELU in one launch (evt_gelu_rows_fwd), residual +
                      LayerNorm (evt_add_layernorm_fwd)
  12 encoder layers   packed q | k | v projection + attention core (evt_mha_fwd through hip/enc.py::rel_self_attention,
                      window None), out projection, LayerNorm(x + attn), Linear 768 -> 3072, GELU, Linear 3072 -> 768,
                      LayerNorm(x + ffn)
Inference only (`torch.no_grad`); dropout / SpecAugment masking are training-time and absent, as in `model.eval()`.
There is no torch fallback: a shape the kernels do not serve raises.
"""
import torch
from torch import nn

from ..hip import lib as L
from ..hip.conv import EvtConv1d
from ..hip.enc import new_site, rel_self_attention
from ..hip.feat import add_layernorm, channel_norm_gelu, gelu_rows
from ..module.attentions import linear_rows
from ..runtime import ModelRuntime

CONV_DIM = (512, 512, 512, 512, 512, 512, 512)
CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)


class _LN(nn.Module):
    """nn.LayerNorm's parameters (`weight`, `bias`); applied through evt_add_layernorm_fwd by the owner"""

    def __init__(self, channels, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))

    def forward(self, x, r=None):
        return add_layernorm(x, r, self.weight, self.bias, self.eps)


class _ConvLayer(nn.Module):
    """HubertGroupNormConvLayer (layer 0: `conv`, `layer_norm` = GroupNorm's weight / bias) / HubertNoLayerNormConvLayer"""

    def __init__(self, cin, cout, k, stride, group_norm):
        super().__init__()
        self.conv = EvtConv1d(cin, cout, k, stride=stride, bias=False)
        self.layer_norm = _LN(cout, 1e-5) if group_norm else None

    def forward(self, x):
        y = self.conv(x)
        if self.layer_norm is not None:
            return channel_norm_gelu(y, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        return gelu_rows(y)


class _FeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()
        dims = (1,) + CONV_DIM
        self.conv_layers = nn.ModuleList(_ConvLayer(dims[i], dims[i + 1], CONV_KERNEL[i], CONV_STRIDE[i], i == 0)
                                         for i in range(len(CONV_DIM)))

    def forward(self, x):
        for layer in self.conv_layers:
            x = layer(x)
        return x


class _FeatureProjection(nn.Module):
    def __init__(self, cin, hidden, eps):
        super().__init__()
        self.layer_norm = _LN(cin, eps)
        self.projection = linear_rows(cin, hidden)

    def forward(self, x):
        return self.projection(self.layer_norm(x))


class _PosConv(nn.Module):
    """HubertPositionalConvEmbedding: `conv` holds the FOLDED weight (weight_norm over dim 2 in the checkpoint: the pair
    parametrizations.weight.original0 [1, 1, k] / original1 [C, C/groups, k], or weight_g / weight_v in older files)"""

    def __init__(self, hidden, k, groups):
        super().__init__()
        self.conv = EvtConv1d(hidden, hidden, k, padding=k // 2, groups=groups, bias=True)
        self.k = k

    def forward(self, x):
        # the convolution without its bias epilogue; bias, the dropped last frame of an even kernel and GELU in one launch
        y = self.conv(x)
        return gelu_rows(y, None, y.size(1) - (1 if self.k % 2 == 0 else 0))


class _Attention(nn.Module):
    def __init__(self, hidden, heads):
        super().__init__()
        self.num_heads = heads
        self.k_proj, self.v_proj, self.q_proj = linear_rows(hidden, hidden), linear_rows(hidden, hidden), linear_rows(hidden, hidden)
        self.out_proj = linear_rows(hidden, hidden)
        self._site = new_site()

    def forward(self, x, lens):
        o = rel_self_attention(x, self.q_proj, self.k_proj, self.v_proj, None, None, lens, self.num_heads, None, 0.0,
                               self._site)
        return self.out_proj(o)


class _FeedForward(nn.Module):
    def __init__(self, hidden, inner):
        super().__init__()
        self.intermediate_dense = linear_rows(hidden, inner)
        self.output_dense = linear_rows(inner, hidden)

    def forward(self, x):
        return self.output_dense(gelu_rows(self.intermediate_dense(x)))


class _EncoderLayer(nn.Module):
    """HubertEncoderLayer (post-LN): x = LN(x + attn(x)); x = LN'(x + ffn(x))"""

    def __init__(self, hidden, heads, inner, eps):
        super().__init__()
        self.attention = _Attention(hidden, heads)
        self.layer_norm = _LN(hidden, eps)
        self.feed_forward = _FeedForward(hidden, inner)
        self.final_layer_norm = _LN(hidden, eps)

    def forward(self, x, lens):
        x = self.layer_norm(x, self.attention(x, lens))
        return self.final_layer_norm(x, self.feed_forward(x))


class _Encoder(nn.Module):
    def __init__(self, hidden, heads, inner, layers, eps, pos_k, pos_groups):
        super().__init__()
        self.pos_conv_embed = _PosConv(hidden, pos_k, pos_groups)
        self.layer_norm = _LN(hidden, eps)
        self.layers = nn.ModuleList(_EncoderLayer(hidden, heads, inner, eps) for _ in range(layers))

    def forward(self, x, lens):
        x = self.layer_norm(x, self.pos_conv_embed(x))
        for layer in self.layers:
            x = layer(x, lens)
        return x


class HubertModel(nn.Module):
    """transformers.HubertModel (base configuration) on the library; state_dict keys = the checkpoint's, except the
    positional convolution's weight-norm pair, which `load_hf_state_dict` folds"""

    def __init__(self, hidden=768, heads=12, inner=3072, layers=12, eps=1e-5, pos_k=128, pos_groups=16):
        super().__init__()
        self.masked_spec_embed = nn.Parameter(torch.zeros(hidden))      # training-time SpecAugment filler: unused in eval
        self.feature_extractor = _FeatureExtractor()
        self.feature_projection = _FeatureProjection(CONV_DIM[-1], hidden, eps)
        self.encoder = _Encoder(hidden, heads, inner, layers, eps, pos_k, pos_groups)

    def load_hf_state_dict(self, sd):
        sd = dict(sd)
        pre = "encoder.pos_conv_embed.conv."
        g = sd.pop(pre + "parametrizations.weight.original0", None)
        v = sd.pop(pre + "parametrizations.weight.original1", None)
        if g is None:
            g, v = sd.pop(pre + "weight_g", None), sd.pop(pre + "weight_v", None)
        if g is not None:
            # torch.nn.utils.weight_norm(conv, dim=2): w = g * v / ||v|| with the norm over all axes but the tap axis
            v = v.float()
            sd[pre + "weight"] = g.float() * v / v.norm(2, dim=(0, 1), keepdim=True)
        missing, unexpected = self.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
        if missing or unexpected:
            raise KeyError(f"HuBERT checkpoint does not match: missing {missing}, unexpected {unexpected}")
        return self

    @staticmethod
    def frames(n_samples):
        for k, s in zip(CONV_KERNEL, CONV_STRIDE):
            n_samples = (n_samples - k) // s + 1
        return n_samples

    def forward(self, wav):
        """wav [B, n] float (16 kHz) -> last_hidden_state [B, T, 768]; no padding mask: the reference extracts one clip
        at a time (normalize.py:158), every frame is live"""
        rt = self._rt
        x = wav.to(rt.device, rt.dtype).unsqueeze(-1).contiguous()                 # [B, n, 1] channels-last
        x = self.feature_projection(self.feature_extractor(x))
        lens = torch.full((x.size(0),), x.size(1), dtype=torch.int32, device=x.device)
        return self.encoder(x, lens)


class CNHubert:
    """src/easevoice/feature_extractor/cnhubert.py:16-32 on the GPU: `model(wav16k [n]) -> [1, 768, T]` float32 on the CPU,
    the tensor normalize.py:165,176 saves as 4-cnhubert/<name>.pt.  `weights`: a transformers HubertModel state_dict (or a
    directory / file holding one); None keeps the random initialisation (tests)."""

    def __init__(self, weights=None, device="cuda:0", dtype=torch.float32):
        net = HubertModel()
        if weights is not None:
            net.load_hf_state_dict(_read_state_dict(weights))
        net.eval()
        L.set_half(dtype)
        self.rt = ModelRuntime(net, dtype=dtype, device=device)
        self.rt.bank.weight_grads = False
        self.rt.prepare(force=True)
        net._rt = self.rt
        self.model, self.device, self.dtype = net, torch.device(device), dtype

    @torch.no_grad()
    def last_hidden_state(self, wav16k):
        L.set_half(self.dtype)
        wav = torch.as_tensor(wav16k, dtype=torch.float32)
        if wav.dim() == 1:
            wav = wav.unsqueeze(0)
        return self.model(wav)

    @torch.no_grad()
    def __call__(self, wav16k):
        return self.last_hidden_state(wav16k).transpose(1, 2).float().cpu()


def _read_state_dict(weights):
    import os

    if isinstance(weights, dict):
        return weights
    path = weights
    if os.path.isdir(path):
        for name in ("model.safetensors", "pytorch_model.bin"):
            if os.path.exists(os.path.join(path, name)):
                path = os.path.join(path, name)
                break
        else:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {weights}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=False)
