"""Chinese-RoBERTa phone-level text features on the MI355X library.

Reference: src/normalization/normalize.py:65-106 (`text` / `_get_bert_feature`): transformers' BertForMaskedLM
(chinese-roberta-wwm-ext-large: a BERT-large encoder) on the CPU, `hidden_states[-3:-2]` -- the output of encoder layer 22
of 24 -- without the [CLS] / [SEP] rows, every character's row repeated word2ph[i] times, transposed: [1024, n_phones],
saved as 3-bert/<name>.pt and read back by the s1 dataset (dataset.py:165-176: bert_feature).

The encoder is restated module for module with the checkpoint's parameter names (`bert.embeddings.*`,
`bert.encoder.layer.N.attention.self.{query,key,value}`, `.attention.output.{dense,LayerNorm}`, `.intermediate.dense`,
`.output.{dense,LayerNorm}`), so `load_state_dict(checkpoint, strict=False)` fills it; the masked-LM head (`cls.*`) and the
layers above the one that is read are not built.  Rows are [B, T, C]; per layer: three 1x1-convolution projections + the
attention core (hip/enc.py::rel_self_attention, window None, key padding by lengths), the output projection,
LayerNorm(x + attn) (evt_add_layernorm_fwd), Linear -> GELU (evt_gelu_rows_fwd) -> Linear, LayerNorm(x + ffn).  The
embedding sum (word + position + token type) is a gather and its LayerNorm one launch.  Inference only."""
import torch
from torch import nn

from ..hip import lib as L
from ..hip.enc import new_site, rel_self_attention
from ..hip.feat import add_layernorm, gelu_rows
from ..module.attentions import linear_rows
from ..runtime import ModelRuntime


class _LayerNorm(nn.Module):
    def __init__(self, channels, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))

    def forward(self, x, r=None):
        return add_layernorm(x, r, self.weight, self.bias, self.eps)


class _Embeddings(nn.Module):
    def __init__(self, vocab, hidden, max_pos, types, eps):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab, hidden)
        self.position_embeddings = nn.Embedding(max_pos, hidden)
        self.token_type_embeddings = nn.Embedding(types, hidden)
        self.LayerNorm = _LayerNorm(hidden, eps)

    def forward(self, input_ids, token_type_ids, cd):
        T = input_ids.size(1)
        pos = torch.arange(T, device=input_ids.device)
        w = self.word_embeddings(input_ids)
        r = self.position_embeddings(pos).unsqueeze(0) + self.token_type_embeddings(token_type_ids)
        return self.LayerNorm(w.to(cd).contiguous(), r.to(cd).expand_as(w).contiguous())


class _SelfAttention(nn.Module):
    def __init__(self, hidden, heads):
        super().__init__()
        self.heads = heads
        self.query, self.key, self.value = linear_rows(hidden, hidden), linear_rows(hidden, hidden), linear_rows(hidden, hidden)
        self._site = new_site()

    def forward(self, x, lens):
        return rel_self_attention(x, self.query, self.key, self.value, None, None, lens, self.heads, None, 0.0, self._site)


class _SelfOutput(nn.Module):
    def __init__(self, cin, hidden, eps):
        super().__init__()
        self.dense = linear_rows(cin, hidden)
        self.LayerNorm = _LayerNorm(hidden, eps)

    def forward(self, h, x):
        return self.LayerNorm(x, self.dense(h))


class _Attention(nn.Module):
    def __init__(self, hidden, heads, eps):
        super().__init__()
        self.self = _SelfAttention(hidden, heads)
        self.output = _SelfOutput(hidden, hidden, eps)

    def forward(self, x, lens):
        return self.output(self.self(x, lens), x)


class _Intermediate(nn.Module):
    def __init__(self, hidden, inner):
        super().__init__()
        self.dense = linear_rows(hidden, inner)

    def forward(self, x):
        return gelu_rows(self.dense(x))


class _Layer(nn.Module):
    def __init__(self, hidden, heads, inner, eps):
        super().__init__()
        self.attention = _Attention(hidden, heads, eps)
        self.intermediate = _Intermediate(hidden, inner)
        self.output = _SelfOutput(inner, hidden, eps)

    def forward(self, x, lens):
        x = self.attention(x, lens)
        return self.output(self.intermediate(x), x)


class _Encoder(nn.Module):
    def __init__(self, n, hidden, heads, inner, eps):
        super().__init__()
        self.layer = nn.ModuleList(_Layer(hidden, heads, inner, eps) for _ in range(n))


class _Bert(nn.Module):
    def __init__(self, vocab, hidden, heads, inner, n, max_pos, types, eps):
        super().__init__()
        self.embeddings = _Embeddings(vocab, hidden, max_pos, types, eps)
        self.encoder = _Encoder(n, hidden, heads, inner, eps)


class BertEncoderStack(nn.Module):
    """`bert.*` of a BertForMaskedLM checkpoint up to the layer whose output is read (`read_layer` of `num_layers`;
    hidden_states[-3] of a 24-layer model = the output of layer 22)"""

    def __init__(self, vocab=21128, hidden=1024, heads=16, inner=4096, num_layers=24, read_layer=22, max_pos=512, types=2,
                 eps=1e-12):
        super().__init__()
        if not 0 <= read_layer <= num_layers:
            raise ValueError("read_layer out of range")
        self.bert = _Bert(vocab, hidden, heads, inner, read_layer, max_pos, types, eps)
        self.num_layers, self.read_layer = num_layers, read_layer
        self.cd = torch.float32

    def load_hf_state_dict(self, sd):
        own = self.state_dict()
        picked = {k: v.float() for k, v in sd.items() if k in own}
        missing = [k for k in own if k not in picked]
        if missing:
            raise KeyError(f"BERT checkpoint does not match: missing {missing[:8]} ...")
        self.load_state_dict(picked)
        return self

    def forward(self, input_ids, attention_mask=None, token_type_ids=None):
        """-> hidden_states[read_layer] [B, T, hidden]; attention_mask [B, T] of 1 / 0 with the live tokens first (the
        tokenizer's right padding): keys beyond a row's length are excluded"""
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        B, T = input_ids.shape
        lens = (attention_mask.sum(1) if attention_mask is not None else torch.full((B,), T, device=input_ids.device)).to(torch.int32)
        x = self.bert.embeddings(input_ids, token_type_ids, self.cd)
        for layer in self.bert.encoder.layer:
            x = layer(x, lens)
        return x


class BertFeatures:
    """`_get_bert_feature` (normalize.py:88-106) on the GPU.  `weights`: a BertForMaskedLM / BertModel state_dict (keys
    `bert.*`), or a directory / file holding one; None keeps the random initialisation (tests); `config`: overrides of the
    BERT-large dimensions."""

    def __init__(self, weights=None, device="cuda:0", dtype=torch.float32, **config):
        from .cnhubert import _read_state_dict

        net = BertEncoderStack(**config)
        if weights is not None:
            net.load_hf_state_dict(_read_state_dict(weights))
        net.eval()
        L.set_half(dtype)
        self.rt = ModelRuntime(net, dtype=dtype, device=device)
        self.rt.bank.weight_grads = False
        self.rt.prepare(force=True)
        self.model, self.device, self.dtype = net, torch.device(device), dtype

    @torch.no_grad()
    def hidden(self, input_ids, attention_mask=None, token_type_ids=None):
        L.set_half(self.dtype)
        mv = lambda t: None if t is None else torch.as_tensor(t).to(self.device)
        return self.model(mv(input_ids), mv(attention_mask), mv(token_type_ids))

    @torch.no_grad()
    def phone_level_feature(self, input_ids, word2ph, token_type_ids=None):
        """one sentence: input_ids [1, n + 2] ([CLS] chars [SEP]), word2ph: n counts -> [hidden, sum(word2ph)] float32 on the
        CPU (normalize.py:93-105: res[1:-1], row i repeated word2ph[i] times, transposed)"""
        res = self.hidden(input_ids, None, token_type_ids)[0, 1:-1].float()
        if len(word2ph) != res.size(0):
            raise ValueError("text and word2ph not match")         # the reference's failure message, normalize.py:96-97
        rep = torch.as_tensor(list(word2ph), device=res.device)
        return torch.repeat_interleave(res, rep, dim=0).T.contiguous().cpu()
