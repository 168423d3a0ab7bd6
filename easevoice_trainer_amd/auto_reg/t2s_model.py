"""s1 AR text->semantic GPT (training path), MI355X-native.

Same constructor config, parameter keys (295 state_dict entries) and arithmetic as the reference's
Text2SemanticDecoder.forward_old (src/easevoice/soundstorm/auto_reg/models/t2s_model.py:255-338,431-490,557-561),
TransformerEncoder / TransformerEncoderLayer (modules/transformer.py:106-339), MultiheadAttention
(modules/activation.py:17-428, patched_mha_with_cache.py:14-465), TokenEmbedding / SinePositionalEmbedding
(modules/embedding.py:8-81).  Differences in HOW:
  * the [B*16, L, L] float mask is never built: the flash-attention kernel evaluates the prefix-LM + padding rule from
    (x_len, x_lens, y_lens);
  * post-LN residuals, cross-entropy(sum) + top-3 accuracy are single fused HIP launches;
  * Linear layers run on the library's own MFMA GEMM kernels (hip/linear.py -> evt_gemm_bf16_*), from bf16 weight
    images that a LinearBank rebuilds once per optimiser step (no per-call weight casts, no vendor BLAS).
`forward` is the DPO branch (t2s_model.py:393-429), `forward_old` the plain one the default config trains with;
`infer_panel*` (KV-cache decoding, t2s_model.py:732-878) lives in t2s_infer.py.
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from ..hip import lib as L
from ..hip.enc import bump_rng, new_site, relu_dropout, res_drop_ln
from ..hip.linear import LinearBank, linear as _hip_linear
from .blocks import attn_block, ffn_block
from .ops import AddLayerNormFn, CrossEntropyRowsFn, CrossEntropySumFn, PrefixLMAttentionFn
from .utils import dpo_loss, make_reject_y


def linear(x, weight, bias=None, relu=False):
    """F.linear (+ relu) on the HIP GEMM kernels; the weight must be attached to a LinearBank (S1Engine does it).
    (A module-level name so that the CPU wiring tests can substitute it: tests/cpu_emu.py.)"""
    return _hip_linear(x, weight, bias, relu)


class TokenEmbedding(nn.Module):
    def __init__(self, embedding_dim, vocab_size, dropout=0.0):
        super().__init__()
        self.vocab_size, self.embedding_dim = vocab_size, embedding_dim
        self.dropout = nn.Dropout(p=dropout)
        self.word_embeddings = nn.Embedding(vocab_size, embedding_dim)

    @property
    def weight(self):
        return self.word_embeddings.weight

    def forward(self, x):
        return self.dropout(self.word_embeddings(x))


class SinePositionalEmbedding(nn.Module):
    """x * x_scale + alpha * PE (embedding.py:36-81); `pe` is a plain attribute there, so it is not in the state_dict."""

    def __init__(self, embedding_dim, dropout=0.0, scale=False, alpha=False):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.x_scale = math.sqrt(embedding_dim) if scale else 1.0
        self.alpha = nn.Parameter(torch.ones(1), requires_grad=alpha)
        self.dropout = nn.Dropout(p=dropout)
        self._pe = None

    def pe(self, length, device, dtype):
        if self._pe is None or self._pe.size(0) < length or self._pe.device != device:
            n = max(length, 4000)
            position = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
            div_term = torch.exp(torch.arange(0, self.embedding_dim, 2, dtype=torch.float32)
                                 * -(math.log(10000.0) / self.embedding_dim))
            pe = torch.zeros(n, self.embedding_dim)
            pe[:, 0::2] = torch.sin(position * div_term)
            pe[:, 1::2] = torch.cos(position * div_term)
            self._pe = pe.to(device)
        return self._pe[:length].to(dtype)

    def forward(self, x):
        out = x * self.x_scale + self.alpha.to(x.dtype) * self.pe(x.size(1), x.device, x.dtype).unsqueeze(0)
        return self.dropout(out)


class MultiheadAttention(nn.Module):
    """keys: in_proj_weight [3E, E], in_proj_bias [3E], out_proj.{weight,bias} (activation.py:112-150)"""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, x, x_lens, y_lens, x_len, seed):
        qkv = linear(x, self.in_proj_weight, self.in_proj_bias)
        p = self.dropout if self.training else 0.0
        o = PrefixLMAttentionFn.apply(qkv.contiguous(), x_lens, y_lens, x_len, self.num_heads, p, seed)
        return linear(o, self.out_proj.weight, self.out_proj.bias)


class LayerNorm(nn.Module):
    def __init__(self, normalized_shape, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))


class TransformerEncoderLayer(nn.Module):
    """post-LN block, transformer.py:186-339 with norm_first=False, activation relu"""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        self._sites = [new_site() for _ in range(3)]

    def forward(self, x, x_lens, y_lens, x_len, seed):
        p = self.dropout.p if self.training else 0.0
        if x.is_cuda:
            # each sub-block is ONE autograd node (auto_reg/blocks.py): dropout1 / dropout2 ride in the residual +
            # LayerNorm launch, the inner relu + dropout in linear1's store, its derivative and the residual gradient sums
            # in the backward GEMMs' epilogues (masks from the device-counter hash stream of hip/enc.py, one id per site)
            x = attn_block(x, self.self_attn, self.norm1, x_lens, y_lens, x_len, seed, p, self._sites[0])
            return ffn_block(x, self.linear1, self.linear2, self.norm2, p, self._sites[1], self._sites[2])
        sa = self.self_attn(x, x_lens, y_lens, x_len, seed)
        sa = self.dropout1(sa)
        x = AddLayerNormFn.apply(x, sa, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        h = linear(x, self.linear1.weight, self.linear1.bias, relu=True)       # relu = the GEMM's epilogue
        ff = self.dropout2(linear(self.dropout(h), self.linear2.weight, self.linear2.bias))
        return AddLayerNormFn.apply(x, ff, self.norm2.weight, self.norm2.bias, self.norm2.eps)


class TransformerEncoder(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, dropout, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout)
                                     for _ in range(num_layers)])

    grad_hook = None          # callable(block index): the gradient has reached the input of that block (train/s1_engine.py
    grad_hook_blocks = ()     # uses it to start reducing the finished part of the gradient arena); the blocks to watch

    def forward(self, x, x_lens, y_lens, x_len, seed):
        for i, layer in enumerate(self.layers):
            if i in self.grad_hook_blocks and x.requires_grad:
                x.register_hook(lambda g, i=i: (self.grad_hook(i) if self.grad_hook is not None else None, None)[1])
            x = layer(x, x_lens, y_lens, x_len, seed + 7919 * i)
        return x


def make_pad_mask(lengths, max_len=None):
    """True at padded positions (models/utils.py:16-41)"""
    n = int(max_len if max_len is not None else lengths.max())
    return torch.arange(n, device=lengths.device).unsqueeze(0) >= lengths.unsqueeze(1)


class Text2SemanticDecoder(nn.Module):
    def __init__(self, config, norm_first=False, top_k=3):
        super().__init__()
        m = config["model"]
        self.model_dim, self.embedding_dim, self.num_head = m["hidden_dim"], m["embedding_dim"], m["head"]
        self.num_layers, self.vocab_size, self.phoneme_vocab_size = m["n_layer"], m["vocab_size"], m["phoneme_vocab_size"]
        self.p_dropout, self.EOS, self.top_k = m["dropout"], m["EOS"], top_k
        if norm_first:
            raise L.EvtError("norm_first=True is not used by configs/gpt.yaml")
        assert self.EOS == self.vocab_size - 1
        self.bert_proj = nn.Linear(1024, self.embedding_dim)
        self.ar_text_embedding = TokenEmbedding(self.embedding_dim, self.phoneme_vocab_size, self.p_dropout)
        self.ar_text_position = SinePositionalEmbedding(self.embedding_dim, dropout=0.1, scale=False, alpha=True)
        self.ar_audio_embedding = TokenEmbedding(self.embedding_dim, self.vocab_size, self.p_dropout)
        self.ar_audio_position = SinePositionalEmbedding(self.embedding_dim, dropout=0.1, scale=False, alpha=True)
        # the reference hard-codes dropout=0.1 in the layers regardless of config model.dropout (t2s_model.py:290)
        self.h = TransformerEncoder(self.model_dim, self.num_head, self.model_dim * 4, 0.1, self.num_layers)
        self.ar_predict_layer = nn.Linear(self.model_dim, self.vocab_size, bias=False)
        self.cd = torch.float32
        self._seed = 0

    def dense_specs(self):
        """(name, weight, bias) of every Linear of the training forward, for hip/linear.LinearBank"""
        specs = [("bert_proj", self.bert_proj.weight, self.bert_proj.bias)]
        for i, l in enumerate(self.h.layers):
            specs += [(f"h.{i}.in_proj", l.self_attn.in_proj_weight, l.self_attn.in_proj_bias),
                      (f"h.{i}.out_proj", l.self_attn.out_proj.weight, l.self_attn.out_proj.bias),
                      (f"h.{i}.linear1", l.linear1.weight, l.linear1.bias),
                      (f"h.{i}.linear2", l.linear2.weight, l.linear2.bias)]
        specs.append(("ar_predict_layer", self.ar_predict_layer.weight, None))
        return specs

    def attach_bank(self, dtype, device):
        """prepared weight images for the HIP GEMMs; call after the parameters have reached their final storage"""
        self._bank = LinearBank(self.dense_specs(), dtype, device)
        return self._bank

    def pad_y_eos(self, y, y_mask_int, eos_id):
        targets = F.pad(y, (0, 1), value=0) + eos_id * F.pad(y_mask_int, (0, 1), value=1)
        return targets[:, :-1], targets[:, 1:]

    def _logits(self, x, x_lens, y, y_lens, bert_feature):
        """embeddings -> 24 post-LN blocks -> predict layer over the y positions; returns (logits [B, Ty, V], targets)"""
        cd = self.cd
        bank = getattr(self, "_bank", None)
        if bank is not None:
            bank.prepare()       # weight images of all dense layers: one launch, only when the weights changed
        xe = self.ar_text_embedding(x)
        bf = torch.empty((x.size(0), x.size(1), bert_feature.size(1)), dtype=cd, device=bert_feature.device)
        bf.copy_(bert_feature.transpose(1, 2))          # [B, 1024, Tx] -> [B, Tx, 1024] in the compute dtype, one pass
        xe = xe + linear(bf, self.bert_proj.weight, self.bert_proj.bias).to(xe.dtype)
        xe = self.ar_text_position(xe)
        y_mask_int = make_pad_mask(y_lens, y.size(1)).to(torch.int64)
        codes = y.to(torch.int64) * (1 - y_mask_int)
        y_in, targets = self.pad_y_eos(codes, y_mask_int, eos_id=self.EOS)
        x_len = x.size(1)
        y_pos = self.ar_audio_position(self.ar_audio_embedding(y_in))
        xy = torch.cat([xe, y_pos], dim=1).to(cd).contiguous()
        self._seed = (self._seed * 1664525 + 1013904223) & 0x7FFFFFFF
        if xy.is_cuda:
            bump_rng(xy.device)      # new dropout masks for the fused residual/LayerNorm and relu launches
        xy_dec = self.h(xy, x_lens.to(torch.int32).contiguous(), y_lens.to(torch.int32).contiguous(), x_len, self._seed)
        # [B, Ty, ld]: ld = 1152 on the HIP path (the 1025-entry vocabulary padded to the GEMM tile, zero columns)
        logits = linear(xy_dec[:, x_len:].contiguous(), self.ar_predict_layer.weight)
        return logits.reshape(-1, logits.size(-1)), targets

    def forward_old(self, x, x_lens, y, y_lens, bert_feature):
        """x phoneme ids [B, Tx], y semantic ids [B, Ty], bert_feature [B, 1024, Tx] -> (loss sum, top-3 acc)"""
        logits, targets = self._logits(x, x_lens, y, y_lens, bert_feature)
        loss, hits = CrossEntropySumFn.apply(logits, targets.reshape(-1), self.top_k, self.EOS, self.vocab_size)
        acc = hits[0].float() / hits[1].clamp(min=1).float()
        return loss, acc

    def forward(self, x, x_lens, y, y_lens, bert_feature):
        """DPO branch (t2s_model.py:393-429): cross-entropy(sum) of the chosen pass plus the reference-free DPO term
        between the summed target log-probabilities of the chosen sequences and of `make_reject_y`'s corrupted copies.
        As in the reference the log-probabilities are summed over every position of the padded batch (padded positions
        have target EOS).  One fused per-row cross-entropy launch per pass gives the summed loss, the per-sequence
        log-probabilities and the saved softmax-onehot for both gradients; the accuracy stays a device scalar."""
        reject_y, reject_y_lens = make_reject_y(y, y_lens)
        B = x.size(0)
        logits, targets = self._logits(x, x_lens, y, y_lens, bert_feature)
        row, hits = CrossEntropyRowsFn.apply(logits, targets.reshape(-1), self.top_k, self.EOS, self.vocab_size)
        r_logits, r_targets = self._logits(x, x_lens, reject_y, reject_y_lens, bert_feature)
        r_row, _ = CrossEntropyRowsFn.apply(r_logits, r_targets.reshape(-1), self.top_k, self.EOS, self.vocab_size)
        chosen_logps, rejected_logps = -row.view(B, -1).sum(-1), -r_row.view(B, -1).sum(-1)
        loss = row.sum() + dpo_loss(chosen_logps, rejected_logps, 0.2)
        acc = hits[0].float() / hits[1].clamp(min=1).float()
        return loss, acc

    # ---- inference (t2s_model.py:732-878): KV-cache decoding, see t2s_infer.py ----
    def _infer(self):
        if getattr(self, "_infer_front", None) is None:
            from .t2s_infer import T2SInfer
            object.__setattr__(self, "_infer_front", T2SInfer(self))
        return self._infer_front

    def infer_panel_naive(self, x, x_lens, prompts, bert_feature, top_k=-100, top_p=100, early_stop_num=-1,
                          temperature=1.0, repetition_penalty=1.35, **kwargs):
        return self._infer().infer_panel_naive(x, x_lens, prompts, bert_feature, top_k=top_k, top_p=top_p,
                                               early_stop_num=early_stop_num, temperature=temperature,
                                               repetition_penalty=repetition_penalty, **kwargs)

    infer_panel = infer_panel_naive

    def infer_panel_batch_infer(self, x, x_lens, prompts, bert_feature, **kwargs):
        return self._infer().infer_panel_batch_infer(x, x_lens, prompts, bert_feature, **kwargs)

    def infer_panel_naive_batched(self, x, x_lens, prompts, bert_feature, **kwargs):
        return self._infer().infer_panel_naive_batched(x, x_lens, prompts, bert_feature, **kwargs)
