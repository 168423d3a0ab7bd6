"""TEST INFRASTRUCTURE ONLY — CPU fp32 restatement of the reference's s1 (text->semantic GPT) micro-step.

Follows Text2SemanticDecoder.forward_old  src/easevoice/soundstorm/auto_reg/models/t2s_model.py:431-490 (incl. the
materialised float attention mask :456-479 and pad_y_eos :557-561), TransformerEncoderLayer (post-LN, relu)
modules/transformer.py:266-339, multi_head_attention_forward_patched modules/patched_mha_with_cache.py:242-460,
SinePositionalEmbedding modules/embedding.py:36-81, and ScaledAdam modules/optim.py:206-622 (per-tensor form).
The DPO branch (t2s_model.py:393-429, models/utils.py:160-228) is forward_dpo / make_reject_y; KV-cache decoding
(t2s_model.py:762-863, models/utils.py:118-160) is infer_panel_naive / logits_to_probs, pinned by s1_infer.pt.
PINNED against tests/golden/s1_small.pt and s1_dpo.pt (reference's own modules) by tests/test_oracle_cpu.py."""
import math

import torch
import torch.nn.functional as F


def sine_pe(n, dim):
    pos = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe = torch.zeros(n, dim)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def prefix_lm_mask(x_lens, y_lens, x_len, y_len):
    """bool [B, L, L], True = masked (t2s_model.py:456-479)"""
    B = x_lens.numel()
    pad = torch.cat([torch.arange(x_len)[None] >= x_lens[:, None], torch.arange(y_len)[None] >= y_lens[:, None]], 1)
    x_rows = F.pad(torch.zeros(x_len, x_len, dtype=torch.bool), (0, y_len), value=True)
    y_rows = F.pad(torch.triu(torch.ones(y_len, y_len, dtype=torch.bool), diagonal=1), (x_len, 0), value=False)
    base = torch.cat([x_rows, y_rows], 0)
    return base[None].expand(B, -1, -1) | pad[:, None, :]


def attention(qkv, mask_bool, n_head):
    """qkv [B, L, 3E] -> [B, L, E]; mask_bool [B, L, L] True = masked"""
    B, L_, E3 = qkv.shape
    E = E3 // 3
    d = E // n_head
    q, k, v = qkv.split(E, dim=-1)
    q = q.view(B, L_, n_head, d).transpose(1, 2)
    k = k.view(B, L_, n_head, d).transpose(1, 2)
    v = v.view(B, L_, n_head, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(d)
    s = s.masked_fill(mask_bool[:, None], float("-inf"))
    return torch.matmul(F.softmax(s, dim=-1), v).transpose(1, 2).reshape(B, L_, E)


def forward_old(sd, cfg, x, x_lens, y, y_lens, bert_feature):
    m = cfg["model"]
    E, H, nl, EOS, V = m["hidden_dim"], m["head"], m["n_layer"], m["EOS"], m["vocab_size"]
    xe = F.embedding(x, sd["ar_text_embedding.word_embeddings.weight"])
    xe = xe + F.linear(bert_feature.transpose(1, 2), sd["bert_proj.weight"], sd["bert_proj.bias"])
    xe = xe + sd["ar_text_position.alpha"] * sine_pe(x.size(1), E)[None]
    y_mask = (torch.arange(y.size(1))[None] >= y_lens[:, None]).to(torch.int64)
    codes = y.to(torch.int64) * (1 - y_mask)
    t_full = F.pad(codes, (0, 1), value=0) + EOS * F.pad(y_mask, (0, 1), value=1)
    y_in, targets = t_full[:, :-1], t_full[:, 1:]
    ye = F.embedding(y_in, sd["ar_audio_embedding.word_embeddings.weight"])
    ye = ye + sd["ar_audio_position.alpha"] * sine_pe(y.size(1), E)[None]
    h = torch.cat([xe, ye], dim=1)
    x_len, y_len = x.size(1), y.size(1)
    mask = prefix_lm_mask(x_lens, y_lens, x_len, y_len)
    for i in range(nl):
        p = f"h.layers.{i}."
        qkv = F.linear(h, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
        sa = F.linear(attention(qkv, mask, H), sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(h + sa, (E,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        ff = F.linear(F.relu(F.linear(h, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"],
                      sd[p + "linear2.bias"])
        h = F.layer_norm(h + ff, (E,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    logits = F.linear(h[:, x_len:], sd["ar_predict_layer.weight"]).permute(0, 2, 1)
    loss = F.cross_entropy(logits, targets, reduction="sum")
    top3 = logits.detach().topk(3, dim=1).indices
    keep = targets != EOS
    acc = ((top3 == targets.unsqueeze(1)).any(dim=1) & keep).sum().float() / keep.sum().clamp(min=1).float()
    return loss, acc, logits


def make_reject_y(y, y_lens):
    """rejected sequences of the DPO branch, models/utils.py:185-228.  Restated with its quirks: the branch selector is
    `randint(0, 1)` (always 0 -> always the repeat rule, the drop rule is dead code), the two cut points are drawn over
    the PADDED row length and the row is used including its padding, and the returned length is the whole new row.
    Draw order on torch's global CPU generator: per item one randint(0,1,(1,)) then one randint(0,len,(2,))."""
    rows, lens = [], []
    for b in range(len(y_lens)):
        torch.randint(0, 1, size=(1,))
        i0, i1 = torch.randint(0, len(y[b]), size=(2,)).sort()[0].tolist()
        rows.append(torch.cat([y[b][:i0], y[b][i0:i1], y[b][i0:i1], y[b][i1:]]))
        lens.append(len(rows[-1]))
    width = max(lens)
    out = torch.stack([F.pad(r, (0, width - len(r))) for r in rows], 0)
    return out, torch.tensor(lens)


def forward_dpo(sd, cfg, x, x_lens, y, y_lens, bert_feature, reject=None):
    """Text2SemanticDecoder.forward, t2s_model.py:393-429: cross-entropy(sum) of the chosen pass + the reference-free
    DPO term (beta 0.2, models/utils.py:160-183) between the summed target log-probabilities of the chosen and the
    rejected pass -- summed over every position, padded ones included (their target is EOS)."""
    reject_y, reject_lens = reject if reject is not None else make_reject_y(y, y_lens)
    loss_1, acc, logits = forward_old(sd, cfg, x, x_lens, y, y_lens, bert_feature)
    _, _, rlogits = forward_old(sd, cfg, x, x_lens, reject_y, reject_lens, bert_feature)
    EOS = cfg["model"]["EOS"]

    def targets_of(yy, ll):
        mask = (torch.arange(yy.size(1))[None] >= ll[:, None]).to(torch.int64)
        full = F.pad(yy.to(torch.int64) * (1 - mask), (0, 1), value=0) + EOS * F.pad(mask, (0, 1), value=1)
        return full[:, 1:]

    def logps(lg, tg):     # lg [B, V, T]
        return torch.gather(lg.log_softmax(1), 1, tg.unsqueeze(1)).squeeze(1).sum(-1)

    chosen, rejected = logps(logits, targets_of(y, y_lens)), logps(rlogits, targets_of(reject_y, reject_lens))
    loss_2 = (-F.logsigmoid(0.2 * (chosen - rejected))).mean()
    return loss_1 + loss_2, acc, (chosen, rejected, loss_2)


def logits_to_probs(logits, previous_tokens, temperature=1.0, top_k=None, top_p=None, repetition_penalty=1.0):
    """models/utils.py:125-160, on a copy: repetition penalty on the already generated ids, nucleus cut on the
    un-tempered logits (first sorted entry always kept), temperature, top-k pivot (ties kept), softmax"""
    logits = logits.clone()
    if previous_tokens is not None and repetition_penalty != 1.0 and previous_tokens.numel() > 0:
        prev = previous_tokens.long()
        score = torch.gather(logits, 1, prev)
        score = torch.where(score < 0, score * repetition_penalty, score / repetition_penalty)
        logits.scatter_(1, prev, score)
    if top_p is not None and top_p < 1.0:
        srt, order = torch.sort(logits, descending=True)
        remove = torch.cumsum(F.softmax(srt, dim=-1), dim=-1) > top_p
        remove[:, 0] = False
        logits = logits.masked_fill(remove.scatter(1, order, remove), -float("inf"))
    logits = logits / max(temperature, 1e-5)
    if top_k is not None:
        pivot = torch.topk(logits, min(top_k, logits.size(-1)))[0][:, -1:]
        logits = torch.where(logits < pivot, -float("inf"), logits)
    return F.softmax(logits, dim=-1)


def infer_panel_naive(sd, cfg, x, prompts, bert_feature, q, top_k=15, top_p=1, early_stop_num=-1, temperature=1.0,
                      repetition_penalty=1.35, max_steps=1500, no_eos_steps=11, info=None):
    """KV-cache decoding, t2s_model.py:762-863, for batch 1 on plain tensors.  `q` [steps, V] is the exponential noise
    of multinomial_sample_one_no_sync (models/utils.py:118-122): token = argmax(probs / q[step]).  Returns (y without
    its last token, idx-1 or 0 when there was no prompt, list of the raw logits per step)."""
    m = cfg["model"]
    E, H, nl, EOS = m["hidden_dim"], m["head"], m["n_layer"], m["EOS"]
    d = E // H
    pe = sine_pe(4000, E)
    xe = F.embedding(x, sd["ar_text_embedding.word_embeddings.weight"])
    xe = xe + F.linear(bert_feature.transpose(1, 2), sd["bert_proj.weight"], sd["bert_proj.bias"])
    xe = xe + sd["ar_text_position.alpha"] * pe[None, :x.size(1)]
    x_len = x.size(1)
    if prompts is not None:
        y = prompts
        ye = F.embedding(y, sd["ar_audio_embedding.word_embeddings.weight"])
        h = torch.cat([xe, ye + sd["ar_audio_position.alpha"] * pe[None, :y.size(1)]], 1)
    else:
        y = torch.zeros(1, 0, dtype=torch.long)
        h = xe
    y_len = prefix_len = y.size(1)
    mask = prefix_lm_mask(torch.tensor([x_len]), torch.tensor([y_len]), x_len, y_len)
    k_cache, v_cache = [None] * nl, [None] * nl
    all_logits = []
    idx = 0
    for idx in range(max_steps):
        for i in range(nl):
            p = f"h.layers.{i}."
            qkv = F.linear(h, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
            if idx == 0:
                k_cache[i], v_cache[i] = qkv[..., E:2 * E], qkv[..., 2 * E:]
                a = attention(qkv, mask, H)
            else:
                k_cache[i] = torch.cat([k_cache[i], qkv[..., E:2 * E]], 1)
                v_cache[i] = torch.cat([v_cache[i], qkv[..., 2 * E:]], 1)
                L_ = k_cache[i].size(1)
                qh = qkv[..., :E].view(1, 1, H, d).transpose(1, 2)
                kh = k_cache[i].view(1, L_, H, d).transpose(1, 2)
                vh = v_cache[i].view(1, L_, H, d).transpose(1, 2)
                w = F.softmax(torch.matmul(qh, kh.transpose(-2, -1)) / math.sqrt(d), dim=-1)
                a = torch.matmul(w, vh).transpose(1, 2).reshape(1, 1, E)
            sa = F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
            h = F.layer_norm(h + sa, (E,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
            ff = F.linear(F.relu(F.linear(h, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                          sd[p + "linear2.weight"], sd[p + "linear2.bias"])
            h = F.layer_norm(h + ff, (E,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
        logits = F.linear(h[:, -1], sd["ar_predict_layer.weight"])
        if idx < no_eos_steps:
            logits = logits[:, :-1]
        all_logits.append(logits.clone())
        probs = logits_to_probs(logits, y, temperature, top_k, top_p, repetition_penalty)
        tok = torch.argmax(probs / q[idx, :probs.size(-1)], dim=-1, keepdim=True)
        y = torch.cat([y, tok], 1)
        stop = early_stop_num != -1 and (y.size(1) - prefix_len) > early_stop_num
        eos = int(torch.argmax(logits, dim=-1)[0]) == EOS or int(tok[0, 0]) == EOS
        if info is not None:
            info["eos"] = eos
        if stop or eos:
            break
        ye = F.embedding(y[:, -1:], sd["ar_audio_embedding.word_embeddings.weight"])
        h = ye + sd["ar_audio_position.alpha"] * pe[None, y_len + idx:y_len + idx + 1]
    return y[:, :-1], (0 if prompts is None else idx - 1), all_logits


def infer_panel_batch_infer(sd, cfg, xs, prompts, berts, q, **kw):
    """infer_panel_batch_infer, t2s_model.py:563-730 (the TTS default, parallel_infer=True), restated through the
    independence of the batch rows: padded text positions are masked for every query and never read back, and a finished
    row is only removed from the batch, so row b decodes exactly like a batch of one on its unpadded text with its own
    noise q[:, b].  Differences to infer_panel_naive that are kept: the EOS column is dropped at step 0 only
    (:661-663), the returned index is idx-1 for an EOS stop but idx for the early stop (:687, :705)."""
    ys, idxs = [], []
    for b, (x, bert) in enumerate(zip(xs, berts)):
        info = {}
        y, idx_m1, _ = infer_panel_naive(sd, cfg, x[None], prompts[b:b + 1], bert[None], q[:, b], no_eos_steps=1, info=info,
                                         **kw)
        ys.append(y[0])
        idxs.append(idx_m1 if info["eos"] else idx_m1 + 1)
    return ys, idxs


class ScaledAdamRef:
    """per-tensor restatement of optim.py:206-622 on plain tensors"""

    def __init__(self, params, lr=0.01, betas=(0.9, 0.95), clipping_scale=2.0, clipping_update_period=1000,
                 scalar_lr_scale=0.1, eps=1e-8, param_min_rms=1e-5, param_max_rms=3.0, scalar_max=10.0,
                 size_update_period=4):
        self.p = params          # dict name -> tensor (updated in place)
        self.lr, self.betas, self.cs, self.period = lr, betas, clipping_scale, clipping_update_period
        self.slr, self.eps, self.minr, self.maxr, self.smax, self.P = scalar_lr_scale, eps, param_min_rms, param_max_rms, scalar_max, size_update_period
        self.step_n = 0
        self.st = {}
        self.norms = torch.zeros(clipping_update_period)
        self.thr = None

    def step(self, grads):
        b1, b2 = self.betas
        step = self.step_n
        if step == 0:
            for k, p in self.p.items():
                self.st[k] = dict(delta=torch.zeros_like(p), v=torch.zeros_like(p),
                                  rms=(p ** 2).mean().sqrt() if p.numel() > 1 else None, sq=torch.zeros(()),
                                  sg=torch.zeros(self.P))
        clip = 1.0
        if step > 0:
            tot = sum(((grads[k] * self.st[k]["rms"]) ** 2).sum() if p.numel() > 1 else (grads[k] ** 2).sum()
                      for k, p in self.p.items())
            tot_norm = tot.sqrt()
            self.norms[step % self.period] = tot_norm
            if step % self.period == 0:
                srt = self.norms.sort()[0]
                self.thr = self.cs * srt[min(self.period - 1, (self.period // 4) * 2)].item()
            if step >= self.period and self.thr is not None:
                clip = min(1.0, (self.thr / (tot_norm + 1e-20)).item())
        for k, p in self.p.items():
            # NOTE (reference behaviour, optim.py:462-464 vs :574,:609): the clipped gradient is a LOCAL variable of
            # _step_one_batch and only feeds scale_grads; _step / _step_scalar re-read the unclipped p.grad.
            s, g = self.st[k], grads[k]
            s["delta"].mul_(b1)
            if p.numel() > 1:
                s["sg"][step % self.P] = (p * (g * clip)).sum()
                if step % self.P == self.P - 1:
                    s["rms"] = (p ** 2).mean().sqrt()
                    if step > 0:
                        size_lr = self.lr * self.slr
                        b2c = b2 ** self.P
                        s["sq"] = s["sq"] * b2c + (1 - b2c) * (s["sg"] ** 2).mean()
                        bc2 = 1 - b2c ** ((step + 1) // self.P)
                        sstep = -size_lr * (bc2 ** 0.5) * s["sg"].sum() / (s["sq"].sqrt() + self.eps)
                        if s["rms"] < self.minr:
                            sstep = torch.zeros(())
                        if s["rms"] > self.maxr:
                            sstep = torch.tensor(-size_lr * self.P)
                        s["delta"].add_(p * sstep, alpha=1 - b1)
                s["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
                bc2 = 1 - b2 ** (step + 1)
                v = s["v"] / bc2 if bc2 < 0.99 else s["v"]
                alpha = -self.lr * (1 - b1) * s["rms"].clamp(min=self.minr)
                s["delta"].add_(g / (v.sqrt() + self.eps) * alpha)
                p.add_(s["delta"])
            else:
                s["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
                bc2 = 1 - b2 ** (step + 1)
                denom = (s["v"] / bc2).sqrt() + self.eps
                s["delta"].add_(g / denom, alpha=-self.lr * self.slr * (1 - b1))
                p.clamp_(min=-self.smax, max=self.smax)
                p.add_(s["delta"])
        self.step_n += 1
