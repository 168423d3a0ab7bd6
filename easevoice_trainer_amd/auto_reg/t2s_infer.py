"""KV-cache decoding of the s1 model: Text2SemanticDecoder.infer_panel_naive,
src/easevoice/soundstorm/auto_reg/models/t2s_model.py:762-863, with T2SBlock.process_prompt / decode_next_token
(:124-222) and sample() (models/utils.py:118-171).

Same token sequence as the reference for the same sampling noise; a different execution plan:
  * prompt pass: the training kernels without gradients (packed qkv GEMM, analytic prefix-LM flash attention, fused
    residual+LayerNorm); its keys/values are copied once into a preallocated cache [layers][B][Lmax][E];
  * token steps: five launches per block (csrc/s1_decode.hip: in-projection, cache attention, out-projection, two for
    the MLP) + logits + one for sampling / embedding / counters, all reading their per-step state (cache length, step
    index, token count) from device memory, captured once into a HIP graph and replayed per token.  The host reads the stop flag every `poll` steps; tokens decoded past the stop are discarded.
The reference reads two device scalars per token (the EOS tests of :846) and reallocates every cache tensor per token."""
import ctypes as C
import os

import torch
from torch.nn import functional as F

from ..hip import lib as L
from ..hip.linear import LinearBank, gemm_fwd
from .ops import AddLayerNormFn, PrefixLMAttentionFn

MAX_STEPS = 1500          # t2s_model.py:822
NO_EOS_STEPS = 11         # t2s_model.py:833


class _Weights:
    """per-block matrices in the streaming dtype (fp32 parameters as they are, or one-time bf16 copies), vectors fp32"""

    def __init__(self, model, dtype):
        self.stamp = self.stamp_of(model)
        cv = (lambda t: t.detach().contiguous()) if dtype == torch.float32 else (lambda t: t.detach().to(dtype).contiguous())
        f32 = lambda t: t.detach().float().contiguous()
        self.layers = []
        for lyr in model.h.layers:
            a = lyr.self_attn
            self.layers.append(dict(
                wqkv=cv(a.in_proj_weight), bqkv=f32(a.in_proj_bias), wo=cv(a.out_proj.weight), bo=f32(a.out_proj.bias),
                w1=cv(lyr.linear1.weight), b1=f32(lyr.linear1.bias), w2=cv(lyr.linear2.weight), b2=f32(lyr.linear2.bias),
                g1=f32(lyr.norm1.weight), be1=f32(lyr.norm1.bias), g2=f32(lyr.norm2.weight), be2=f32(lyr.norm2.bias),
                eps1=float(lyr.norm1.eps), eps2=float(lyr.norm2.eps)))
        self.wpred = cv(model.ar_predict_layer.weight)
        self.emb = f32(model.ar_audio_embedding.word_embeddings.weight)
        self.alpha = f32(model.ar_audio_position.alpha)

    @staticmethod
    def stamp_of(model):
        return tuple(p._version for p in model.parameters()) + tuple(p.data_ptr() for p in model.parameters())


class DecodeSession:
    """static buffers + the captured step graph for one (batch, cache capacity, dtype)"""

    def __init__(self, model, B, Lmax, ymax, dtype, device):
        self.model, self.B, self.Lmax, self.ymax, self.dtype, self.device = model, B, Lmax, ymax, dtype, device
        E, nl, V = model.model_dim, model.num_layers, model.vocab_size
        self.E, self.H, self.nl, self.V = E, model.num_head, nl, V
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)
        self.kc = z(nl, B, Lmax, E, dt=dtype)
        self.vc = z(nl, B, Lmax, E, dt=dtype)
        self.xa, self.xb = z(B, E), z(B, E)
        self.qkv, self.att, self.t, self.u = z(B, 3 * E), z(B, E), z(B, E), z(B, E)
        self.hid = z(B, 4 * E)
        self.logits = z(B, V)
        self.y = z(B, ymax, dt=torch.int64)
        self.ctr = z(8, dt=torch.int32)
        self.stop = torch.full((B,), -1, dtype=torch.int32, device=device)
        self.x_lens, self.x_len = None, 0      # key-padding of a batch of texts (infer_panel_batch_infer); None = no padding
        self.x_lens_buf = z(B, dt=torch.int32)
        self.graph, self.graph_key = None, None

    # ---- launches ----
    def _gemv(self, w, bias, a, r, g, b, eps, x_out, y, relu=0):
        N, K = w.shape
        L.check(L.lib().evt_dec_gemv(L.dt_of(w), L.ptr(w), L.ptr(bias), L.ptr(a), L.ptr(r), L.ptr(g), L.ptr(b),
                                     C.c_float(eps), L.ptr(x_out), L.ptr(y), self.B, N, K, int(relu), L.stream_ptr()),
                "evt_dec_gemv")

    def _sample_embed_advance(self, W, sp, noise, pe, dpos):
        """sampling, append, embedding of the new token and the counter update: one launch for one sequence, three for a
        batch (the counters may only move after every row's workgroup has read them)"""
        lib = L.lib()
        xs = C.c_float(self.model.ar_audio_position.x_scale)
        if self.B == 1:
            L.check(lib.evt_dec_sample_embed(
                C.byref(sp), L.ptr(self.logits), L.ptr(self.y), L.ptr(self.ctr), L.ptr(noise), L.ptr(self.stop),
                L.ptr(W.emb), L.ptr(pe), L.ptr(W.alpha), xs, L.ptr(self.xa), self.E, pe.size(0), dpos, L.stream_ptr()),
                "evt_dec_sample_embed")
            return
        L.check(lib.evt_dec_sample(C.byref(sp), L.ptr(self.logits), L.ptr(self.y), L.ptr(self.ctr), L.ptr(noise),
                                   L.ptr(self.stop), None, self.B, L.stream_ptr()), "evt_dec_sample")
        L.check(lib.evt_dec_embed(L.ptr(W.emb), L.ptr(pe), L.ptr(W.alpha), xs, L.ptr(self.y), L.ptr(self.ctr), L.ptr(self.xa),
                                  self.B, self.E, self.ymax, pe.size(0), L.stream_ptr()), "evt_dec_embed")
        L.check(lib.evt_dec_advance(L.ptr(self.ctr), dpos, L.stream_ptr()), "evt_dec_advance")

    def _qkv_attn(self, i, w, a, r, g, b, eps, x_out):
        L.check(L.lib().evt_dec_qkv_attn(L.dt_of(self.kc), L.ptr(w["wqkv"]), L.ptr(w["bqkv"]), L.ptr(a), L.ptr(r), L.ptr(g),
                                         L.ptr(b), C.c_float(eps), L.ptr(x_out), L.ptr(self.kc[i]), L.ptr(self.vc[i]),
                                         L.ptr(self.ctr), L.ptr(self.att), self.B, self.H, self.E // self.H, self.Lmax,
                                         L.ptr(self.x_lens), self.x_len, L.stream_ptr()), "evt_dec_qkv_attn")

    def step_launches(self, W, sp, noise, pe, fused_qkv=False):
        """one token: 24 x (in-projection, cache attention, out-proj, ffn1, ffn2) + logits + one launch for sampling /
        embedding / counters = 122 launches.  fused_qkv puts the in-projection into the attention launch (98 launches):
        16 workgroups then stream all of W_qkv, which is slower on the device (measured 696 vs 614 us per token under
        graph replay) but faster when the HOST is the bottleneck (eager launches: 959 vs 1210 us)."""
        prev = None
        for i, w in enumerate(W.layers):
            ln = (None, None, None, 0.0, None) if prev is None else (self.u, prev["g2"], prev["be2"], prev["eps2"], self.xa)
            src = self.xa if prev is None else self.xb   # later blocks: LayerNorm2 of the previous one, stored to xa
            if fused_qkv:
                self._qkv_attn(i, w, src, *ln)
            else:
                self._gemv(w["wqkv"], w["bqkv"], src, *ln, self.qkv)
                L.check(L.lib().evt_dec_attn(L.dt_of(self.kc), L.ptr(self.qkv), L.ptr(self.kc[i]), L.ptr(self.vc[i]),
                                             L.ptr(self.ctr), L.ptr(self.att), self.B, self.H, self.E // self.H, self.Lmax,
                                             L.ptr(self.x_lens), self.x_len, L.stream_ptr()), "evt_dec_attn")
            self._gemv(w["wo"], w["bo"], self.att, None, None, None, 0.0, None, self.t)
            self._gemv(w["w1"], w["b1"], self.xa, self.t, w["g1"], w["be1"], w["eps1"], self.xb, self.hid, relu=1)
            self._gemv(w["w2"], w["b2"], self.hid, None, None, None, 0.0, None, self.u)
            prev = w
        self._gemv(W.wpred, None, self.xb, self.u, prev["g2"], prev["be2"], prev["eps2"], None, self.logits)
        self._sample_embed_advance(W, sp, noise, pe, 1)


class T2SInfer:
    """decoder front end bound to one Text2SemanticDecoder; `infer_panel_naive` has the reference's signature"""

    def __init__(self, model):
        self.model = model
        self._w, self._sessions, self._dense = None, {}, None

    def weights(self, dtype):
        if self._w is None or self._w[0] != dtype or self._w[1].stamp != _Weights.stamp_of(self.model):
            self._w = (dtype, _Weights(self.model, dtype))
            self._sessions.clear()          # captured graphs hold pointers into the old copies
        return self._w[1]

    def dense(self, dtype, device):
        """x, weight[, relu] -> act(x W^T + b) for the prompt pass (T2SBlock.process_prompt, t2s_model.py:124-185 of the
        reference: five F.linear per block) on the library's GEMM entry points.  Inside a trainer the model's own
        LinearBank serves (same images as the training forward); a bare model (inference/t2s.py) gets a bank of its own
        here, rebuilt when the weights were replaced or written (load_state_dict)."""
        m = self.model
        bank = getattr(m, "_bank", None)
        if bank is None or bank.dtype != dtype or bank.device != torch.device(device):
            stamp = _Weights.stamp_of(m)
            if self._dense is None or self._dense[0] != (dtype, str(device)) or self._dense[2] != stamp:
                keep = {id(w): getattr(w, "_evt_slot", None) for _n, w, _b in m.dense_specs()}
                own = LinearBank(m.dense_specs(), dtype, device)
                slots = {id(s.weight): s for s in own.slots}
                for _n, w, _b in m.dense_specs():       # a training bank of another dtype keeps its slots on the weights
                    if keep[id(w)] is not None:
                        w._evt_slot = keep[id(w)]
                    else:
                        del w._evt_slot
                self._dense = ((dtype, str(device)), own, stamp, slots)
            bank, slots = self._dense[1], self._dense[3]
        else:
            slots = None
        bank.prepare()

        def run(x, weight, relu=False):
            slot = slots[id(weight)] if slots is not None else weight._evt_slot
            return gemm_fwd(slot, x, relu=relu)

        return run

    def session(self, B, Lneed, yneed, dtype, device):
        Lmax, ymax = -(-Lneed // 512) * 512, -(-yneed // 512) * 512
        key = (B, Lmax, ymax, dtype, str(device))
        if key not in self._sessions:
            self._sessions[key] = DecodeSession(self.model, B, Lmax, ymax, dtype, device)
        return self._sessions[key]

    MAX_ROWS = 4      # rows per session (kMaxB of csrc/s1_decode.hip); larger batches run in groups

    @torch.no_grad()
    def _decode(self, xs, berts, prompts, no_eos_steps, top_k, top_p, early_stop_num, temperature, repetition_penalty,
                noise=None, seed=None, poll=8):
        """xs: B id vectors (any lengths), berts: B x [1024, n_b], prompts [B, y_len] or None.  Returns (token buffer
        [B, >= y_len + steps] on the device, per-row index of the last sampled step, per-row EOS flag, y_len)."""
        m = self.model
        if m.training:
            raise L.EvtError("decoding needs model.eval() (the reference decodes with dropout off)")
        dev, cd = xs[0].device, m.cd
        L.set_half(cd)                        # a 16-bit streaming dtype selects the build of the library that serves it
        B = len(xs)
        W = self.weights(cd)
        # ---- prompt pass (t2s_model.py:575-660 / 775-825, T2SBlock.process_prompt) ----
        x_lens = [int(x.numel()) for x in xs]
        x_len = max(x_lens)
        rows = []
        dense = self.dense(cd, dev)           # the library's GEMMs on prepared weight images (hip/linear.py), no vendor BLAS
        for x, bert in zip(xs, berts):
            xe = m.ar_text_embedding(x.unsqueeze(0))
            xe = xe + dense(bert.transpose(0, 1).unsqueeze(0).to(cd).contiguous(), m.bert_proj.weight).to(xe.dtype)
            xe = m.ar_text_position(xe).squeeze(0)
            rows.append(F.pad(xe, (0, 0, 0, x_len - xe.size(0))))      # padded text positions: masked as keys below
        xe = torch.stack(rows, dim=0)
        if prompts is None:
            y_len, xy = 0, xe
        else:
            y_len = prompts.size(1)
            xy = torch.cat([xe, m.ar_audio_position(m.ar_audio_embedding(prompts))], dim=1)
        xy = xy.to(cd).contiguous()
        src_len = x_len + y_len
        n_max = MAX_STEPS if early_stop_num == -1 else max(1, min(MAX_STEPS, int(early_stop_num) + 1))
        noise_rows = 1
        if noise is not None:       # an injected noise table (parity runs) also bounds the number of steps
            noise = noise.to(dev, torch.float32).contiguous()
            noise_rows = 1 if noise.dim() == 2 else noise.size(1)
            assert noise.size(-1) == m.vocab_size and noise_rows in (1, B)
            n_max = min(n_max, noise.size(0))
        S = self.session(B, src_len + n_max + 1, y_len + n_max + 1, cd, dev)
        xl = torch.tensor(x_lens, dtype=torch.int32, device=dev)
        yl = torch.full((B,), y_len, dtype=torch.int32, device=dev)
        for i, lyr in enumerate(m.h.layers):
            w = W.layers[i]
            qkv = dense(xy, lyr.self_attn.in_proj_weight)
            S.kc[i, :, :src_len].copy_(qkv[..., S.E:2 * S.E])
            S.vc[i, :, :src_len].copy_(qkv[..., 2 * S.E:])
            o = PrefixLMAttentionFn.apply(qkv, xl, yl, x_len, S.H, 0.0, 0)
            sa = dense(o.contiguous(), lyr.self_attn.out_proj.weight)
            xy = AddLayerNormFn.apply(xy, sa, w["g1"], w["be1"], w["eps1"])
            ff = dense(dense(xy.contiguous(), lyr.linear1.weight, relu=True), lyr.linear2.weight)
            xy = AddLayerNormFn.apply(xy, ff, w["g2"], w["be2"], w["eps2"])
        # ---- state ----
        S.y.zero_()
        if prompts is not None:
            S.y[:, :y_len].copy_(prompts)
        # the sampling seed is device state like the counters (a new one per call must not force a re-capture); without
        # an explicit seed it is drawn from torch's CPU generator, so torch.manual_seed makes a run repeatable
        seed = int(seed if seed is not None else torch.randint(0, 2 ** 31 - 1, (1,)).item()) & 0x7FFFFFFF
        S.ctr.copy_(torch.tensor([src_len, 0, y_len, y_len, seed, 0, 0, 0], dtype=torch.int32))
        S.stop.fill_(-1)
        padded = min(x_lens) < x_len
        if padded:
            S.x_lens_buf.copy_(xl)
        S.x_lens, S.x_len = (S.x_lens_buf if padded else None), x_len
        sp = L.SampleParams(S.V, m.EOS, int(top_k) if top_k is not None else 0, no_eos_steps, S.ymax, float(top_p),
                            float(temperature), float(repetition_penalty), 0x5EED5EED, noise_rows)
        pe = m.ar_audio_position.pe(max(4000, y_len + n_max + 1), dev, torch.float32).contiguous()
        # ---- step 0: logits of the last prompt position, sample, embed ----
        S.xb.copy_(xy[:, -1].float())
        S._gemv(W.wpred, None, S.xb, None, None, None, 0.0, None, S.logits)
        S._sample_embed_advance(W, sp, noise, pe, 0)
        # ---- token steps: one graph replay each ----
        from .. import hip_graphs_safe

        use_graph = os.environ.get("EVT_DECODE_GRAPH", "1") != "0" and hip_graphs_safe()
        gkey = (bytes(sp), None if noise is None else noise.data_ptr(), pe.data_ptr(), id(W), padded, x_len)
        if use_graph and S.graph_key != gkey:
            # warm-up launches outside the capture, on throw-away counters: restore the state afterwards
            keep = (S.ctr.clone(), S.y.clone(), S.stop.clone(), S.xa.clone())
            side = L.role_stream(dev, "decode_warm", ring=1)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                S.step_launches(W, sp, noise, pe)
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="relaxed"):
                S.step_launches(W, sp, noise, pe)
            S.ctr.copy_(keep[0]); S.y.copy_(keep[1]); S.stop.copy_(keep[2]); S.xa.copy_(keep[3])
            S.graph, S.graph_key, S._keep = g, gkey, (sp, noise, pe, W)
        done = 1
        stop = S.stop.tolist() if n_max == 1 else [-1] * B
        while done < n_max and min(stop) < 0:
            if use_graph:
                S.graph.replay()
            else:
                S.step_launches(W, sp, noise, pe, fused_qkv=True)
            done += 1
            if done % poll == 0 or done == n_max:
                stop = S.stop.tolist()              # the only device->host read of the loop
        eos = [s >= 0 for s in stop]
        last = [s if s >= 0 else n_max - 1 for s in stop]   # else: early_stop_num reached, or 1500 steps without EOS
        return S.y, last, eos, y_len

    def infer_panel_naive(self, x, x_lens, prompts, bert_feature, top_k=-100, top_p=100, early_stop_num=-1,
                          temperature=1.0, repetition_penalty=1.35, noise=None, seed=None, poll=8, **kwargs):
        """t2s_model.py:762-863: one sequence; returns (y without its last token, idx - 1), or (.., 0) without a prompt"""
        if x.size(0) != 1:
            raise L.EvtError("one sequence per call (infer_panel_naive_batched loops over the items, t2s_model.py:732-760)")
        ybuf, last, _eos, y_len = self._decode([x[0]], [bert_feature[0]], prompts, NO_EOS_STEPS, top_k, top_p, early_stop_num,
                                               temperature, repetition_penalty, noise=noise, seed=seed, poll=poll)
        y = ybuf[:, :y_len + last[0]].clone()       # the last sampled token (EOS or the stop token) is dropped
        if prompts is None:
            return y.to(torch.int32), 0
        return y, last[0] - 1

    def infer_panel_naive_batched(self, x, x_lens, prompts, bert_feature, **kw):
        ys, idxs = [], []
        for i in range(len(x)):
            y, idx = self.infer_panel_naive(x[i].unsqueeze(0), x_lens[i], prompts[i].unsqueeze(0) if prompts is not None
                                            else None, bert_feature[i].unsqueeze(0), **kw)
            ys.append(y[0])
            idxs.append(idx)
        return ys, idxs

    def infer_panel_batch_infer(self, x, x_lens, prompts, bert_feature, top_k=-100, top_p=100, early_stop_num=-1,
                                temperature=1.0, repetition_penalty=1.35, noise=None, seed=None, poll=8, **kwargs):
        """t2s_model.py:563-730, the TTS default (parallel_infer=True): texts of different lengths decoded together.
        x: list of id vectors, bert_feature: list of [1024, n].  Rows are independent (padded text positions are masked
        as keys, a finished row only leaves the batch), so instead of compacting the batch whenever a row meets EOS, all
        rows of a group keep stepping through the same graph and each row's tokens are cut at its own stop.  Kept
        differences to infer_panel_naive: the EOS column is dropped at step 0 only, the returned index is idx - 1 for an
        EOS stop and idx for the early stop."""
        if prompts is None:
            return self.infer_panel_naive_batched(x, x_lens, prompts, bert_feature, top_k=top_k, top_p=top_p,
                                                  early_stop_num=early_stop_num, temperature=temperature, noise=noise,
                                                  seed=seed, poll=poll)
        ys, idxs = [], []
        for g0 in range(0, len(x), self.MAX_ROWS):
            rows = list(range(g0, min(len(x), g0 + self.MAX_ROWS)))
            nz = noise if noise is None or noise.dim() == 2 else noise[:, rows]
            ybuf, last, eos, y_len = self._decode([x[r] for r in rows], [bert_feature[r] for r in rows], prompts[rows], 1,
                                                  top_k, top_p, early_stop_num, temperature, repetition_penalty, noise=nz,
                                                  seed=None if seed is None else seed + g0, poll=poll)
            for k in range(len(rows)):
                ys.append(ybuf[k, :y_len + last[k]].clone())
                idxs.append(last[k] - 1 if eos[k] else last[k])
        return ys, idxs
