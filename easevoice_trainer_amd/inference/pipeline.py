"""The model-side core of TTS.run for one batch of text fragments (src/easevoice/inference/tts.py:756-817): semantic
tokens from the s1 decoder, waveform fragments from the s2 decoder.  Everything around it in the reference -- text
front-end, BERT / CN-HuBERT feature extraction, bucketing, audio post-processing -- is outside SURVEY §8."""
import math

import torch


@torch.no_grad()
def synthesize_fragments(t2s, voice, batch_phones, all_phoneme_ids, all_bert_features, prompt_semantic, refer_specs,
                         top_k=5, top_p=1, temperature=1.0, repetition_penalty=1.35, speed_factor=1.0,
                         parallel_infer=True, max_len=None, decode_kwargs=None, sample_kwargs=None):
    """batch_phones: the fragments' own phoneme ids (list of 1-D); all_phoneme_ids / all_bert_features: prompt + fragment
    ids and their BERT features [1024, n] (lists); prompt_semantic [1, P] or None; refer_specs: list of [1, spec, T].
    Returns the list of waveform fragments (1-D tensors on the device), tts.py:794-817."""
    model = t2s.model
    dev = t2s.device
    n = len(all_phoneme_ids)
    prompt = None if prompt_semantic is None else prompt_semantic.expand(n, -1).to(dev)
    infer = model.infer_panel_batch_infer if parallel_infer else model.infer_panel_naive_batched
    lens = torch.tensor([int(p.numel()) for p in all_phoneme_ids])
    pred, idx_list = infer([p.to(dev) for p in all_phoneme_ids], lens, prompt, [b.to(dev) for b in all_bert_features],
                           top_k=top_k, top_p=top_p, temperature=temperature, early_stop_num=t2s.early_stop_num,
                           max_len=max_len, repetition_penalty=repetition_penalty, **(sample_kwargs or {}))
    refer = [r.to(dev) for r in refer_specs]
    kw = decode_kwargs or {}
    if speed_factor == 1.0:
        # one decode over the concatenated fragments, then cut (tts.py:795-807)
        pred = [p[-i:] for p, i in zip(pred, idx_list)]
        up = math.prod(voice.model.upsample_rates)
        ends = [0]
        for p in pred:
            ends.append(ends[-1] + p.shape[0] * 2 * up)
        sem = torch.cat(pred).unsqueeze(0).unsqueeze(0)
        phones = torch.cat([p.to(dev) for p in batch_phones]).unsqueeze(0)
        audio = voice.model.decode(sem, phones, refer, speed=speed_factor, **kw)[0, 0, :]
        return [audio[ends[i - 1]:ends[i]] for i in range(1, len(ends))]
    out = []
    for p, i, ph in zip(pred, idx_list, batch_phones):
        sem = p[-i:].unsqueeze(0).unsqueeze(0)
        out.append(voice.model.decode(sem, ph.to(dev).unsqueeze(0), refer, speed=speed_factor, **kw)[0, 0, :])
    return out
