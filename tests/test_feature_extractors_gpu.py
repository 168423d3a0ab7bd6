"""GPU: CN-HuBERT and Chinese-RoBERTa feature extraction on the library against transformers' own
HubertModel / BertForMaskedLM run on the host CPU in fp32 -- the computation src/normalization/normalize.py:88-106,158
performs -- on seeded random-initialised weights of the real architectures (no checkpoint download: there is no network;
the architectures and the parameter names are the checkpoints').  fp32 at the north_star's 1e-3; bf16 looser.  Then the
written files: 4-cnhubert/<name>.pt through SynthesizerTrn.extract_latent into a 6-name2semantic.tsv line, byte for byte
what the reference's `token` step writes for the same features."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _hf_hubert(seed=0):
    from transformers import HubertConfig
    from transformers import HubertModel as HF

    torch.manual_seed(seed)
    hf = HF(HubertConfig()).eval()
    with torch.no_grad():
        # random init leaves every LayerNorm at (1, 0) and the biases at 0: move them, so that a dropped scale / bias shows
        for n, p in hf.named_parameters():
            if n.endswith("layer_norm.weight") or n.endswith("LayerNorm.weight"):
                p.add_(0.1 * torch.randn_like(p))
            elif n.endswith(".bias"):
                p.add_(0.05 * torch.randn_like(p))
    return hf


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)], ids=["f32", "bf16"])
def test_cnhubert_last_hidden_state_matches_transformers(gpu, dtype, tol):
    from easevoice_trainer_amd.feature_extractor import CNHubert

    hf = _hf_hubert()
    ours = CNHubert(hf.state_dict(), gpu, dtype)
    g = torch.Generator().manual_seed(3)
    for n in (16000 * 2 + 37, 9000):
        wav = torch.randn(n, generator=g) * 0.1 * 1145.14 / 32768 * 30          # the amplitude range of normalize.py:153
        with torch.no_grad():
            ref = hf(wav.unsqueeze(0))["last_hidden_state"]                          # [1, T, 768]
        got = ours(wav)                                                              # [1, 768, T] on the CPU
        assert got.shape == (1, 768, ref.size(1)) and got.dtype == torch.float32
        assert _rel(got.transpose(1, 2), ref) < tol, (n, _rel(got.transpose(1, 2), ref))
    # the stages, fp32 only: where a deviation would come from
    if dtype == torch.float32:
        wav = torch.randn(12000, generator=g) * 0.1
        with torch.no_grad():
            f_ref = hf.feature_extractor(wav.unsqueeze(0)).transpose(1, 2)              # [1, T, 512]
            p_ref = hf.feature_projection(f_ref)
            p_ref = p_ref[0] if isinstance(p_ref, tuple) else p_ref
            e_ref = hf.encoder.layer_norm(p_ref + hf.encoder.pos_conv_embed(p_ref))
            m = ours.model
            x = wav.to(gpu).view(1, -1, 1).contiguous()
            f = m.feature_extractor(x)
            p = m.feature_projection(f)
            e = m.encoder.layer_norm(p, m.encoder.pos_conv_embed(p))
        assert _rel(f, f_ref) < 1e-3 and _rel(p, p_ref) < 1e-3 and _rel(e, e_ref) < 1e-3, (_rel(f, f_ref), _rel(p, p_ref), _rel(e, e_ref))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)], ids=["f32", "bf16"])
def test_roberta_phone_level_features_match_transformers(gpu, dtype, tol):
    """BERT-large dimensions with 24 layers (22 are run: hidden_states[-3]); a smaller vocabulary keeps the random model's
    host memory down -- the embedding gather does not depend on the table's height"""
    from transformers import BertConfig, BertForMaskedLM

    from easevoice_trainer_amd.feature_extractor import BertFeatures

    cfg = BertConfig(vocab_size=2048, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                     max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12)
    torch.manual_seed(1)
    hf = BertForMaskedLM(cfg).eval()
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if n.endswith("LayerNorm.weight"):
                p.add_(0.1 * torch.randn_like(p))
            elif n.endswith(".bias"):
                p.add_(0.05 * torch.randn_like(p))
    ours = BertFeatures(hf.state_dict(), gpu, dtype, vocab=2048)
    g = torch.Generator().manual_seed(4)
    for n_chars in (17, 1, 120):
        ids = torch.randint(5, 2048, (1, n_chars + 2), generator=g)
        word2ph = torch.randint(1, 4, (n_chars,), generator=g).tolist()
        with torch.no_grad():
            res = hf(input_ids=ids, attention_mask=torch.ones_like(ids), token_type_ids=torch.zeros_like(ids),
                     output_hidden_states=True)
            res = torch.cat(res["hidden_states"][-3:-2], -1)[0].cpu()[1:-1]            # normalize.py:93
        ref = torch.cat([res[i].repeat(word2ph[i], 1) for i in range(n_chars)], dim=0).T   # normalize.py:99-105
        got = ours.phone_level_feature(ids, word2ph)
        assert got.shape == ref.shape == (1024, sum(word2ph)) and got.dtype == torch.float32
        assert _rel(got, ref) < tol, (n_chars, _rel(got, ref))
    # a right-padded batch: padded keys are excluded, live rows equal the single-sentence rows
    ids = torch.randint(5, 2048, (2, 30), generator=g)
    mask = torch.ones(2, 30, dtype=torch.long)
    mask[1, 19:] = 0
    with torch.no_grad():
        hs = hf(input_ids=ids, attention_mask=mask, output_hidden_states=True)["hidden_states"][-3]
    got = ours.hidden(ids, mask)
    assert _rel(got[0], hs[0]) < tol and _rel(got[1, :19], hs[1, :19]) < tol
    with pytest.raises(ValueError):
        ours.phone_level_feature(ids[:1], [1, 2, 3])


def test_features_to_semantic_tsv_chain(gpu, tmp_path):
    """wav -> FeatureWriter.ssl (CN-HuBERT on the GPU) -> 4-cnhubert/<name>.pt -> extract_latent -> 6-name2semantic.tsv:
    the line equals what the reference's `token` step (normalize.py:181-211) derives from the SAME feature file with the
    oracle's quantiser (codes = nearest codebook row of ssl_proj(ssl))"""
    import json

    from easevoice_trainer_amd.feature_extractor import CNHubert
    from easevoice_trainer_amd.feature_extractor.normalize import FeatureWriter
    from easevoice_trainer_amd.inference.semantic import write_semantic_tsv
    from easevoice_trainer_amd.module import models
    from easevoice_trainer_amd.runtime import ModelRuntime
    from util_fill import fill_module

    hub = CNHubert(_hf_hubert(5).state_dict(), gpu, torch.float32)
    w = FeatureWriter(str(tmp_path), hub, None)
    audio = (np.random.RandomState(6).rand(32000 * 3 + 211) - 0.5).astype(np.float32)
    assert w.ssl("u.wav", audio)
    ssl = torch.load(tmp_path / "4-cnhubert" / "u.wav.pt")
    assert ssl.dim() == 3 and ssl.size(0) == 1 and ssl.size(1) == 768 and torch.isfinite(ssl).all()
    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    d = hps["data"]
    net = models.SynthesizerTrn(d["filter_length"] // 2 + 1, hps["train"]["segment_size"] // d["hop_length"],
                                n_speakers=d["n_speakers"], **hps["model"])
    fill_module(net, 1)
    net.eval()
    rt = ModelRuntime(net, torch.float32, gpu)
    rt.prepare(force=True)
    with torch.no_grad():
        n = write_semantic_tsv(lambda s: net.extract_latent(s.to(gpu)), ["u.wav", "missing.wav"], str(tmp_path / "4-cnhubert"),
                               str(tmp_path / "6-name2semantic.tsv"), device=gpu)
    assert n == 1
    lines = open(tmp_path / "6-name2semantic.tsv", encoding="utf8").read().split("\n")
    assert lines[0] == "item_name\tsemantic_audio" and lines[1].startswith("u.wav\t") and lines[2] == ""
    codes = [int(c) for c in lines[1].split("\t")[1].split(" ")]
    # independent restatement of extract_latent (models.py:1015-1018): ssl_proj (k = 2, stride 2) then the nearest codebook row
    from oracle import ops as O  # noqa: F401  (test infrastructure)

    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    y = torch.nn.functional.conv1d(ssl.float(), sd["ssl_proj.weight"], sd["ssl_proj.bias"], stride=2)
    emb = sd["quantizer.vq.layers.0._codebook.embed"]
    flat = y[0].T
    dist = -(flat.pow(2).sum(1, keepdim=True) - 2 * flat @ emb.T + emb.pow(2).sum(1)[None])
    want = dist.argmax(-1).tolist()
    assert len(codes) == len(want)
    agree = sum(int(a == b) for a, b in zip(codes, want)) / len(want)
    assert agree >= 0.99, agree          # a tie within fp32 rounding may pick the neighbouring row
