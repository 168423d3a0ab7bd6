"""CPU: host side of the feature extractors -- checkpoint key mapping against transformers' own
modules, the half-band resampler against its definition, the file formats of the writer read back by the training readers.
The forward parity of the two models is the GPU tier's (tests/test_feature_extractors_gpu.py)."""
import os

import sys

import numpy as np
import pytest
import torch


@pytest.fixture(autouse=True)
def _no_reference_stand_ins():
    """oracle/refshim.py (installed by the differential tests of this tier when /root/reference is present) puts stand-in
    `librosa` modules into sys.modules; transformers probes optional back-ends with importlib.util.find_spec at import time
    and chokes on them.  They are taken out for these tests and put back afterwards."""
    held = {k: sys.modules.pop(k) for k in list(sys.modules)
            if k.split(".")[0] in ("librosa", "torchmetrics") and getattr(sys.modules[k], "__file__", None) is None}
    yield
    sys.modules.update(held)



def test_hubert_checkpoint_keys_and_folded_positional_weight():
    from transformers import HubertConfig
    from transformers import HubertModel as HF

    from easevoice_trainer_amd.feature_extractor.cnhubert import CONV_DIM, CONV_KERNEL, CONV_STRIDE, HubertModel

    cfg = HubertConfig()
    assert (tuple(cfg.conv_dim), tuple(cfg.conv_kernel), tuple(cfg.conv_stride)) == (CONV_DIM, CONV_KERNEL, CONV_STRIDE)
    assert cfg.feat_extract_norm == "group" and not cfg.do_stable_layer_norm and cfg.hidden_act == "gelu"
    torch.manual_seed(0)
    hf = HF(cfg)
    m = HubertModel().load_hf_state_dict(hf.state_dict())          # raises on any missing / unexpected key
    # weight_norm over the tap axis, folded once: the convolution weight transformers computes on every call
    assert torch.allclose(m.encoder.pos_conv_embed.conv.weight, hf.encoder.pos_conv_embed.conv.weight, atol=1e-6)
    own = dict(m.named_parameters())
    for k, v in hf.state_dict().items():
        if "parametrizations" in k:
            continue
        assert torch.equal(own[k].detach(), v), k
    for n in (16000, 16001, 400, 32000 * 7 + 13):
        with torch.no_grad():
            want = hf._get_feat_extract_output_lengths(torch.tensor(n)).item()
        assert HubertModel.frames(n) == want


def test_bert_checkpoint_keys_and_layer_that_is_read():
    from transformers import BertConfig, BertForMaskedLM

    from easevoice_trainer_amd.feature_extractor.roberta import BertEncoderStack

    cfg = BertConfig(vocab_size=300, hidden_size=64, num_hidden_layers=6, num_attention_heads=4, intermediate_size=128)
    torch.manual_seed(0)
    hf = BertForMaskedLM(cfg)
    # hidden_states[-3] of an L-layer model is the output of layer L - 2 (hidden_states[0] is the embedding output)
    s = BertEncoderStack(vocab=300, hidden=64, heads=4, inner=128, num_layers=6, read_layer=4).load_hf_state_dict(hf.state_dict())
    assert len(s.bert.encoder.layer) == 4
    own = dict(s.named_parameters())
    sd = hf.state_dict()
    for k, v in own.items():
        assert torch.equal(v.detach(), sd[k]), k
    with pytest.raises(KeyError):
        BertEncoderStack(vocab=300, hidden=64, heads=4, inner=128, num_layers=6, read_layer=4).load_hf_state_dict(
            {k: v for k, v in sd.items() if "layer.3." not in k})
    # the reference's defaults (chinese-roberta-wwm-ext-large read at hidden_states[-3:-2], normalize.py:93)
    big = BertEncoderStack.__init__.__defaults__
    assert big[:6] == (21128, 1024, 16, 4096, 24, 22)


def test_half_band_resampler_is_its_definition_and_a_half_band_filter():
    from easevoice_trainer_amd.feature_extractor.normalize import _half_band_taps, resample_half

    h = _half_band_taps()
    assert len(h) == 128 and abs(2 * h.sum() - h[0] - 1.0) < 1e-6          # unit gain at DC
    x = np.random.RandomState(0).randn(1001).astype(np.float32)
    y = resample_half(x)
    assert len(y) == 500
    for t in (0, 1, 63, 250, 499):
        s = sum(h[abs(j)] * x[2 * t + j] for j in range(-127, 128) if 0 <= 2 * t + j < len(x))
        assert abs(s - y[t]) < 1e-5
    tt = np.arange(32000) / 32000.0
    rms = lambda f: float(np.sqrt((resample_half(np.sin(2 * np.pi * f * tt).astype(np.float32))[500:-500] ** 2).mean()))
    assert abs(rms(1000.0) - 2 ** -0.5) < 1e-4 and abs(rms(7000.0) - 2 ** -0.5) < 1e-4     # pass band
    assert rms(9000.0) < 1e-6 and rms(15000.0) < 1e-6                                       # above the new Nyquist


def test_rescale_follows_the_reference_lines():
    from easevoice_trainer_amd.feature_extractor.normalize import rescale_clip

    a = (np.random.RandomState(1).rand(5000).astype(np.float32) - 0.5) * 0.8
    i16, f = rescale_clip(a)
    mx = np.abs(a).max()
    assert i16.dtype == np.int16 and np.array_equal(i16, ((a / mx * (0.95 * 0.5 * 32768)) + (0.5 * 32768) * a).astype("int16"))
    assert np.allclose(f, (a / mx * (0.95 * 0.5 * 1145.14)) + (0.5 * 1145.14) * a)
    assert rescale_clip(a * 10.0) is None          # peak above 2.2: the reference skips the clip (normalize.py:150-151)


def test_writer_files_are_read_back_by_the_training_readers(tmp_path):
    """FeatureWriter with stand-in models: file names, tensor layouts and the text line as the reference writes them, read
    back by train/dataset.py's s2 and s1 readers"""
    from scipy.io import wavfile

    from easevoice_trainer_amd.feature_extractor.cnhubert import HubertModel
    from easevoice_trainer_amd.feature_extractor.normalize import FeatureWriter, resample_half
    from easevoice_trainer_amd.train import dataset as D

    seen = {}

    def fake_hubert(wav16):
        seen["n16"] = len(wav16)
        return torch.randn(1, 768, HubertModel.frames(len(wav16)))

    class FakeBert:
        def phone_level_feature(self, input_ids, word2ph):
            return torch.randn(1024, sum(word2ph))

    w = FeatureWriter(str(tmp_path), fake_hubert, FakeBert())
    audio = (np.random.RandomState(2).rand(32000 * 2) - 0.5).astype(np.float32)
    assert w.ssl("clip_a.wav", audio)
    assert seen["n16"] == len(resample_half(audio)) == 32000
    phones, word2ph = ["n", "i3", "h", "ao3"], [2, 2]
    w.text("clip_a.wav", phones, word2ph, "你好", "zh", input_ids=torch.zeros(1, 4, dtype=torch.long))
    w.text("clip_b.wav", ["AH0"], [1], "a", "en")                  # no BERT feature for other languages, line still written
    w.close()
    ssl = torch.load(tmp_path / "4-cnhubert" / "clip_a.wav.pt")
    assert ssl.shape == (1, 768, HubertModel.frames(32000)) and ssl.dtype == torch.float32
    rate, pcm = wavfile.read(tmp_path / "5-wav32k" / "clip_a.wav")
    assert rate == 32000 and pcm.dtype == np.int16 and len(pcm) == len(audio)
    bert = torch.load(tmp_path / "3-bert" / "clip_a.wav.pt")
    assert bert.shape == (1024, 4)
    assert not os.path.exists(tmp_path / "3-bert" / "clip_b.wav.pt")
    lines = open(tmp_path / "2-name2text.txt", encoding="utf8").read().split("\n")
    assert lines[0] == "clip_a.wav\tn i3 h ao3\t[2, 2]\t你好" and lines[1] == "clip_b.wav\tAH0\t[1]\ta" and lines[2] == ""
    table = D.read_name2text(str(tmp_path / "2-name2text.txt")) if hasattr(D, "read_name2text") else None
    if table is not None:
        assert table["clip_a.wav"][0] == "n i3 h ao3"
    with pytest.raises(ValueError):
        w.text("clip_c.wav", ["a"], [1, 1], "ab", "zh", input_ids=torch.zeros(1, 4, dtype=torch.long))
