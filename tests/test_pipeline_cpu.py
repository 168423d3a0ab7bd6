"""CPU: the inference-side chain (inference/t2s.py, inference/pipeline.py): export loading like TTS.init_t2s_weights and
the model-side core of TTS.run -- batched s1 decoding, then SynthesizerTrn.decode over the concatenated fragments (speed
1.0) or per fragment -- against outputs of the reference's own two models chained the same way
(tests/golden/pipeline.pt).  HIP launches are emulated (tests/cpu_emu.py); the kernels are covered by the -m gpu tests."""
import json
import os
import sys
from types import SimpleNamespace

import torch
import yaml

from cpu_emu import cpu_emulation, cpu_emulation_decode
from util_fill import decode_inputs, fill_module

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))


def rel(a, b):
    return ((a.detach().float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12)).item()


def test_semantic_to_audio_chain_matches_reference(tmp_path):
    from make_golden_s1_inputs import pipeline_inputs
    from easevoice_trainer_amd.auto_reg.t2s_model import Text2SemanticDecoder
    from easevoice_trainer_amd.inference.pipeline import synthesize_fragments
    from easevoice_trainer_amd.inference.t2s import T2SVoice
    from easevoice_trainer_amd.module import models

    torch.set_num_threads(8)
    gold = torch.load(os.path.join(HERE, "golden", "pipeline.pt"), weights_only=False)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    d, dd = pipeline_inputs(), decode_inputs()
    with cpu_emulation(), cpu_emulation_decode():
        # an s1 export in the trainer's layout ("model." keys), loaded the way the reference's TTS loads it
        src = Text2SemanticDecoder(cfg)
        fill_module(src, 3)
        path = str(tmp_path / "t2s-e1.ckpt")
        torch.save({"weight": {"model." + k: v.clone() for k, v in src.state_dict().items()}, "config": cfg, "info": "GPT-e1"},
                   path)
        t2s = T2SVoice(path, device="cpu", dtype=torch.float32)
        assert t2s.early_stop_num == 50 * cfg["data"]["max_sec"] and not t2s.model.training
        net_g = models.SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
        fill_module(net_g, 1)
        net_g.eval()
        voice = SimpleNamespace(model=net_g, hps=hps)
        kw = dict(top_k=1100, top_p=1, temperature=1.0, repetition_penalty=1.35, sample_kwargs=dict(noise=d["q"]),
                  decode_kwargs=dict(noise=dd["noise"]))
        frags = synthesize_fragments(t2s, voice, d["batch_phones"], d["all_ids"], d["bert"], d["prompt"], dd["refers"],
                                     speed_factor=1.0, **kw)
        assert [f.numel() for f in frags] == [g.numel() for g in gold["speed1"]]
        for f, g in zip(frags, gold["speed1"]):
            assert rel(f, g) < 1e-4
        frags = synthesize_fragments(t2s, voice, d["batch_phones"], d["all_ids"], d["bert"], d["prompt"], dd["refers"],
                                     speed_factor=1.25, **kw)
        for f, g in zip(frags, gold["speed125"]):
            assert f.shape == g.shape and rel(f, g) < 1e-4
        # the token lists themselves (EOS stops at different steps, idx = step - 1)
        ys, idxs = t2s.model.infer_panel_batch_infer(d["all_ids"], None, d["prompt"].expand(2, -1), d["bert"], top_k=1100,
                                                     top_p=1, early_stop_num=t2s.early_stop_num, noise=d["q"])
        assert idxs == gold["idx"] and all(torch.equal(y.long(), g.long()) for y, g in zip(ys, gold["pred"]))
