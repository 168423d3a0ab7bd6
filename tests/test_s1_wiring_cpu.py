"""CPU: the product's s1 module wiring and ScaledAdam host logic vs fixtures generated from the reference's own
Text2SemanticDecoder / ScaledAdam.  HIP launches are substituted by torch CPU ops (tests/cpu_emu.py); the kernels are
covered by the -m gpu tests."""
import os

import pytest
import torch
import yaml

from cpu_emu import cpu_emulation_s1
from util_fill import fill_module, s1_batch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def rel(a, b):
    return ((a.detach().float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12)).item()


def test_s1_forward_backward_wiring():
    from easevoice_trainer_amd.auto_reg.t2s_model import Text2SemanticDecoder

    gold = torch.load(os.path.join(HERE, "golden", "s1_small.pt"), weights_only=False)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    with cpu_emulation_s1():
        m = Text2SemanticDecoder(cfg)
        fill_module(m, 3)
        m.eval()     # dropout off (the fixture zeroes every dropout of the reference)
        c = gold["config"]
        b = s1_batch(c["B"], c["x_len"], c["y_len"])
        loss, acc = m.forward_old(b["phoneme_ids"], torch.tensor(c["x_lens"]), b["semantic_ids"],
                                  torch.tensor(c["y_lens"]), b["bert_feature"])
        assert abs(float(loss) - gold["loss"]) <= 1e-4 * gold["loss"]
        assert abs(float(acc) - gold["acc"]) < 1e-6
        loss.backward()
        params = dict(m.named_parameters())
        for n, s in gold["grad_slices"].items():
            assert rel(params[n].grad.flatten()[:96], s) < 2e-3, n


def test_s1_dpo_forward_backward_wiring():
    """the DPO branch: same rejected sequences as the reference for the same torch seed, same loss / gradients"""
    from easevoice_trainer_amd.auto_reg.t2s_model import Text2SemanticDecoder
    from easevoice_trainer_amd.auto_reg.utils import make_reject_y

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    for gold in torch.load(os.path.join(HERE, "golden", "s1_dpo.pt"), weights_only=False)["cases"]:
        c = gold["config"]
        b = s1_batch(c["B"], c["x_len"], c["y_len"])
        y_lens = torch.tensor(c["y_lens"])
        torch.manual_seed(c["seed"])
        ry, rl = make_reject_y(b["semantic_ids"], y_lens)
        assert torch.equal(ry, gold["reject_y"]) and torch.equal(rl, gold["reject_y_lens"])
        with cpu_emulation_s1():
            m = Text2SemanticDecoder(cfg)
            fill_module(m, 3)
            m.eval()
            torch.manual_seed(c["seed"])
            loss, acc = m.forward(b["phoneme_ids"], torch.tensor(c["x_lens"]), b["semantic_ids"], y_lens, b["bert_feature"])
            assert abs(float(loss) - gold["loss"]) <= 1e-4 * gold["loss"]
            assert abs(float(acc) - gold["acc"]) < 1e-6
            loss.backward()
            params = dict(m.named_parameters())
            for n, s in gold["grad_slices"].items():
                assert rel(params[n].grad.flatten()[:96], s) < 2e-3, (c["seed"], n)


def test_scaled_adam_host_logic_matches_reference():
    from easevoice_trainer_amd.auto_reg.optim import ScaledAdam
    from easevoice_trainer_amd.runtime import ParamArena

    gold = torch.load(os.path.join(HERE, "golden", "s1_small.pt"), weights_only=False)["scaled_adam"]

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for k, v in gold["init"].items():
                setattr(self, k, torch.nn.Parameter(v.clone()))

    with cpu_emulation_s1():
        h = Holder()
        arena = ParamArena(h, "cpu")
        opt = ScaledAdam(arena, lr=0.01, betas=(0.9, 0.95), clipping_scale=2.0, clipping_update_period=4)
        params = dict(h.named_parameters())
        for step, (grads, want) in enumerate(zip(gold["grads"], gold["traj"])):
            arena.zero_grad()
            for k, g in grads.items():
                params[k].grad.copy_(g)
            opt.step()
            opt.param_groups[0]["lr"] = 0.002
            for k in params:
                assert torch.allclose(params[k].detach(), want[k], rtol=5e-5, atol=2e-6), (step, k)


def test_decoding_host_logic_matches_reference_tokens():
    """auto_reg/t2s_infer.py with its launches emulated on the session's buffers: prompt pass, device-counter protocol,
    stop polling, output cuts and index conventions of infer_panel_naive and infer_panel_batch_infer (padded batch, rows
    stopping at different steps, groups) reproduce the reference's token sequences"""
    import sys
    from cpu_emu import cpu_emulation_decode
    from easevoice_trainer_amd.auto_reg.t2s_model import Text2SemanticDecoder
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden_s1_inputs import batch_infer_inputs, infer_inputs

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    with cpu_emulation_decode():
        m = Text2SemanticDecoder(cfg)
        fill_module(m, 3)
        m.eval()
        d = infer_inputs()
        for gold in torch.load(os.path.join(HERE, "golden", "s1_infer.pt"), weights_only=False)["cases"]:
            a = dict(gold["args"])
            prompts = d["prompts"] if a.pop("prompt") else None
            y, idx = m.infer_panel_naive(d["x"], torch.tensor([24]), prompts, d["bert"], noise=d["q"], poll=5, **a)
            assert y.dtype == gold["y"].dtype and torch.equal(y.long(), gold["y"].long()) and idx == gold["idx"]
        d = batch_infer_inputs()
        for gold in torch.load(os.path.join(HERE, "golden", "s1_batch_infer.pt"), weights_only=False)["cases"]:
            a = dict(gold["args"])
            rows = a.pop("rows")
            ys, idxs = m.infer_panel_batch_infer([d["x"][r] for r in rows], d["x_lens"][rows], d["prompts"][rows],
                                                 [d["bert"][r] for r in rows], noise=d["q"][:, rows], **a)
            assert idxs == gold["idx"]
            for y, g in zip(ys, gold["y"]):
                assert torch.equal(y.long(), g.long())
        # more rows than a session holds -> two groups; duplicates of a text decode to the same tokens
        order = [0, 1, 2, 1, 0, 2]
        ys, idxs = m.infer_panel_batch_infer([d["x"][r] for r in order], d["x_lens"][order], d["prompts"][order],
                                             [d["bert"][r] for r in order], top_k=15, top_p=1, early_stop_num=6,
                                             noise=d["q"][:, order])
        assert idxs == [6] * 6 and torch.equal(ys[1], ys[3]) and torch.equal(ys[0], ys[4]) and torch.equal(ys[2], ys[5])
        m.train()
        with pytest.raises(Exception, match="eval"):
            m.infer_panel_naive(d["x"][0][None], None, d["prompts"][:1], d["bert"][0][None], top_k=5, early_stop_num=2)
