"""GPU: the s1 micro-step in the reference's `precision: 16-mixed` mode (configs/gpt.yaml:6; Lightning's AMP plugin around the
manual optimisation of t2s_lightning_module.py:41-89) -- dtype torch.float16 on the IEEE-half build + loss scaling
(train/s1_engine.py).  Lightning is not installable here, so there is no reference-generated fixture for this mode: the
float16 run is held against the library's own float32 run of the same micro-batches (itself pinned to the reference's
goldens at 1e-3, tests/test_s1_c3_gpu.py), and the scaler protocol (scale -> accumulate four micro-batches -> unscale ->
skip on overflow -> update) against its definition."""
import os

import pytest
import torch
import yaml

from util_fill import fill_module, s1_batch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _engine(gpu, dtype, **kw):
    from easevoice_trainer_amd.train.s1_engine import S1Engine

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    cfg["model"]["dropout"] = 0.0
    torch.manual_seed(0)
    eng = S1Engine(cfg, gpu, dtype, **kw)
    fill_module(eng.model, 3)
    eng.bank.mark_dirty()
    eng.model.eval()
    return eng


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def test_s1_fp16_clean_window_follows_fp32(gpu):
    b = {k: v.to(gpu) for k, v in s1_batch(2, 64, 192).items()}
    res = {}
    for dtype in (torch.float32, torch.float16):
        eng = _engine(gpu, dtype, **(dict(scaler_args=dict(init_scale=2.0 ** 10)) if dtype == torch.float16 else {}))
        p0 = eng.arena.param.detach().clone()
        losses, stepped = [], []
        for i in range(5):                      # the optimiser steps on micro-batch 4 (batch_idx > 0 and % 4 == 0)
            loss, _acc, st = eng.micro_step(b, i)
            losses.append(float(loss))
            stepped.append(bool(st))
            if i == 3:
                torch.cuda.synchronize()
                g = eng.arena.grad.detach().float().clone() / (eng.scaler.get_scale() if eng.scaler.enabled else 1.0)
        torch.cuda.synchronize()
        res[dtype] = dict(losses=losses, stepped=stepped, grad=g, delta=(eng.arena.param.detach() - p0).clone(),
                          skipped=eng.skipped_steps, scale=eng.scaler.get_scale(), tracker=int(eng.scaler._tracker.item()))
        del eng
        torch.cuda.empty_cache()
    a, h = res[torch.float32], res[torch.float16]
    assert a["stepped"] == h["stepped"] == [False] * 4 + [True]
    assert h["skipped"] == 0 and h["scale"] == 2.0 ** 10 and h["tracker"] == 1
    for x, y in zip(h["losses"], a["losses"]):
        assert abs(x - y) <= 5e-3 * abs(y), (x, y)
    # the accumulated (unscaled) gradient of four micro-batches and the ScaledAdam update it leads to
    assert _cos(h["grad"], a["grad"]) > 0.995, _cos(h["grad"], a["grad"])
    assert abs(float(h["grad"].norm()) / float(a["grad"].norm()) - 1.0) < 2e-2
    # ScaledAdam's first update is sign-like (every element moves by +-lr * rms): elements whose gradient sits inside the
    # half-precision noise flip sides, so the update's cosine is far below the gradient's (measured 0.886 against 0.995+)
    assert float(a["delta"].abs().max()) > 0 and _cos(h["delta"], a["delta"]) > 0.8, _cos(h["delta"], a["delta"])


def test_s1_fp16_overflow_skips_the_step(gpu):
    # (2**16, torch's default, still overflows here: the cross-entropy is SUM-reduced, so d loss / d logit reaches -1 and the
    # scaled -65536 is beyond half's 65504 -- the reference's first Lightning step under 16-mixed backs off for the same reason)
    eng = _engine(gpu, torch.float16, scaler_args=dict(init_scale=2.0 ** 40, backoff_factor=2.0 ** -28, growth_interval=1))
    b = {k: v.to(gpu) for k, v in s1_batch(2, 64, 192).items()}
    p0 = eng.arena.param.detach().clone()
    for i in range(5):
        _loss, _acc, st = eng.micro_step(b, i)
    torch.cuda.synchronize()
    # a loss scaled by 2**40 cannot be differentiated in half precision: the window's step is skipped, nothing moved, the
    # gradients were dropped, the scale backed off
    assert st and eng.skipped_steps == 1 and eng.optimizer.step_count == 0
    assert torch.equal(eng.arena.param, p0) and float(eng.arena.grad.abs().max()) == 0.0
    assert eng.scaler.get_scale() == 2.0 ** 12 and int(eng.scaler._tracker.item()) == 0
    for i in range(5, 9):                       # the next window at 2**12 is clean: the optimiser steps, the scale grows
        _loss, _acc, st = eng.micro_step(b, i)
    torch.cuda.synchronize()
    assert st and eng.skipped_steps == 1 and eng.optimizer.step_count == 1
    assert not torch.equal(eng.arena.param, p0) and torch.isfinite(eng.arena.param).all()
    assert eng.scaler.get_scale() == 2.0 ** 13


def test_scaler_state_dict_is_torchs(gpu):
    """DeviceGradScaler.state_dict() has torch.amp.GradScaler's layout (what Lightning stores under its AMP plugin's name in
    an s1 resume file, what a caller of the s2 engine can hand to torch's scaler), and round-trips"""
    from easevoice_trainer_amd.runtime import DeviceGradScaler

    ref = torch.amp.GradScaler("cpu", init_scale=512.0, growth_interval=7)
    ref.scale(torch.zeros(1))
    sc = DeviceGradScaler(gpu, init_scale=512.0, growth_interval=7)
    assert sc.state_dict() == ref.state_dict()
    sc.load_state_dict(dict(ref.state_dict(), scale=2048.0, _growth_tracker=3))
    assert sc.get_scale() == 2048.0 and int(sc._tracker.item()) == 3
    ref.load_state_dict(sc.state_dict())
    assert ref.get_scale() == 2048.0
    assert DeviceGradScaler(gpu, enabled=False).state_dict() == {}
