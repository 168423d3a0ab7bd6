"""GPU: the s2 step in the reference's fp16_run mode -- dtype torch.float16 on the IEEE-half build
of the library plus the device-side GradScaler -- against tests/golden/s2_c1_fp16.pt: the reference's own modules run
through the reference's loop body (src/train/sovits.py:459-525) under torch's float16 autocast with torch.amp.GradScaler
and torch.optim.AdamW for three steps (tests/golden/make_golden_fp16.py).  The fixture's scaler constants make the run
visit every branch: step 1 overflows (both optimisers skipped, scale 2**40 -> 2**8), steps 2 and 3 are clean (both step;
the scale grows to 2**9 after the second clean step).

Exact: the skip decisions (optimiser step counters), the scale and growth tracker after every step.  Within tolerance:
the five loss terms per step, the unscaled gradient norms, the generated waveform, the direction and size of the
parameter updates.  The scaler kernels on their own are compared with torch.amp.GradScaler element for element."""
import json
import os

import pytest
import torch

from util_fill import fill_module, s2_batch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _engine(gpu, gold, graphs):
    from easevoice_trainer_amd.module.mel_processing import spectrogram_torch
    from easevoice_trainer_amd.train.s2_engine import S2Engine

    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    hps["model"]["p_dropout"] = 0.0
    eng = S2Engine(hps, gpu, torch.float16, scaler_args=gold["scaler"])
    for m in eng.net_g.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    fill_module(eng.net_g, 1)
    fill_module(eng.net_d, 2)
    eng.build_optimizers()
    if graphs:
        eng.enable_graphs(warmup_steps=1)
    c = gold["config"]
    b = s2_batch(c["B"], c["T"], c["t_text"])
    wav = b["wav"].to(gpu)
    spec = spectrogram_torch(wav.squeeze(1), 2048, 32000, 640, 2048)
    args = (b["ssl"].to(gpu), spec, b["lengths"].to(gpu), wav, b["text"].to(gpu), b["text_lengths"].to(gpu))
    kw = dict(eps=b["eps"].to(gpu), ids_slice=b["ids_slice"].to(gpu))
    return eng, args, kw


def _cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


@pytest.mark.parametrize("graphs", [False, True], ids=["eager", "graph-replay"])
def test_fp16_three_steps_match_reference_autocast_gradscaler(gpu, graphs):
    gold = torch.load(os.path.join(HERE, "golden", "s2_c1_fp16.pt"), weights_only=False)
    eng, args, kw = _engine(gpu, gold, graphs)
    assert eng.scaler.enabled and eng.scaler.get_scale() == gold["scaler"]["init_scale"]
    p0 = {"g": eng.rt_g.arena.param.detach().clone(), "d": eng.rt_d.arena.param.detach().clone()}
    named_g, named_d = dict(eng.net_g.named_parameters()), dict(eng.net_d.named_parameters())
    p0_named = {n: p.detach().clone() for n, p in list(named_g.items()) + list(named_d.items())}
    report = []
    for i, ref in enumerate(gold["steps"]):
        out = eng.step(*args, **kw)
        torch.cuda.synchronize()
        got = dict(disc=float(out.disc), gen=float(out.gen), fm=float(out.fm), mel=float(out.mel), kl=float(out.kl))
        # ---- exact: skip decisions, scale trajectory ----
        steps_applied = dict(d=eng.optim_d.sync_step_count(), g=eng.optim_g.sync_step_count())
        assert steps_applied == ref["opt_steps"], (i, steps_applied, ref["opt_steps"])
        assert eng.scaler.get_scale() == ref["scale_after"], (i, eng.scaler.get_scale(), ref["scale_after"])
        assert int(eng.scaler._tracker.item()) == ref["growth_tracker"], i
        if ref["found_inf"]["d"] and ref["found_inf"]["g"]:
            # a skipped step changes nothing: parameters and both moment buffers are bit-identical
            assert torch.equal(eng.rt_g.arena.param, p0["g"]) and torch.equal(eng.rt_d.arena.param, p0["d"])
            assert float(eng.optim_g.exp_avg.abs().max()) == 0.0 and float(eng.optim_d.exp_avg_sq.abs().max()) == 0.0
        # ---- within tolerance: the loss terms (the discriminator has stepped before gen / fm are evaluated; from the
        # third step on every term has an AdamW update behind it) ----
        for k, v in got.items():
            r = ref["losses"][k]
            tol = 2e-2 if (i == 0 or (i == 1 and k in ("disc", "mel", "kl"))) else 6e-2
            report.append((i, k, round(v, 4), round(r, 4), round(abs(v - r) / abs(r), 5)))
            assert abs(v - r) <= tol * abs(r), (i, k, v, r)
        if not ref["found_inf"]["d"]:
            gd, gg = float(out.grad_sumsq_d), float(out.grad_sumsq_g)
            report.append((i, "grad_sumsq", gd, ref["grad_sumsq"]["d"], gg, ref["grad_sumsq"]["g"]))
            assert abs(gd - ref["grad_sumsq"]["d"]) <= 0.1 * ref["grad_sumsq"]["d"], (i, gd, ref["grad_sumsq"]["d"])
            assert abs(gg - ref["grad_sumsq"]["g"]) <= 0.1 * ref["grad_sumsq"]["g"], (i, gg, ref["grad_sumsq"]["g"])
        if i == 0:
            y = out.extras["y_hat"].float().squeeze(-1).squeeze(1).cpu()
            y = y.reshape(y.size(0), -1)[:, ::37]
            err = float((y - ref["y_hat"]).abs().max())
            report.append(("y_hat max abs err", err, "of", float(ref["y_hat"].abs().max())))
            assert err <= 2e-2 * float(ref["y_hat"].abs().max()) + 2e-3, err
    print("fp16 steps (step, term, got, reference, rel):", report)
    # ---- the updates of the two clean steps: direction per sampled tensor, size per top-level module ----
    worst = (2.0, "")
    for n, d_ref in list(gold["delta_g"].items()) + list(gold["delta_d"].items()):
        p = named_g[n] if n in named_g else named_d[n]
        d = (p.detach() - p0_named[n]).flatten()[:256]
        worst = min(worst, (_cos(d, d_ref), n))
    print("fp16 update direction, worst cosine:", worst)
    assert worst[0] > 0.8, worst
    tot = {}
    for n, p in named_g.items():
        top = n.split(".")[0]
        tot[top] = tot.get(top, 0.0) + float((p.detach() - p0_named[n]).double().pow(2).sum())
    for k, v in gold["delta_sumsq_g"].items():
        if v > 0:
            assert abs(tot[k] - v) <= 0.1 * v, (k, tot[k], v)
    dd = float(sum((p.detach() - p0_named[n]).double().pow(2).sum() for n, p in named_d.items()))
    assert abs(dd - gold["delta_sumsq_d"]) <= 0.1 * gold["delta_sumsq_d"]
    if graphs:
        assert eng.graph_steps["replayed"] >= 2, eng.graph_steps


def test_scaler_kernels_equal_torch_gradscaler(gpu):
    """evt_scaler_unscale / evt_adamw_flat_dev_guarded / evt_scaler_update on synthetic arenas against torch.amp.GradScaler +
    torch.optim.AdamW on the CPU over eight steps with overflows injected into one optimiser, the other, both, none:
    gradients after unscale_, skip decisions, parameters, scale and tracker are compared after every step."""
    from easevoice_trainer_amd.runtime import DeviceGradScaler, FlatAdamW, ParamArena

    torch.manual_seed(5)
    args = dict(init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2)

    def models():
        return torch.nn.Linear(37, 29), torch.nn.Linear(11, 13)

    ma, mb = models()
    ra, rb = models()
    ra.load_state_dict(ma.state_dict())
    rb.load_state_dict(mb.state_dict())
    arena_a, arena_b = ParamArena(ma, gpu), ParamArena(mb, gpu)

    class Owner:
        def __init__(self, arena):
            self.arena = arena

    oa, ob = Owner(arena_a), Owner(arena_b)
    opt_a = FlatAdamW(arena_a, [dict(names=[n for n, _ in ma.named_parameters()], lr=1e-2)], betas=(0.8, 0.99), eps=1e-9)
    opt_b = FlatAdamW(arena_b, [dict(names=[n for n, _ in mb.named_parameters()], lr=1e-2)], betas=(0.8, 0.99), eps=1e-9)
    ref_a = torch.optim.AdamW(ra.parameters(), 1e-2, betas=(0.8, 0.99), eps=1e-9)
    ref_b = torch.optim.AdamW(rb.parameters(), 1e-2, betas=(0.8, 0.99), eps=1e-9)
    sc = DeviceGradScaler(gpu, **args)
    ref = torch.amp.GradScaler("cpu", **args)
    g = torch.Generator().manual_seed(9)
    plan = ["none", "a", "none", "none", "b", "both", "none", "none"]
    for step, bad in enumerate(plan):
        ref.scale(torch.zeros(1))            # torch creates its scale tensor lazily in scale(), as a training loop would
        scale = ref.get_scale()
        assert sc.get_scale() == scale
        for (m, r) in ((ma, ra), (mb, rb)):
            for (n, p), q in zip(m.named_parameters(), r.parameters()):
                gr = torch.randn(q.shape, generator=g) * scale
                q.grad = gr.clone()
                p.grad.copy_(gr.to(gpu))
        if bad in ("a", "both"):
            ra.weight.grad[3, 5] = float("inf")
            ma.weight.grad[3, 5] = float("inf")
        if bad in ("b", "both"):
            rb.bias.grad[2] = float("nan")
            mb.bias.grad[2] = float("nan")
        ref.unscale_(ref_a)
        sc.unscale_(oa)
        if bad not in ("a", "both"):
            assert torch.allclose(ma.weight.grad.cpu(), ra.weight.grad, rtol=1e-6, atol=0), step
        ref.step(ref_a)
        opt_a.step(skip=sc.found_inf(oa))
        ref.unscale_(ref_b)
        sc.unscale_(ob)
        ref.step(ref_b)
        opt_b.step(skip=sc.found_inf(ob))
        ref.update()
        sc.update()
        torch.cuda.synchronize()
        assert sc.get_scale() == ref.get_scale(), (step, sc.get_scale(), ref.get_scale())
        assert int(sc._tracker.item()) == int(ref._growth_tracker.item()), step
        for (m, r) in ((ma, ra), (mb, rb)):
            for (n, p), q in zip(m.named_parameters(), r.parameters()):
                assert torch.allclose(p.detach().cpu(), q.detach(), rtol=2e-6, atol=1e-7), (step, n)
        assert float(sc.found_inf(oa)) == 0.0 and float(sc.found_inf(ob)) == 0.0      # cleared by update()
    assert opt_a.sync_step_count() == 6 and opt_b.sync_step_count() == 6       # 8 steps, 2 skipped each
