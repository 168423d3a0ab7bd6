"""GPU: the s2 step with pipelined bookkeeping (S2Engine._phase_pipe: every sub-model's weight-norm gradient, AdamW update
and refold on a side stream right behind its backward) must train exactly like the step that does them after the backward.

Same kernels, same per-element arithmetic: what can go wrong is ORDER -- a piece updated before its gradient was complete, a
row refolded before its parameters were updated (or never), a range updated twice or not at all.  Checked here: the
gradients of the first step, loss terms / gradient norms / parameters over several steps (eager and graph replay), that the
bf16 images at the end of a step ARE the fold of the updated parameters, and that the ranges partition the arenas."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(gpu, pipe, seed=1234):
    from easevoice_trainer_amd.train.s2_engine import S2Engine

    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    hps["model"]["p_dropout"] = 0.0
    torch.manual_seed(seed)
    old = os.environ.get("EVT_BOOK_PIPE")
    os.environ["EVT_BOOK_PIPE"] = "1" if pipe else "0"
    try:
        eng = S2Engine(hps, gpu, torch.bfloat16)
    finally:
        if old is None:
            del os.environ["EVT_BOOK_PIPE"]
        else:
            os.environ["EVT_BOOK_PIPE"] = old
    assert eng.pipe == bool(pipe)
    for m in eng.net_g.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    cb = eng.net_g.quantizer.vq.layers[0]._codebook
    g = torch.Generator().manual_seed(7)
    cb.embed.copy_(torch.randn(cb.embed.shape, generator=g))
    cb.inited.fill_(1.0)
    eng.build_optimizers()
    return eng


def _batch(gpu):
    from easevoice_trainer_amd.module.mel_processing import spectrogram_torch

    B, T, Tt = 4, 100, 30
    g = torch.Generator().manual_seed(11)
    wav = (torch.rand(B, 1, T * 640, generator=g) - 0.5).to(gpu)
    ssl = torch.randn(B, 768, T, generator=g).to(gpu)
    text = torch.randint(0, 732, (B, Tt), generator=g).to(gpu)
    lengths = torch.full((B,), T, dtype=torch.long, device=gpu)
    tl = torch.full((B,), Tt, dtype=torch.long, device=gpu)
    spec = spectrogram_torch(wav.squeeze(1), 2048, 32000, 640, 2048)
    eps = torch.randn(B, 192, T, generator=g).to(gpu)
    ids = torch.tensor([3, 10, 40, 60], device=gpu)
    return (ssl, spec, lengths, wav, text, tl), dict(eps=eps, ids_slice=ids)


def test_pieces_partition_rows_and_arenas(gpu):
    eng = _engine(gpu, True)
    comp = eng._complement
    # discriminators: the sub-models' ranges / rows cover everything
    assert comp(eng._d_ranges, eng.rt_d.arena.numel) == []
    assert comp(eng._d_rows, eng.rt_d.bank._nrows) == []
    for ivs in (eng._d_ranges, eng._d_rows, [eng._dec_range, eng._fq_range], [eng._dec_rows, eng._fq_rows]):
        s = sorted(ivs)
        assert all(a[1] <= b[0] for a, b in zip(s, s[1:])), ivs           # disjoint
    # dec.cond: gradient rows with the vocoder, fold rows and parameters with the rest
    lo, hi = eng.rt_g.bank.rows_of([eng.net_g.dec.cond])
    assert eng._dec_rows[0] <= lo and hi <= eng._dec_rows[1]
    assert not (eng._dec_rows_fold[0] <= lo < eng._dec_rows_fold[1])
    clo, chi = eng.rt_g.arena.range_of_prefix("dec.cond.")
    assert chi <= eng._dec_range[0] or clo >= eng._dec_range[1]


@pytest.mark.parametrize("graphs", [0, 1], ids=["eager", "graphs"])
def test_pipelined_bookkeeping_trains_like_serial(gpu, graphs):
    args, kw = _batch(gpu)
    hist, final, grads1 = {}, {}, {}
    for mode in ("serial", "pipe"):
        eng = _engine(gpu, mode == "pipe")
        if graphs:
            eng.enable_graphs(warmup_steps=1)
        rows = []
        for it in range(5):
            out = eng.step(*args, **kw)
            if it == 0:
                grads1[mode] = (eng.rt_g.arena.grad.clone(), eng.rt_d.arena.grad.clone())
            rows.append([float(out.disc), float(out.gen), float(out.fm), float(out.mel), float(out.kl),
                         float(out.grad_sumsq_d), float(out.grad_sumsq_g)])
        hist[mode] = torch.tensor(rows)
        if graphs:
            assert any(e["graphs"] is not None for e in eng._graph_cache.values()), "no graph was captured"
        torch.cuda.synchronize()
        final[mode] = (eng.rt_g.arena.param.clone(), eng.rt_d.arena.param.clone())
        assert eng.optim_g.step_count == eng.optim_d.step_count == 5
        assert int(eng.optim_g._step_dev.item()) == int(eng.optim_d._step_dev.item()) == 5
        if mode == "pipe":
            # the images left by the step are the fold of the parameters as they are now -- for every row of both banks
            for rt in (eng.rt_g, eng.rt_d):
                reg, alt = rt.bank.reg_arena.clone(), rt.bank.alt_arena.clone()
                rt.bank.fold()
                torch.cuda.synchronize()
                assert torch.equal(reg, rt.bank.reg_arena) and torch.equal(alt, rt.bank.alt_arena)
    for a, b in zip(grads1["pipe"], grads1["serial"]):
        rel = ((a - b).abs().max() / b.abs().max()).item()
        assert rel < 1e-4, rel               # same sums; fp32 atomics of the few non-deterministic gradients
        assert b.abs().max() > 0
    assert torch.isfinite(hist["pipe"]).all(), hist["pipe"]
    rel = ((hist["pipe"] - hist["serial"]).abs() / (hist["serial"].abs() + 1e-6)).max(dim=0).values
    assert (rel[:5] < 3e-2).all() and (rel[5:] < 1e-1).all(), (rel, hist["serial"], hist["pipe"])
    for a, b in zip(final["pipe"], final["serial"]):
        d = (a - b).abs().max().item()
        assert d < 5e-3, d                   # five AdamW updates of lr 1e-4: identical up to sign flips of near-zero gradients


def test_pipe_engine_serves_do_opt_false_and_hooks(gpu):
    """do_opt=False / hook_after_d need the optimiser calls at their serial places: the engine falls back to the cut
    program for such a step, and the next pipelined step picks the images up where that one left them"""
    args, kw = _batch(gpu)
    eng = _engine(gpu, True)
    p0 = eng.rt_g.arena.param.clone()
    eng.step(*args, do_opt=False, **kw)
    assert torch.equal(p0, eng.rt_g.arena.param) and eng.optim_g.step_count == 0
    seen = []
    eng.step(*args, hook_after_d=lambda: seen.append(float(eng.rt_d.arena.grad.abs().max())), **kw)
    assert seen and seen[0] > 0 and eng.optim_g.step_count == 1
    out = eng.step(*args, **kw)
    assert eng.optim_g.step_count == 2 and torch.isfinite(out.gen_all)
    ref = _engine(gpu, False)
    ref.step(*args, do_opt=False, **kw)
    ref.step(*args, **kw)
    ref.step(*args, **kw)
    d = (ref.rt_g.arena.param - eng.rt_g.arena.param).abs().max().item()
    assert d < 3e-3, d
