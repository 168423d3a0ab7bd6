"""GPU: the s2 step with its independent sub-models on branch streams (hip/disc.py: the six sub-discriminators dealt onto two
streams in the D step and in the generator step's discriminator node, EVT_MPD_STREAMS; the prior encoder beside posterior
encoder / flow / vocoder, EVT_ENC_STREAM; two of the three parallel ResBlocks of the wide vocoder stages beside the third,
EVT_DEC_STREAM) must train exactly like the one-stream step.

Same launches on the same operands: what can go wrong is ORDER -- a branch reading an operand before its producer stream
wrote it, a joined stream reading a branch's result early, a block handed out again while another stream still reads it.
Checked: the gradients of the first step, loss terms / gradient norms / parameters over several steps, eager and graph replay
(the replay is where the branches actually overlap)."""
import pytest
import torch

from test_s2_book_pipe_gpu import _batch, _engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graphs", [0, 1], ids=["eager", "graphs"])
def test_branch_streams_train_like_one_stream(gpu, graphs):
    from easevoice_trainer_amd.hip import disc as HD

    args, kw = _batch(gpu)
    hist, final, grads1 = {}, {}, {}
    old = (HD.MPD_STREAMS, HD.ENC_STREAM, HD.DEC_STREAM)
    try:
        for mode in ("one", "one_again", "branches"):
            HD.MPD_STREAMS, HD.ENC_STREAM, HD.DEC_STREAM = (2, True, True) if mode == "branches" else (1, False, False)
            eng = _engine(gpu, False)
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
            del eng
    finally:
        HD.MPD_STREAMS, HD.ENC_STREAM, HD.DEC_STREAM = old
    # the one-stream step against itself gives the noise floor (fp32 atomics of a few gradient kernels)
    for a, b, c in zip(grads1["branches"], grads1["one"], grads1["one_again"]):
        noise = ((c - b).abs().max() / b.abs().max()).item()
        rel = ((a - b).abs().max() / b.abs().max()).item()
        assert b.abs().max() > 0 and rel <= max(3.0 * noise, 1e-5), (rel, noise)
    assert torch.isfinite(hist["branches"]).all(), hist["branches"]
    noise = ((hist["one_again"] - hist["one"]).abs() / (hist["one"].abs() + 1e-6)).max(dim=0).values
    rel = ((hist["branches"] - hist["one"]).abs() / (hist["one"].abs() + 1e-6)).max(dim=0).values
    assert (rel <= torch.maximum(3.0 * noise, torch.full_like(noise, 1e-3))).all(), (rel, noise)
    for a, b in zip(final["branches"], final["one"]):
        d = (a - b).abs().max().item()
        assert d < 5e-3, d                   # five AdamW updates of lr 1e-4: identical up to sign flips of near-zero gradients
