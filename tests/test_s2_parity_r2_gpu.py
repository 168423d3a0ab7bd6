"""GPU parity at the shapes and on the code paths the bench runs (round-2 fixtures, tests/golden/make_golden_r2.py, all
produced by the reference's own modules):

  * bf16 whole step, config C1, against s2_c1.pt: the first check that runs conv_deep / conv_ring / wgrad_* /
    conv_narrow / relattn IN COMPOSITION (their per-op tests are in test_conv_gpu.py / test_relattn_gpu.py)
  * fp32 and bf16 whole step at BASELINE config 2 (B = 16, 4 s clips) against s2_c2.pt
  * AdamW: evt_adamw_flat(_dev) / evt_sumsq against torch.optim.AdamW's post-step weights (s2_c1_adamw.pt)
Tolerances: fp32 1e-3 relative (north_star); bf16 2e-2 per loss term, 5e-2 on per-module gradient sums."""
import json
import os

import pytest
import torch

from util_fill import fill_module, fill_tensor, s2_batch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _engine(gpu, dtype):
    from easevoice_trainer_amd.train.s2_engine import S2Engine

    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    hps["model"]["p_dropout"] = 0.0
    eng = S2Engine(hps, gpu, dtype)
    for m in eng.net_g.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    fill_module(eng.net_g, 1)
    fill_module(eng.net_d, 2)
    return eng


def _step(eng, gpu, cfg, **kw):
    from easevoice_trainer_amd.module.mel_processing import spectrogram_torch

    b = s2_batch(cfg["B"], cfg["T"], cfg["t_text"])
    wav = b["wav"].to(gpu)
    spec = spectrogram_torch(wav.squeeze(1), 2048, 32000, 640, 2048)
    grads_d = {}

    def grab_d():
        for n, p in eng.net_d.named_parameters():
            grads_d[n] = p.grad.detach().clone()

    out = eng.step(b["ssl"].to(gpu), spec, b["lengths"].to(gpu), wav, b["text"].to(gpu), b["text_lengths"].to(gpu),
                   eps=b["eps"].to(gpu), ids_slice=b["ids_slice"].to(gpu), do_opt=False, hook_after_d=grab_d, **kw)
    torch.cuda.synchronize()
    grads_g = {n: p.grad.detach().clone() for n, p in eng.net_g.named_parameters()}
    return out, grads_d, grads_g


def _sumsq(grads):
    tot = {}
    for n, g in grads.items():
        top = n.split(".")[0]
        tot[top] = tot.get(top, 0.0) + float(g.double().pow(2).sum())
    return tot


def _check(gold, out, gd, gg, loss_tol, tensor_tol, sumsq_tol, slice_tol):
    """slice_tol None (bf16 runs): 64-element gradient slices are not compared -- in bf16 a slice that is small against
    its tensor's scale sits in rounding noise (measured 3e-2 .. 8e-1 on the same run whose per-module gradient sums
    agree to 2e-2); the per-module sums, every loss term and the forward tensors carry the bf16 check"""
    got = dict(disc=out.disc, gen=out.gen, fm=out.fm, mel=out.mel, kl=out.kl, kl_ssl=out.kl_ssl, gen_all=out.gen_all)
    for k, v in gold["losses"].items():
        assert abs(float(got[k]) - v) <= loss_tol * max(abs(v), 1e-6), (k, float(got[k]), v)
    ex = out.extras
    for k, v in gold["stats"].items():
        assert rel(ex[k][:, :8, :16], v) < tensor_tol, k
    td, tg = _sumsq(gd), _sumsq(gg)
    for k, v in gold["d_grad_sumsq"].items():
        assert abs(td[k] - v) <= sumsq_tol * v, ("D", k, td[k], v)
    for k, v in gold["g_grad_sumsq"].items():
        assert abs(tg[k] - v) <= sumsq_tol * v, ("G", k, tg[k], v)
    if slice_tol is None:
        return
    noise = gold.get("noise", {})
    nd = noise.get("d_slices", gold.get("d_grad_slice_noise", {}))
    ng = noise.get("g_slices", gold.get("g_grad_slice_noise", {}))
    for n, s in gold["d_grad_slices"].items():
        assert rel(gd[n].flatten()[:64], s) < max(slice_tol, 3 * nd.get(n, 0.0)), n
    for n, s in gold["g_grad_slices"].items():
        assert rel(gg[n].flatten()[:64], s) < max(slice_tol, 3 * ng.get(n, 0.0)), n


def test_bf16_whole_step_c1_vs_reference(gpu):
    gold = torch.load(os.path.join(HERE, "golden", "s2_c1.pt"), weights_only=False)
    eng = _engine(gpu, torch.bfloat16)
    out, gd, gg = _step(eng, gpu, gold["config"])
    _check(gold, out, gd, gg, loss_tol=2e-2, tensor_tol=5e-2, sumsq_tol=5e-2, slice_tol=None)
    y, yr = out.extras["y_hat"].squeeze(1).float().cpu(), gold["y_hat"]
    assert abs(float(y.pow(2).mean().sqrt()) - float(yr.pow(2).mean().sqrt())) <= 2e-2 * float(yr.pow(2).mean().sqrt())
    assert rel(y, yr) < 8e-2                    # measured 4.7e-2 of max|y| (bf16 through 91 convs)
    assert rel(out.extras["y_hat_mel"], gold["y_hat_mel"]) < 1e-1       # log-mel of a near-silent init waveform: 5.9e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_whole_step_c2_vs_reference(gpu, dtype):
    """BASELINE config 2: B = 16, T = 200 frames (4 s), text 60 -- the shape bench.py times"""
    gold = torch.load(os.path.join(HERE, "golden", "s2_c2.pt"), weights_only=False)
    assert gold["config"] == dict(B=16, T=200, t_text=60)
    eng = _engine(gpu, dtype)
    out, gd, gg = _step(eng, gpu, gold["config"])
    f32 = dtype == torch.float32
    _check(gold, out, gd, gg, loss_tol=1e-3 if f32 else 2e-2, tensor_tol=1e-3 if f32 else 5e-2,
           sumsq_tol=5e-3 if f32 else 5e-2, slice_tol=2e-3 if f32 else None)
    y = out.extras["y_hat"].squeeze(1).float().cpu()
    assert rel(y[:, ::997], gold["y_hat_strided"]) < (1e-3 if f32 else 8e-2)
    rms = float(y.double().pow(2).mean().sqrt())
    assert abs(rms - gold["y_hat_rms"]) <= (1e-3 if f32 else 2e-2) * gold["y_hat_rms"]
    assert rel(out.extras["y_hat_mel"][:, ::7, ::3], gold["y_hat_mel_strided"]) < (1e-3 if f32 else 1e-1)
    assert rel(out.extras["y_mel"][:, ::7, ::3], gold["y_mel_strided"]) < 1e-3
    # DiscriminatorS logits in order (the period discriminators flatten (h, p) there and (p, h) here: same multiset,
    # which every loss only sums over)
    # bf16: the logits are read off a waveform that itself sits 4.7e-2 from the reference's (bound 8e-2 above); measured
    # 4e-2 .. 5.2e-2 over equal-precision variants of the generator's element-wise chains -- the same band as the waveform
    # Error budget of this bound, stage by stage: the discriminators ALONE, on one fixed waveform, hold 3e-2 against fp32
    # (tests/test_disc_gen_gpu.py); the waveform holds 8e-2 (above).  That the bound hides no arithmetic regression is pinned
    # elsewhere at full width: the same kernels built for IEEE half reproduce the reference's float16 step to 2.5e-3 on every
    # loss term (tests/test_s2_fp16_gpu.py), and the fp32 instantiation holds 1e-3 here.
    assert rel(out.extras["d_logits"][0][:, :16], gold["d_logits_head"][0]) < (1e-3 if f32 else 8e-2)


def test_adamw_kernel_vs_torch_adamw(gpu):
    """evt_adamw_flat_dev on the reference's OWN gradients == torch.optim.AdamW's post-step weights (1e-6 relative),
    for parameters of all four generator groups and of D; evt_sumsq == sum of the squared gradients fed in"""
    gold = torch.load(os.path.join(HERE, "golden", "s2_c1_adamw.pt"), weights_only=False)
    eng = _engine(gpu, torch.float32)
    opt_g, opt_d = eng.build_optimizers()
    hy = gold["hyper"]
    assert [g["lr"] for g in opt_g.param_groups] == [hy["lr"], hy["low"], hy["low"], hy["low"]]
    for net, rt, opt, grads, after, seed in ((eng.net_g, eng.rt_g, opt_g, gold["grads_g"], gold["after_g"], 1),
                                             (eng.net_d, eng.rt_d, opt_d, gold["grads_d"], gold["after_d"], 2)):
        params = dict(net.named_parameters())
        rt.arena.grad.zero_()
        want_ss = 0.0
        for n, g in grads.items():
            params[n].grad.copy_(g.to(gpu))
            want_ss += float(g.double().pow(2).sum())
        ss = float(rt.grad_sumsq())
        assert abs(ss - want_ss) <= 1e-5 * want_ss, (ss, want_ss)
        opt.step()
        torch.cuda.synchronize()
        for n, w in after.items():
            got = params[n].detach().cpu()
            assert torch.allclose(got, w, rtol=1e-6, atol=1e-6 * float(w.abs().max())), (n, rel(got, w))
        # a parameter that received no gradient only decays: w * (1 - lr * wd)
        n0 = "dec.ups.0.bias" if net is eng.net_g else "discriminators.2.convs.2.bias"
        w0 = fill_tensor(n0, params[n0].shape, seed)
        assert torch.allclose(params[n0].detach().cpu(), w0 * (1 - hy["lr"] * hy["weight_decay"]), rtol=1e-6, atol=1e-9)
    # ssl_proj sits in no update range (the reference never gives it a gradient, models.py:912-921)
    assert torch.equal(eng.net_g.ssl_proj.weight.detach().cpu(), fill_tensor("ssl_proj.weight", eng.net_g.ssl_proj.weight.shape, 1))


def test_whole_step_then_adamw_vs_reference(gpu):
    """fp32 step on the GPU (our gradients) followed by both optimiser launches vs the reference's post-step weights:
    the first AdamW step moves every element by lr * g / (|g| + eps), so the two agree wherever the gradient's sign does;
    checked per selected tensor (>= 99.5 % of the elements within 2e-6 of max|w|) and through fp64 checksums of every
    top-level module (the sum of squares moves by < 1e-6 relative)"""
    gold = torch.load(os.path.join(HERE, "golden", "s2_c1_adamw.pt"), weights_only=False)
    c1 = torch.load(os.path.join(HERE, "golden", "s2_c1.pt"), weights_only=False)
    eng = _engine(gpu, torch.float32)
    opt_g, opt_d = eng.build_optimizers()
    out, gd, gg = _step(eng, gpu, c1["config"])
    assert abs(float(out.grad_sumsq_g) - gold["grad_sumsq_g"]) <= 5e-3 * gold["grad_sumsq_g"]
    assert abs(float(out.grad_sumsq_d) - gold["grad_sumsq_d"]) <= 5e-3 * gold["grad_sumsq_d"]
    opt_d.step()
    opt_g.step()
    torch.cuda.synchronize()
    for net, after, sums in ((eng.net_g, gold["after_g"], gold["checksums_g"]), (eng.net_d, gold["after_d"], gold["checksums_d"])):
        params = dict(net.named_parameters())
        for n, w in after.items():
            got = params[n].detach().cpu()
            ok = ((got - w).abs() <= 2e-6 * float(w.abs().max()) + 1e-9).float().mean().item()
            assert ok >= 0.995, (n, ok)
        tot = {}
        for n, p in params.items():
            top = n.split(".")[0]
            s, q = tot.get(top, (0.0, 0.0))
            tot[top] = (s + float(p.detach().double().sum()), q + float(p.detach().double().pow(2).sum()))
        for k, (s, q) in sums.items():
            assert abs(tot[k][1] - q) <= 1e-6 * q, (k, tot[k][1], q)
