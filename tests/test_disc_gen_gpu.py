"""GPU: the discriminators' part of the generator step as one autograd node (hip/disc.py: real and generated audio batched
through every convolution, backward-data over the generated half only with the feature-loss gradients added in the
epilogue) against the composition it replaces -- forward_single on both signals + feature_loss + generator_loss, i.e.
src/train/sovits.py:509-516 with src/easevoice/module/losses.py:7-15,35-43 -- which the golden fixtures pin."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 3e-2)], ids=["f32", "bf16"])
def test_generator_step_through_discriminators(gpu, dtype, tol):
    from easevoice_trainer_amd.module.losses import feature_loss, generator_loss
    from easevoice_trainer_amd.module.models import MultiPeriodDiscriminator
    from easevoice_trainer_amd.runtime import ModelRuntime

    torch.manual_seed(4)
    net_d = MultiPeriodDiscriminator(False)
    rt = ModelRuntime(net_d, dtype, gpu)
    rt.prepare()
    rt.bank.weight_grads = False
    n, T = 2, 20480
    y = (torch.rand(n, 1, T, device=gpu) - 0.5)
    y_hat = torch.tanh(torch.randn(n, 1, T, device=gpu) * 0.5)

    a = y_hat.clone().requires_grad_(True)
    with torch.no_grad():
        _, fmap_r = net_d.forward_single(y)
    logits_ref, fmap_g = net_d.forward_single(a)
    fm_ref, gen_ref = feature_loss(fmap_r, fmap_g), generator_loss(logits_ref)
    (gen_ref * 0.7 + fm_ref * 1.3).backward()

    b = y_hat.clone().requires_grad_(True)
    gen, fm, logits = net_d.generator_losses(y, b)
    (gen * 0.7 + fm * 1.3).backward()
    torch.cuda.synchronize()

    def rel(u, v):
        u, v = u.detach().float(), v.detach().float()
        return ((u - v).abs().max() / (v.abs().max() + 1e-12)).item()

    assert rel(gen, gen_ref) < tol and rel(fm, fm_ref) < tol, (float(gen), float(gen_ref), float(fm), float(fm_ref))
    for lo, lr in zip(logits, logits_ref):
        assert lo.shape == lr.shape and rel(lo, lr) < tol
    assert rel(b.grad, a.grad) < tol, rel(b.grad, a.grad)


@pytest.mark.parametrize("streams", [2, 3])
def test_sub_discriminators_on_branch_streams_give_the_same_bits(gpu, streams):
    """EVT_MPD_STREAMS: the six sub-discriminators dealt onto the current stream and side streams -- the same launches on the
    same operands, so losses, logits and the waveform gradient are bit-identical to the one-stream pass; repeated to give a
    missing wait or an early block reuse a chance to show (the second pass allocates into the blocks the first one freed)"""
    from easevoice_trainer_amd.hip import disc as HD
    from easevoice_trainer_amd.module.models import MultiPeriodDiscriminator
    from easevoice_trainer_amd.runtime import ModelRuntime

    torch.manual_seed(5)
    net_d = MultiPeriodDiscriminator(False)
    rt = ModelRuntime(net_d, torch.bfloat16, gpu)
    rt.prepare()
    rt.bank.weight_grads = False
    n, T = 4, 20480
    y = (torch.rand(n, 1, T, device=gpu) - 0.5)
    y_hat = torch.tanh(torch.randn(n, 1, T, device=gpu) * 0.5)

    def run(ns):
        old = HD.MPD_STREAMS
        HD.MPD_STREAMS = ns
        try:
            b = y_hat.clone().requires_grad_(True)
            gen, fm, logits = net_d.generator_losses(y, b)
            (gen * 0.7 + fm * 1.3).backward()
            torch.cuda.synchronize()
            return [gen.detach().clone(), fm.detach().clone(), b.grad.clone()] + [l.clone() for l in logits]
        finally:
            HD.MPD_STREAMS = old

    ref = run(1)
    for _ in range(4):
        got = run(streams)
        # (the feature loss is summed with fp32 atomics: its last bits differ from run to run on one stream as well)
        assert abs(float(got[1]) - float(ref[1])) <= 1e-5 * abs(float(ref[1]))
        for i, (u, v) in enumerate(zip(got, ref)):
            assert i == 1 or torch.equal(u, v), i


@pytest.mark.parametrize("streams", [2, 3])
def test_discriminator_step_on_branch_streams(gpu, streams):
    """the D step's batched pass (forward_batched + the batched LSGAN loss, sovits.py:497-507) with the sub-discriminators on
    branch streams: logits bit-identical, the weight gradients (queued from the branch streams, launched on the bank's side
    stream) those of the one-stream pass"""
    from easevoice_trainer_amd.hip import disc as HD
    from easevoice_trainer_amd.module.losses import discriminator_loss_batched
    from easevoice_trainer_amd.module.models import MultiPeriodDiscriminator
    from easevoice_trainer_amd.runtime import ModelRuntime

    torch.manual_seed(6)
    net_d = MultiPeriodDiscriminator(False)
    rt = ModelRuntime(net_d, torch.bfloat16, gpu)
    n, T = 4, 20480
    y = (torch.rand(n, 1, T, device=gpu) - 0.5)
    y_hat = torch.tanh(torch.randn(n, 1, T, device=gpu) * 0.5)

    def run(ns):
        old = HD.MPD_STREAMS
        HD.MPD_STREAMS = ns
        try:
            rt.zero_grad()
            rt.prepare()
            rt.bank.weight_grads = True
            outs = net_d.forward_batched(y, y_hat)
            loss = discriminator_loss_batched(outs)
            loss.backward()
            rt.finish_grads()
            torch.cuda.synchronize()
            return [o.detach().clone() for o in outs], rt.arena.grad.detach().clone(), float(loss.detach())
        finally:
            HD.MPD_STREAMS = old

    outs0, g0, l0 = run(1)
    _outs, g0b, _l = run(1)
    noise = float((g0b - g0).abs().max())            # kernels that add partial sums with atomics: run-to-run noise
    assert noise <= 1e-5 * float(g0.abs().max())
    for _ in range(4):
        outs, g, l = run(streams)
        for u, v in zip(outs, outs0):
            assert torch.equal(u, v)
        assert abs(l - l0) <= 1e-6 * abs(l0)
        assert float((g - g0).abs().max()) <= max(2.0 * noise, 1e-6 * float(g0.abs().max())), (float((g - g0).abs().max()), noise)
