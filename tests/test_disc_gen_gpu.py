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
