"""GPU: the fused input preparation of the discriminators (csrc/mpd_fold.hip through hip/disc.py) against the torch
composition it replaces (DiscriminatorP.prepare / DiscriminatorS.prepare: reflect pad, period view, dtype cast), forward
and backward; pure index arithmetic, so the comparison is exact."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
PERIODS = (1, 2, 3, 5, 7, 11)


def _ref_prepare(x, p, cd):
    n, t = x.shape
    if p == 1:
        return x.unsqueeze(-1).to(cd).contiguous()
    if t % p != 0:
        x = F.pad(x.unsqueeze(1), (0, p - (t % p)), "reflect").squeeze(1)
        t = x.size(1)
    return x.view(n, t // p, p).transpose(1, 2).reshape(n * p, t // p, 1).to(cd).contiguous()


@pytest.mark.parametrize("T", [20480, 1003, 77, 24])
@pytest.mark.parametrize("cd", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_fold_equals_torch_composition(gpu, T, cd):
    from easevoice_trainer_amd.hip.disc import mpd_fold

    g = torch.Generator().manual_seed(T)
    y = torch.randn(3, T, generator=g).to(gpu)
    yh = torch.randn(2, T, generator=g).to(gpu).to(cd)                  # generated audio arrives in the compute dtype
    outs = mpd_fold(PERIODS, cd, y, yh)
    both = torch.cat([y, yh.float()], dim=0)
    for o, p in zip(outs, PERIODS):
        ref = _ref_prepare(both, p, cd)
        assert o.shape == ref.shape and o.dtype == cd
        assert torch.equal(o, ref), (p, (o.float() - ref.float()).abs().max())
    single = mpd_fold(PERIODS, cd, y)
    for o, p in zip(single, PERIODS):
        assert torch.equal(o, _ref_prepare(y, p, cd))


@pytest.mark.parametrize("T", [20480, 1003, 24])
def test_unfold_is_the_gradient_of_the_composition(gpu, T):
    from easevoice_trainer_amd.hip.disc import MPDFoldFn, mpd_unfold

    g = torch.Generator().manual_seed(5 + T)
    n = 3
    x = torch.randn(n, T, generator=g).to(gpu).requires_grad_(True)
    refs = [_ref_prepare(x, p, torch.float32) for p in PERIODS]
    ws = [torch.randn(r.shape, generator=g).to(gpu) for r in refs]
    sum((r * w).sum() for r, w in zip(refs, ws)).backward()
    want = x.grad.clone()
    x.grad = None
    outs = MPDFoldFn.apply(x, PERIODS, torch.float32)
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    assert torch.allclose(x.grad, want, rtol=1e-6, atol=1e-6), (x.grad - want).abs().max()
    # the generated half of a [real ; generated] batch: rows of items b0 .. b0 + n - 1
    ws2 = [torch.cat([torch.zeros_like(w), w], dim=0) for w in ws]
    got = mpd_unfold(PERIODS, ws2, n, n, T, torch.float32)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-6)
    # a sub-discriminator without gradient (None) contributes zeros
    x.grad = None
    outs = MPDFoldFn.apply(x, PERIODS, torch.float32)
    (outs[2] * ws[2]).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    (_ref_prepare(x2, PERIODS[2], torch.float32) * ws[2]).sum().backward()
    assert torch.allclose(x.grad, x2.grad, rtol=1e-6, atol=1e-6)
