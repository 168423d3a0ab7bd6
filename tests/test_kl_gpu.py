"""GPU: evt_masked_kl_fwd/bwd (csrc/losses.hip) against the reference formula of src/easevoice/module/losses.py:46-61
evaluated in fp64 on the CPU, in both layouts, with ragged lengths and mixed dtypes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(z_p, logs_q, m_p, logs_p, z_mask):
    """the reference's kl_loss, verbatim arithmetic in fp64; tensors [B, C, T], mask [B, 1, T]"""
    z_p, logs_q, m_p, logs_p, z_mask = (t.double() for t in (z_p, logs_q, m_p, logs_p, z_mask))
    kl = logs_p - logs_q - 0.5
    kl = kl + 0.5 * ((z_p - m_p) ** 2) * torch.exp(-2.0 * logs_p)
    return torch.sum(kl * z_mask) / torch.sum(z_mask)


@pytest.mark.parametrize("layout", ["channels_last_views", "bct_contiguous"])
@pytest.mark.parametrize("zdt,T", [(torch.float32, 173), (torch.bfloat16, 173), (torch.float32, 192)],
                         ids=["f32", "bf16", "f32-T_equals_C"])
def test_masked_kl_matches_reference(gpu, layout, zdt, T):
    """T == C (192 frames x 192 channels): the layout must come from the strides, not from the sizes"""
    from easevoice_trainer_amd.module.losses import kl_loss

    B, C = 5, 192
    g = torch.Generator().manual_seed(3)
    lens = torch.tensor([T, 1, 100, 64, T - 1])
    mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)            # [B, 1, T]
    base = [torch.randn(B, C, T, generator=g) * s for s in (1.0, 0.3, 1.0, 0.3)]
    base[0] = base[0].to(zdt).float()           # z_p in the compute dtype, exactly representable
    leaves = [t.clone().double().requires_grad_(True) for t in base]
    ref = _ref(*leaves, mask)
    ref.backward()
    dts = (zdt, torch.float32, torch.float32, torch.float32)
    if layout == "channels_last_views":
        dev = [t.transpose(1, 2).contiguous().to(gpu, dt).requires_grad_(True) for t, dt in zip(base, dts)]
        args = [t.transpose(1, 2) for t in dev]
    else:
        dev = [t.contiguous().to(gpu, dt).requires_grad_(True) for t, dt in zip(base, dts)]
        args = dev
    out = kl_loss(*args, mask.to(gpu), lens=lens.to(gpu))
    assert abs(float(out) - float(ref)) <= 1e-5 * abs(float(ref))
    (out * 3.0).backward()
    for i, (d, l) in enumerate(zip(dev, leaves)):
        got = d.grad.float().cpu()
        got = got.transpose(1, 2) if layout == "channels_last_views" else got
        want = 3.0 * l.grad.float()
        tol = 1e-5 if d.dtype == torch.float32 else 8e-3
        assert torch.allclose(got, want, rtol=tol, atol=tol * float(want.abs().max())), i
        assert not got[1, :, 1:].any()          # masked frames get exact zeros
    # without `lens` the mask sums recover the lengths
    out2 = kl_loss(*[a.detach() for a in args], mask.to(gpu))
    assert abs(float(out2) - float(ref)) <= 1e-5 * abs(float(ref))


def test_kl_loss_layouts_and_shape_errors(gpu):
    """mixed storage layouts (a channels-last view next to contiguous [B, C, T] tensors, slices of a wider tensor) give
    the value of the all-contiguous call; a mask that is not [B, 1, T] raises instead of being guessed at"""
    from easevoice_trainer_amd.hip.lib import EvtError
    from easevoice_trainer_amd.module.losses import kl_loss

    B, C, T = 2, 8, 8
    g = torch.Generator().manual_seed(1)
    base = [torch.randn(B, C, T, generator=g).to(gpu) * s for s in (1.0, 0.3, 1.0, 0.3)]
    mask = torch.ones(B, 1, T, device=gpu)
    want = float(kl_loss(*base, mask))
    cl = base[2].transpose(1, 2).contiguous().transpose(1, 2)                   # same values, channels-last storage
    wide = torch.cat([base[3], base[3]], dim=1)[:, :C]                          # a slice of a [B, 2C, T] tensor
    assert abs(float(kl_loss(base[0], base[1], cl, wide, mask)) - want) <= 1e-5 * abs(want)
    assert abs(float(kl_loss(*base, mask)) - float(_ref(*[t.cpu() for t in base], mask.cpu()))) <= 1e-5 * abs(want)
    with pytest.raises(EvtError):
        kl_loss(*base, torch.ones(B, T, device=gpu))
    with pytest.raises(EvtError):
        kl_loss(base[0], base[1][:, :, :4], base[2], base[3], mask)
