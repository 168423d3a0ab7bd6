"""GPU parity of the s2 input front-end (csrc/frontend.hip, hip/frontend.py): the layout change of the spectrogram / ssl
features, the zero-padded 1025-bin projection enc_q.pre, the frozen quantizer look-up and the target-side mel, each
against the reference's formula evaluated in fp32 with torch on the CPU (models.py:348-352,912-926, core_vq.py:172-190,
mel_processing.py:77-90)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 1025, 100, 1088), (3, 768, 77, 768), (1, 5, 3, 64)])
def test_ncl_to_nlc_is_transpose_cast_pad(gpu, shape, dtype):
    from easevoice_trainer_amd.hip.frontend import ncl_to_nlc

    B, C, T, Cp = shape
    x = torch.randn(B, C, T, generator=torch.Generator().manual_seed(C + T)).to(gpu)
    y = ncl_to_nlc(x, Cp, dtype)
    ref = F.pad(x.transpose(1, 2), (0, Cp - C)).to(dtype)
    assert y.shape == (B, T, Cp) and y.dtype == dtype and y.is_contiguous()
    assert torch.equal(y, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_padded_spectrogram_projection_matches_linear(gpu, dtype):
    """enc_q.pre (Conv1d 1025 -> 192, k = 1) on the zero-padded channels-last spectrogram: forward, weight and bias
    gradients against F.linear on the CPU; the parameter keeps the reference's [192, 1025, 1] shape"""
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.hip.frontend import ncl_to_nlc
    from easevoice_trainer_amd.module.attentions import PaddedInPointwise, pointwise

    B, T = 3, 120
    torch.manual_seed(1)
    m = pointwise(1025, 192).to(gpu)
    assert isinstance(m, PaddedInPointwise) and tuple(m.weight.shape) == (192, 1025, 1) and m.cin == 1088
    with torch.no_grad():
        m.weight.copy_(m.weight.to(dtype).float())
    bank = HC.WeightBank(m, dtype, gpu)
    bank.build_tables()
    bank.fold()
    spec = torch.rand(B, 1025, T, device=gpu) * 3
    x = ncl_to_nlc(spec, m.cin, dtype)
    wgt = torch.randn(B, T, 192, device=gpu)
    y = m(x)
    (y.float() * wgt).sum().backward()
    bank.grads()
    torch.cuda.synchronize()
    w = m.weight.detach().cpu().squeeze(-1).requires_grad_(True)
    b = m.bias.detach().cpu().clone().requires_grad_(True)
    xr = spec.transpose(1, 2).to(dtype).float().cpu()
    ref = F.linear(xr, w, b)
    (ref * wgt.cpu()).sum().backward()
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4

    def rel(a, r):
        return ((a.detach().float().cpu() - r).abs().max() / (r.abs().max() + 1e-9)).item()

    assert rel(y, ref.detach()) < tol
    assert rel(m.weight.grad.squeeze(-1), w.grad) < tol
    assert rel(m.bias.grad, b.grad) < tol


@pytest.mark.parametrize("rate", ["25hz", "50hz"])
def test_rvq_lookup_matches_reference_formula(gpu, rate):
    """ssl_proj + EuclideanCodebook.quantize / dequantize (core_vq.py:172-190) + the x2 nearest up-sampling at 25 Hz
    (models.py:923-926): codes identical to the fp32 formula on the CPU, code vectors exact"""
    from easevoice_trainer_amd.hip.frontend import RvqEncoder

    B, T, D, K = 3, 101, 768, 1024
    k = 2 if rate == "25hz" else 1
    g = torch.Generator().manual_seed(4)
    w = (torch.randn(D, D, k, generator=g) * (D * k) ** -0.5).to(gpu)
    b = (torch.randn(D, generator=g) * 0.1).to(gpu)
    embed = torch.randn(K, D, generator=g).to(gpu)
    ssl = torch.randn(B, D, T, generator=g).to(gpu)
    enc = RvqEncoder(lambda: w, lambda: b, lambda: embed, D, K, k, k, gpu)
    h = enc.project(ssl)
    q, codes = enc.lookup(h, k)
    hr = F.conv1d(ssl.cpu(), w.cpu(), b.cpu(), stride=k).transpose(1, 2)
    assert h.shape == hr.shape
    assert ((h.cpu() - hr).abs().max() / hr.abs().max()).item() < 1e-4
    # the look-up itself is checked on the GPU's own h (a last-bit difference in h can flip a near-tie)
    x = h.cpu().reshape(-1, D)
    e = embed.cpu().t()
    dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ e + e.pow(2).sum(0, keepdim=True))
    ind = dist.max(dim=-1).indices.view(B, -1)
    agree = (codes.cpu() == ind).float().mean().item()
    assert agree >= 0.999, agree
    top2 = dist.topk(2, dim=-1).values
    ties = ((top2[:, 0] - top2[:, 1]).view(B, -1) < 1e-3 * top2[:, 0].abs().view(B, -1))
    assert bool(((codes.cpu() == ind) | ties).all())                      # any disagreement is a rounding-level tie
    assert torch.equal(q, F.embedding(codes, embed).repeat_interleave(k, dim=1))
    assert q.shape == (B, (T // k) * k if k == 2 else T, D)
    # the images follow an in-place change of the codebook (load_state_dict / k-means initialisation)
    embed.mul_(-1.0)
    _q2, codes2 = enc.lookup(h, k)
    dist2 = -(x.pow(2).sum(1, keepdim=True) + 2 * x @ e + e.pow(2).sum(0, keepdim=True))
    assert (codes2.cpu() == dist2.max(dim=-1).indices.view(B, -1)).float().mean().item() >= 0.999


def test_spec_to_mel_full_and_sliced(gpu):
    """mel_processing.py:77-90 on the whole spectrogram and on the training segments only (sovits.py:478-480)"""
    from easevoice_trainer_amd.module import commons, mel_processing as PM

    B, Fb, T = 4, 1025, 150
    g = torch.Generator().manual_seed(2)
    spec = (torch.rand(B, Fb, T, generator=g) * 5).to(gpu)
    spec[:, :, -3:] = 0.0                                                 # silent frames: the 1e-5 clamp
    from oracle.melbank import slaney_mel
    basis = torch.from_numpy(slaney_mel(32000, 2048, 128, 0.0, None))      # the ORACLE's filterbank, not the product's own
    ref = torch.log(torch.clamp(torch.matmul(basis, spec.cpu()), min=1e-5))
    mel = PM.spec_to_mel_torch(spec, 2048, 128, 32000, 0.0, None)
    assert mel.shape == (B, 128, T)
    assert (mel.cpu() - ref).abs().max().item() < 1e-4
    ids = torch.tensor([0, 7, 118, 60], device=gpu)
    got = PM.spec_to_mel_slices(spec, ids, 32, 2048, 128, 32000, 0.0, None)
    want = commons.slice_segments(ref.transpose(1, 2), ids.cpu(), 32).transpose(1, 2)
    assert got.shape == (B, 128, 32)
    assert (got.cpu() - want).abs().max().item() < 1e-4
