"""GPU parity of the fused relative-position attention (csrc/relattn.hip) against the torch formulation of
src/easevoice/module/attentions.py:214-292 evaluated in fp32 on bf16-rounded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(qkv, ek, ev, lens, H, w):
    """the product's torch path (module/attentions.py) on fp32 tensors; rows >= len zeroed like the kernel"""
    from easevoice_trainer_amd.module.attentions import MultiHeadAttention as M

    B, T, C3 = qkv.shape
    C = C3 // 3
    d = C // H
    q, k, v = [t.reshape(B, T, H, d).transpose(1, 2) for t in qkv.split(C, dim=-1)]
    live = (torch.arange(T, device=qkv.device)[None, :] < lens[:, None]).float()
    mask = (live[:, None, :, None] * live[:, None, None, :])
    helper = M.__new__(M)
    helper.window_size = w
    qs = q / math.sqrt(d)
    scores = qs @ k.transpose(-2, -1)
    qe = qs @ ek.unsqueeze(0).transpose(-2, -1)
    scores = scores + M._rel_to_abs(helper._band_to_full(qe, T))
    scores = scores.masked_fill(mask == 0, -1e4)
    p = F.softmax(scores, dim=-1)
    out = p @ v + helper._full_to_band(M._abs_to_rel(p), T) @ ev.unsqueeze(0)
    out = out.transpose(1, 2).reshape(B, T, C)
    return out * live.unsqueeze(-1)


@pytest.mark.parametrize("shape", [(2, 37, 2, 96), (3, 200, 2, 96), (2, 130, 4, 64), (1, 70, 2, 32)])
def test_relattn_parity(gpu, shape):
    from easevoice_trainer_amd.hip.enc import rel_attention

    B, T, H, D = shape
    w, C = 4, H * D
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B, T, 3 * C, generator=g) * 1.5).bfloat16().to(gpu)
    ek = (torch.randn(1, 2 * w + 1, D, generator=g) * D ** -0.5).to(gpu)
    ev = (torch.randn(1, 2 * w + 1, D, generator=g) * D ** -0.5).to(gpu)
    lens = torch.tensor([T, max(3, T // 2), 1][:B], device=gpu, dtype=torch.int32)
    wgt = torch.randn(B, T, C, generator=g).to(gpu)

    ref_in = [qkv.float().requires_grad_(True), ek.clone().requires_grad_(True), ev.clone().requires_grad_(True)]
    ref = _reference(ref_in[0], ref_in[1], ref_in[2], lens, H, w)
    (ref * wgt).sum().backward()

    a, b, c = qkv.clone().requires_grad_(True), ek.clone().requires_grad_(True), ev.clone().requires_grad_(True)
    out = rel_attention(a, b, c, lens, H, w, 0.0, 1)
    (out.float() * wgt).sum().backward()
    live = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).float().unsqueeze(-1)

    def close(x, y, name, tol=3e-2):
        err = (x.float() - y.float()).abs().max().item() / (y.float().abs().max().item() + 1e-6)
        assert err < tol, f"{name}: rel err {err:.3e} shape={shape}"

    close(out, ref, "out")
    # gradients of padded rows: the reference lets a padded QUERY row attend uniformly (its output is discarded by the
    # caller's mask); compare live rows only
    close(a.grad * live, ref_in[0].grad * live, "dqkv")
    close(b.grad, ref_in[1].grad, "demb_k")
    close(c.grad, ref_in[2].grad, "demb_v")


def test_relattn_dropout_consistency(gpu):
    """out is linear in (V, Ev) for a fixed dropout mask: <dO, out(V', Ev')> == <dV, V'> + <dEv, Ev'> holds only if the
    forward and both backward kernels regenerate the same mask."""
    from easevoice_trainer_amd.hip import enc as E

    B, T, H, D, w, p = 2, 100, 2, 96, 4, 0.3
    C = H * D
    g = torch.Generator().manual_seed(9)
    E.seed_rng(gpu, 77)
    lens = torch.tensor([T, 61], device=gpu, dtype=torch.int32)
    qkv = torch.randn(B, T, 3 * C, generator=g).bfloat16().to(gpu).requires_grad_(True)
    ek = (torch.randn(1, 9, D, generator=g) * 0.1).to(gpu).requires_grad_(True)
    ev = (torch.randn(1, 9, D, generator=g) * 0.1).to(gpu).requires_grad_(True)
    d_o = torch.randn(B, T, C, generator=g).bfloat16().to(gpu)
    out = E.rel_attention(qkv, ek, ev, lens, H, w, p, 5)
    assert torch.equal(out, E.rel_attention(qkv, ek, ev, lens, H, w, p, 5))
    assert not torch.equal(out, E.rel_attention(qkv, ek, ev, lens, H, w, p, 6))
    out.backward(d_o)
    v2 = torch.randn(B, T, C, generator=g).bfloat16().to(gpu)
    ev2 = (torch.randn(1, 9, D, generator=g) * 0.1).to(gpu)
    qkv2 = torch.cat([qkv.detach()[..., :2 * C], v2], dim=-1).contiguous()
    with torch.no_grad():
        out2 = E.rel_attention(qkv2, ek.detach(), ev2, lens, H, w, p, 5)
    lhs = (d_o.float() * out2.float()).sum().item()
    rhs = (qkv.grad[..., 2 * C:].float() * v2.float()).sum().item() + (ev.grad * ev2).sum().item()
    assert abs(lhs - rhs) < 2e-2 * max(1.0, abs(lhs)), (lhs, rhs)
    # dropout really drops: the p = 0 result differs, and its mean magnitude is preserved (inverted scaling)
    with torch.no_grad():
        out0 = E.rel_attention(qkv.detach(), ek.detach(), ev.detach(), lens, H, w, 0.0, 5)
    assert not torch.equal(out0, out)
    assert abs(out.float().mean().item() - out0.float().mean().item()) < 0.05
