"""GPU parity of the fused relative-position attention (csrc/relattn.hip) against the ORACLE's restatement of
src/easevoice/module/attentions.py:243-292 (oracle/s2_step.py::mha, pinned to the reference's own outputs by
tests/test_oracle_cpu.py), evaluated in fp32 on the CPU on bf16-rounded inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(qkv, ek, ev, lens, H, w):
    """oracle/s2_step.py::mha (the pinned CPU restatement of MultiHeadAttention.attention, attentions.py:243-292) on
    the packed projection: identity / selector 1x1 projections expose the attention core; evaluated in fp32 on the CPU
    on the bf16-rounded inputs.  qkv [B, T, 3C] -> out [B, T, C] with rows >= len zeroed like the kernel writes them."""
    from oracle.s2_step import SD, mha

    B, T, C3 = qkv.shape
    C = C3 // 3
    eye = torch.eye(C)
    z = torch.zeros(C, C)
    sd = {"conv_q.weight": torch.cat([eye, z, z], 1).unsqueeze(-1), "conv_k.weight": torch.cat([z, eye, z], 1).unsqueeze(-1),
          "conv_v.weight": torch.cat([z, z, eye], 1).unsqueeze(-1), "conv_o.weight": eye.unsqueeze(-1),
          "conv_q.bias": torch.zeros(C), "conv_k.bias": torch.zeros(C), "conv_v.bias": torch.zeros(C),
          "conv_o.bias": torch.zeros(C), "emb_rel_k": ek, "emb_rel_v": ev}
    live = (torch.arange(T)[None, :] < lens.cpu()[:, None]).float()                  # [B, T]
    mask = live[:, None, :, None] * live[:, None, None, :]                           # [B, 1, T, T]
    x = qkv.transpose(1, 2)                                                           # the oracle's [B, C, T] layout
    out = mha(SD(sd), x, x, mask, H, window=w)                                        # [B, C, T]
    return out.transpose(1, 2) * live.unsqueeze(-1)


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

    ref_in = [qkv.float().cpu().requires_grad_(True), ek.cpu().clone().requires_grad_(True),
              ev.cpu().clone().requires_grad_(True)]
    ref = _reference(ref_in[0], ref_in[1], ref_in[2], lens, H, w)
    (ref * wgt.cpu()).sum().backward()

    a, b, c = qkv.clone().requires_grad_(True), ek.clone().requires_grad_(True), ev.clone().requires_grad_(True)
    out = rel_attention(a, b, c, lens, H, w, 0.0, 1)
    (out.float() * wgt).sum().backward()
    live = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).float().unsqueeze(-1)

    def close(x, y, name, tol=3e-2):
        x, y = x.detach().float().cpu(), y.detach().float().cpu()
        err = (x - y).abs().max().item() / (y.abs().max().item() + 1e-6)
        assert err < tol, f"{name}: rel err {err:.3e} shape={shape}"

    close(out, ref, "out")
    # gradients of padded rows: the reference lets a padded QUERY row attend uniformly (its output is discarded by the
    # caller's mask); compare live rows only
    close(a.grad * live, ref_in[0].grad * live.cpu(), "dqkv")
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


@pytest.mark.parametrize("packed", [False, True], ids=["three-launches", "packed-qkv"])
@pytest.mark.parametrize("shape", [(2, 37, 2, 96), (3, 200, 2, 96)])
def test_self_attention_block_parity(gpu, shape, packed):
    """MultiHeadAttention (q / k / v / o projections + relative attention core, attentions.py:179-292) through the fused
    node hip/enc.py::RelSelfAttnFn (three 1x1 conv launches + evt_relattn_*; chained backward-data launches, fused bias
    gradients) against oracle/s2_step.py::mha on the same bf16-rounded weights and inputs: output, dx and every
    parameter gradient."""
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.module.attentions import MultiHeadAttention
    from oracle.s2_step import SD, mha

    B, T, H, D = shape
    w, C = 4, H * D
    torch.manual_seed(T)
    m = MultiHeadAttention(C, C, H, p_dropout=0.0, window_size=w).to(gpu)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if "conv" in n_:
                p_.copy_(p_.bfloat16().float())          # the kernels see bf16 weights: give the oracle the same values
            if n_.endswith("bias"):
                p_.normal_(0, 0.1)
                p_.copy_(p_.bfloat16().float())
    if packed:
        # inside a runtime the three projection weights are adjacent in the arena and run as ONE [3C, C] GEMM
        from easevoice_trainer_amd.runtime import ModelRuntime
        rt = ModelRuntime(m, torch.bfloat16, gpu)
        rt.prepare()
        assert m._qkv_packed is not None and m._qkv_packed._slot is not None
        finish = rt.finish_grads
    else:
        bank = HC.WeightBank(m, torch.bfloat16, gpu)
        bank.build_tables()
        bank.fold()
        assert m._qkv_packed is None
        finish = bank.grads
    x = (torch.randn(B, T, C, device=gpu) * 1.2).bfloat16()
    lens = torch.tensor([T, max(3, T // 2), 1][:B], device=gpu, dtype=torch.int32)
    live = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).float().unsqueeze(-1)     # [B, T, 1]
    x = (x.float() * live).bfloat16()
    wgt = torch.randn(B, T, C, device=gpu)

    xg = x.clone().requires_grad_(True)
    out = m(xg, xg, None, lens=lens)
    ((out.float() * live) * wgt).sum().backward()
    finish()
    torch.cuda.synchronize()

    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict(keep_vars=True).items()}
    xr = x.float().cpu().requires_grad_(True)
    lv = live.cpu().squeeze(-1)
    mask = lv[:, None, :, None] * lv[:, None, None, :]
    ref = mha(SD(sd), xr.transpose(1, 2), xr.transpose(1, 2), mask, H, window=w).transpose(1, 2)
    ((ref * live.cpu()) * wgt.cpu()).sum().backward()

    def close(a, b, name, tol=3e-2):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-6)
        assert err < tol, f"{name}: rel err {err:.3e} shape={shape}"

    close(out.float() * live, ref * live.cpu(), "out")
    close(xg.grad.float() * live, xr.grad * live.cpu(), "dx")
    for k, p_ in m.named_parameters():
        if k == "conv_k.bias":
            # a constant added to every key shifts all scores of a query alike: the softmax does not see it, the exact
            # gradient is 0 and both sides hold rounding noise -- compare it with the size of the value-bias gradient
            assert p_.grad.abs().max().item() < 2e-2 * sd["conv_v.bias"].grad.abs().max().item(), k
            continue
        close(p_.grad, sd[k].grad, k)
