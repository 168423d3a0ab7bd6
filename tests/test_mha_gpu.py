"""GPU parity of the fused attention core (csrc/mha.hip) against the ORACLE's restatement of
src/easevoice/module/attentions.py:243-292 (oracle/s2_step.py::mha, pinned to the reference's own outputs by
tests/test_oracle_cpu.py), evaluated in fp32 on the CPU: the windowed relative-position self-attention of the encoders,
the window-less cross-attention of MRTE (mrte_model.py:25-61) and the style encoder's self-attention
(modules.py:605-682), each in bf16 (MFMA kernels, 3e-2 on bf16-rounded inputs) and in fp32 (the north_star's 1e-3)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.bfloat16: 3e-2, torch.float16: 6e-3, torch.float32: 1e-3}
HALVES = (torch.bfloat16, torch.float16)


def _oracle_core(q, k, v, ek, ev, lens_q, lens_k, H, w, scale=None):
    """oracle/s2_step.py::mha with identity projections exposes the attention core.  q [B, Tq, C], k / v [B, Tk, C] fp32
    CPU tensors -> out [B, Tq, C] with query rows >= lens_q zeroed like the kernel writes them.  The oracle scales by
    1/sqrt(head width); another scale (the style encoder's 1/sqrt(d_model)) is folded into q."""
    from oracle.s2_step import SD, mha

    B, Tq, C = q.shape
    Tk = k.size(1)
    d = C // H
    if scale is not None:
        q = q * (scale * d ** 0.5)
    # one "input" carrying q | k | v side by side, selector projections pick the thirds
    eye, z = torch.eye(C), torch.zeros(C, C)
    sd = {"conv_q.weight": torch.cat([eye, z, z], 1).unsqueeze(-1), "conv_k.weight": torch.cat([z, eye, z], 1).unsqueeze(-1),
          "conv_v.weight": torch.cat([z, z, eye], 1).unsqueeze(-1), "conv_o.weight": eye.unsqueeze(-1),
          "conv_q.bias": torch.zeros(C), "conv_k.bias": torch.zeros(C), "conv_v.bias": torch.zeros(C),
          "conv_o.bias": torch.zeros(C)}
    if w is not None:
        sd["emb_rel_k"], sd["emb_rel_v"] = ek, ev
    live_q = (torch.arange(Tq)[None, :] < lens_q.cpu()[:, None]).float()
    live_k = (torch.arange(Tk)[None, :] < lens_k.cpu()[:, None]).float()
    mask = live_q[:, None, :, None] * live_k[:, None, None, :]                       # [B, 1, Tq, Tk]
    x = torch.cat([q, torch.zeros(B, Tq, 2 * C)], -1).transpose(1, 2)                 # the oracle's [B, C, T] layout
    c = torch.cat([torch.zeros(B, Tk, C), k, v], -1).transpose(1, 2)
    out = mha(SD(sd), x, c, mask, H, window=w)
    return out.transpose(1, 2) * live_q.unsqueeze(-1)


def _close(x, y, name, tol, shape):
    x, y = x.detach().float().cpu(), y.detach().float().cpu()
    err = (x - y).abs().max().item() / (y.abs().max().item() + 1e-6)
    assert err < tol, f"{name}: rel err {err:.3e} shape={shape}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("shape", [(2, 37, 2, 96), (3, 200, 2, 96), (2, 130, 4, 64), (1, 70, 2, 32)])
def test_relattn_parity(gpu, shape, dtype):
    """packed q | k | v projection + window-4 relative positions (the encoders' layers)"""
    from easevoice_trainer_amd.hip.enc import rel_attention

    B, T, H, D = shape
    w, C = 4, H * D
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B, T, 3 * C, generator=g) * 1.5).to(dtype).to(gpu)
    ek = (torch.randn(1, 2 * w + 1, D, generator=g) * D ** -0.5).to(gpu)
    ev = (torch.randn(1, 2 * w + 1, D, generator=g) * D ** -0.5).to(gpu)
    lens = torch.tensor([T, max(3, T // 2), 1][:B], device=gpu, dtype=torch.int32)
    wgt = torch.randn(B, T, C, generator=g).to(gpu)

    ref_in = [qkv.float().cpu().requires_grad_(True), ek.cpu().clone().requires_grad_(True),
              ev.cpu().clone().requires_grad_(True)]
    ref = _oracle_core(ref_in[0][..., :C], ref_in[0][..., C:2 * C], ref_in[0][..., 2 * C:], ref_in[1], ref_in[2], lens, lens,
                       H, w)
    (ref * wgt.cpu()).sum().backward()

    a, b, c = qkv.clone().requires_grad_(True), ek.clone().requires_grad_(True), ev.clone().requires_grad_(True)
    out = rel_attention(a, b, c, lens, H, w, 0.0, 1)
    assert out.dtype == dtype
    (out.float() * wgt).sum().backward()
    live = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).float().unsqueeze(-1)
    tol = TOL[dtype]
    _close(out, ref, "out", tol, shape)
    # gradients of padded rows: the reference lets a padded QUERY row attend uniformly (its output is discarded by the
    # caller's mask); compare live rows only
    _close(a.grad * live, ref_in[0].grad * live.cpu(), "dqkv", tol, shape)
    _close(b.grad, ref_in[1].grad, "demb_k", tol, shape)
    _close(c.grad, ref_in[2].grad, "demb_v", tol, shape)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("shape,scale", [((2, 200, 60, 4, 128), None), ((3, 70, 33, 4, 128), None), ((2, 45, 130, 2, 32), None),
                                         ((2, 150, 150, 2, 64), 128 ** -0.5), ((1, 64, 64, 2, 96), None)],
                         ids=["mrte-200x60", "mrte-ragged", "cross-d32", "style-d64", "plain-d96"])
def test_mha_core_no_window(gpu, shape, scale, dtype):
    """window-less attention: MRTE's cross-attention (queries = ssl frames, keys / values = phonemes, 4 heads x 128,
    separate lengths) and the style encoder's self-attention (2 heads x 64, logits scaled by 1/sqrt(d_model))"""
    from easevoice_trainer_amd.hip.enc import mha_core

    B, Tq, Tk, H, D = shape
    C = H * D
    g = torch.Generator().manual_seed(Tq * 7 + Tk)
    q = (torch.randn(B, Tq, C, generator=g) * 1.2).to(dtype).to(gpu)
    k = (torch.randn(B, Tk, C, generator=g) * 1.2).to(dtype).to(gpu)
    v = torch.randn(B, Tk, C, generator=g).to(dtype).to(gpu)
    lens_q = torch.tensor([Tq, max(2, Tq // 3), 5][:B], device=gpu, dtype=torch.int32)
    lens_k = torch.tensor([max(1, Tk - 7), Tk, 1][:B], device=gpu, dtype=torch.int32)
    wgt = torch.randn(B, Tq, C, generator=g).to(gpu)

    rq, rk, rv = (t.float().cpu().requires_grad_(True) for t in (q, k, v))
    ref = _oracle_core(rq, rk, rv, None, None, lens_q, lens_k, H, None, scale)
    (ref * wgt.cpu()).sum().backward()

    a, b, c = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = mha_core(a, b, c, lens_q, lens_k, H, 0.0, 3, scale if scale is not None else D ** -0.5)
    (out.float() * wgt).sum().backward()
    live_q = (torch.arange(Tq, device=gpu)[None, :] < lens_q[:, None]).float().unsqueeze(-1)
    live_k = (torch.arange(Tk, device=gpu)[None, :] < lens_k[:, None]).float().unsqueeze(-1)
    tol = TOL[dtype]
    _close(out, ref, "out", tol, shape)
    _close(a.grad * live_q, rq.grad * live_q.cpu(), "dq", tol, shape)
    _close(b.grad, rk.grad * live_k.cpu(), "dk", tol, shape)       # the kernel writes zeros for padded keys
    _close(c.grad, rv.grad * live_k.cpu(), "dv", tol, shape)
    assert float((b.grad.float() * (1 - live_k)).abs().max()) == 0.0 and float((c.grad.float() * (1 - live_k)).abs().max()) == 0.0
    assert float((out.float() * (1 - live_q)).abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_relattn_dropout_consistency(gpu, dtype):
    """out is linear in (V, Ev) for a fixed dropout mask: <dO, out(V', Ev')> == <dV, V'> + <dEv, Ev'> holds only if the
    forward and both backward kernels regenerate the same mask."""
    from easevoice_trainer_amd.hip import enc as E

    B, T, H, D, w, p = 2, 100, 2, 96, 4, 0.3
    C = H * D
    g = torch.Generator().manual_seed(9)
    E.seed_rng(gpu, 77)
    lens = torch.tensor([T, 61], device=gpu, dtype=torch.int32)
    qkv = torch.randn(B, T, 3 * C, generator=g).to(dtype).to(gpu).requires_grad_(True)
    ek = (torch.randn(1, 9, D, generator=g) * 0.1).to(gpu).requires_grad_(True)
    ev = (torch.randn(1, 9, D, generator=g) * 0.1).to(gpu).requires_grad_(True)
    d_o = torch.randn(B, T, C, generator=g).to(dtype).to(gpu)
    out = E.rel_attention(qkv, ek, ev, lens, H, w, p, 5)
    assert torch.equal(out, E.rel_attention(qkv, ek, ev, lens, H, w, p, 5))
    assert not torch.equal(out, E.rel_attention(qkv, ek, ev, lens, H, w, p, 6))
    out.backward(d_o)
    v2 = torch.randn(B, T, C, generator=g).to(dtype).to(gpu)
    ev2 = (torch.randn(1, 9, D, generator=g) * 0.1).to(gpu)
    qkv2 = torch.cat([qkv.detach()[..., :2 * C], v2], dim=-1).contiguous()
    with torch.no_grad():
        out2 = E.rel_attention(qkv2, ek.detach(), ev2, lens, H, w, p, 5)
    lhs = (d_o.float() * out2.float()).sum().item()
    rhs = (qkv.grad[..., 2 * C:].float() * v2.float()).sum().item() + (ev.grad * ev2).sum().item()
    assert abs(lhs - rhs) < (2e-2 if dtype in HALVES else 1e-3) * max(1.0, abs(lhs)), (lhs, rhs)
    # dropout really drops: the p = 0 result differs, and its mean magnitude is preserved (inverted scaling)
    with torch.no_grad():
        out0 = E.rel_attention(qkv.detach(), ek.detach(), ev.detach(), lens, H, w, 0.0, 5)
    assert not torch.equal(out0, out)
    assert abs(out.float().mean().item() - out0.float().mean().item()) < 0.05


def test_mha_core_dropout_same_mask_in_both_dtypes_and_paths(gpu):
    """the fp32 kernels and the bf16 kernels draw the SAME keep mask for (seed, site, b, h, i, j) -- with V = identity-like
    one-hot rows the output exposes the dropped probabilities directly"""
    from easevoice_trainer_amd.hip import enc as E

    B, Tq, Tk, H, D, p = 1, 48, 32, 1, 32, 0.4
    E.seed_rng(gpu, 5)
    q = torch.zeros(B, Tq, D, device=gpu)                       # uniform attention: every probability 1/Tk
    k = torch.zeros(B, Tk, D, device=gpu)
    v = torch.eye(Tk, D, device=gpu).unsqueeze(0)               # out[i][j] = dropped p[i][j]
    o32 = E.mha_core(q, k, v, None, None, H, p, 11, 1.0)
    o16 = E.mha_core(q.bfloat16(), k.bfloat16(), v.bfloat16(), None, None, H, p, 11, 1.0).float()
    keep32, keep16 = o32 > 0, o16 > 0
    assert torch.equal(keep32, keep16)
    frac = keep32.float().mean().item()
    assert abs(frac - (1 - p)) < 0.06, frac
    assert torch.allclose(o32[keep32], torch.full_like(o32[keep32], 1.0 / Tk / (1 - p)), rtol=1e-5)


@pytest.mark.parametrize("packed,dtype", [(False, torch.bfloat16), (True, torch.bfloat16), (False, torch.float32)],
                         ids=["three-launches", "packed-qkv", "f32"])
@pytest.mark.parametrize("shape", [(2, 37, 2, 96), (3, 200, 2, 96)])
def test_self_attention_block_parity(gpu, shape, packed, dtype):
    """MultiHeadAttention (q / k / v / o projections + relative attention core, attentions.py:179-292) through the fused
    node hip/enc.py::RelSelfAttnFn (three 1x1 conv launches + evt_mha_*; chained backward-data launches, fused bias
    gradients) against oracle/s2_step.py::mha on the same (bf16-rounded) weights and inputs: output, dx and every
    parameter gradient; bf16 at 3e-2, fp32 at the north_star's 1e-3."""
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
                p_.copy_(p_.to(dtype).float())          # the kernels see 16-bit weights: give the oracle the same values
            if n_.endswith("bias"):
                p_.normal_(0, 0.1)
                p_.copy_(p_.to(dtype).float())
    if packed:
        # inside a runtime the three projection weights are adjacent in the arena and run as ONE [3C, C] GEMM
        from easevoice_trainer_amd.runtime import ModelRuntime
        rt = ModelRuntime(m, dtype, gpu)
        rt.prepare()
        assert m._qkv_packed is not None and m._qkv_packed._slot is not None
        finish = rt.finish_grads
    else:
        bank = HC.WeightBank(m, dtype, gpu)
        bank.build_tables()
        bank.fold()
        assert m._qkv_packed is None
        finish = bank.grads
    x = (torch.randn(B, T, C, device=gpu) * 1.2).to(dtype)
    lens = torch.tensor([T, max(3, T // 2), 1][:B], device=gpu, dtype=torch.int32)
    live = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).float().unsqueeze(-1)     # [B, T, 1]
    x = (x.float() * live).to(dtype)
    wgt = torch.randn(B, T, C, device=gpu)

    xg = x.clone().requires_grad_(True)
    out = m(xg, xg, lens)
    ((out.float() * live) * wgt).sum().backward()
    finish()
    torch.cuda.synchronize()

    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict(keep_vars=True).items()}
    xr = x.float().cpu().requires_grad_(True)
    lv = live.cpu().squeeze(-1)
    mask = lv[:, None, :, None] * lv[:, None, None, :]
    ref = mha(SD(sd), xr.transpose(1, 2), xr.transpose(1, 2), mask, H, window=w).transpose(1, 2)
    ((ref * live.cpu()) * wgt.cpu()).sum().backward()

    def close(a, b, name, tol=TOL[dtype]):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-6)
        assert err < tol, f"{name}: rel err {err:.3e} shape={shape}"

    close(out.float() * live, ref * live.cpu(), "out")
    close(xg.grad.float() * live, xr.grad * live.cpu(), "dx")
    for k, p_ in m.named_parameters():
        if k == "conv_k.bias":
            # a constant added to every key shifts all scores of a query alike: the softmax does not see it, the exact
            # gradient is 0 and both sides hold rounding noise -- compare it with the size of the value-bias gradient
            assert p_.grad.abs().max().item() < (2e-2 if dtype in HALVES else 1e-4) * sd["conv_v.bias"].grad.abs().max().item(), k
            continue
        close(p_.grad, sd[k].grad, k)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_mrte_block_parity(gpu, dtype):
    """MRTE (mrte_model.py:9-61: c_pre / text_pre, 4-head cross-attention of the ssl frames over the phonemes, + ssl_enc
    + ge, c_post) through the product module -- three projection launches + the attention core node -- against
    oracle/s2_step.py's restatement (mha with the text x ssl mask) on the same weights: output and every gradient."""
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.module.models import MRTE
    from oracle.s2_step import SD, mha

    B, T, Tt = 3, 90, 41
    torch.manual_seed(3)
    m = MRTE().to(gpu)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.copy_(p_.to(dtype).float())
    bank = HC.WeightBank(m, dtype, gpu)
    bank.build_tables()
    bank.fold()
    lens = torch.tensor([T, 50, 7], device=gpu, dtype=torch.int32)
    tl = torch.tensor([Tt, 20, Tt], device=gpu, dtype=torch.int32)
    ym = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).float().unsqueeze(-1)
    tm = (torch.arange(Tt, device=gpu)[None, :] < tl[:, None]).float().unsqueeze(-1)
    y = (torch.randn(B, T, 192, device=gpu) * ym).to(dtype)
    t = (torch.randn(B, Tt, 192, device=gpu) * tm).to(dtype)
    ge = torch.randn(B, 512, device=gpu).to(dtype)
    wgt = torch.randn(B, T, 192, device=gpu)
    yg, tg, gg = (v.clone().requires_grad_(True) for v in (y, t, ge))
    with torch.autocast("cuda", dtype=dtype if dtype in HALVES else torch.bfloat16, enabled=dtype in HALVES):
        out = m(yg, ym, tg, tm, gg, lens, tl)
    ((out.float() * ym) * wgt).sum().backward()
    bank.grads()
    torch.cuda.synchronize()

    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict(keep_vars=True).items()}
    s = SD(sd)
    yr, tr_, gr = (v.float().cpu().requires_grad_(True) for v in (y, t, ge))
    ymc, tmc = ym.cpu().transpose(1, 2), tm.cpu().transpose(1, 2)                     # [B, 1, T]
    yc, tc = yr.transpose(1, 2), tr_.transpose(1, 2)
    attn_mask = tmc.unsqueeze(2) * ymc.unsqueeze(-1)
    ssl_enc = F.conv1d(yc * ymc, s["c_pre.weight"], s["c_pre.bias"])
    text_enc = F.conv1d(tc * tmc, s["text_pre.weight"], s["text_pre.bias"])
    x = mha(s.sub("cross_attention"), ssl_enc * ymc, text_enc * tmc, attn_mask, 4) + ssl_enc + gr.unsqueeze(-1)
    ref = F.conv1d(x * ymc, s["c_post.weight"], s["c_post.bias"]).transpose(1, 2)
    ((ref * ym.cpu()) * wgt.cpu()).sum().backward()

    tol = 4e-2 if dtype in HALVES else 1e-3
    _close(out.float() * ym, ref * ym.cpu(), "out", tol, (B, T, Tt))
    _close(yg.grad.float() * ym, yr.grad * ym.cpu(), "dssl", tol, (B, T, Tt))
    _close(tg.grad.float() * tm, tr_.grad * tm.cpu(), "dtext", tol, (B, T, Tt))
    _close(gg.grad, gr.grad, "dge", tol, (B, T, Tt))
    for k, p_ in m.named_parameters():
        if k == "cross_attention.conv_k.bias":      # exact gradient 0 (a constant shift of all scores of a query)
            continue
        _close(p_.grad, sd[k].grad, k, tol, (B, T, Tt))
