"""GPU parity of the fused encoder glue (csrc/enc_ops.hip) against plain torch fp32:
out = LayerNorm(x + dropout(y)) * row_mask, its gradients, and the counter-keyed dropout stream."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, y, gamma, beta, lens, eps=1e-5):
    live = (torch.arange(x.size(1), device=x.device)[None, :] < lens[:, None]).unsqueeze(-1).to(x.dtype)
    return F.layer_norm(x + y, (x.size(-1),), gamma, beta, eps) * live


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 37, 192), (2, 5, 512), (2, 9, 768)])
def test_res_ln_no_dropout(gpu, dtype, tol, shape):
    from easevoice_trainer_amd.hip.enc import res_drop_ln

    B, T, C = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, C, generator=g).to(gpu, dtype)
    y = torch.randn(B, T, C, generator=g).to(gpu, dtype)
    gamma = (torch.rand(C, generator=g) + 0.5).to(gpu)
    beta = torch.randn(C, generator=g).to(gpu)
    lens = torch.tensor([T, max(1, T // 2), 1][:B], device=gpu, dtype=torch.int32)
    w = torch.randn(B, T, C, generator=g).to(gpu)
    xs = [t.detach().float().requires_grad_(True) for t in (x, y, gamma, beta)]
    (_ref(xs[0], xs[1], xs[2], xs[3], lens) * w).sum().backward()
    xg, yg = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    gg, bg = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    out = res_drop_ln(xg, yg, gg, bg, lens, 0.0, 1)
    (out.float() * w).sum().backward()
    ref = _ref(xs[0], xs[1], xs[2], xs[3], lens)

    def close(a, b, name):
        err = (a.float() - b.float()).abs().max().item() / (b.float().abs().max().item() + 1e-6)
        assert err < tol, f"{name}: {err}"

    close(out, ref, "out")
    close(xg.grad, xs[0].grad, "dx")
    close(yg.grad, xs[1].grad, "dy")
    close(gg.grad, xs[2].grad, "dgamma")
    close(bg.grad, xs[3].grad, "dbeta")


def test_res_ln_dropout_stream(gpu):
    from easevoice_trainer_amd.hip import enc as E

    B, T, C, p = 4, 50, 192, 0.25
    lens = torch.full((B,), T, device=gpu, dtype=torch.int32)
    gamma, beta = torch.ones(C, device=gpu), torch.zeros(C, device=gpu)
    E.seed_rng(gpu, 123)
    # x = 0, y = 1: x + drop(y) takes two values per row, LayerNorm keeps them apart -> the kept fraction is visible
    x, y = torch.zeros(B, T, C, device=gpu), torch.ones(B, T, C, device=gpu)
    o1 = E.res_drop_ln(x, y, gamma, beta, lens, p, 7)
    o2 = E.res_drop_ln(x, y, gamma, beta, lens, p, 7)
    assert torch.equal(o1, o2)                                    # same (seed, site): same mask
    kept = (o1 > 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.02, kept
    assert not torch.equal(o1, E.res_drop_ln(x, y, gamma, beta, lens, p, 8))    # another site: another stream
    E.bump_rng(gpu)
    assert not torch.equal(o1, E.res_drop_ln(x, y, gamma, beta, lens, p, 7))    # next step: another mask
    # backward regenerates the same mask: directional derivative along a random direction in y (fp32)
    g = torch.Generator().manual_seed(3)
    xr = torch.randn(B, T, C, generator=g).to(gpu)
    yr = torch.randn(B, T, C, generator=g).to(gpu).requires_grad_(True)
    w = torch.randn(B, T, C, generator=g).to(gpu)
    v = torch.randn(B, T, C, generator=g).to(gpu)
    gam = (torch.rand(C, generator=g) + 0.5).to(gpu)
    out = E.res_drop_ln(xr, yr, gam, beta, lens, p, 9)
    (out * w).sum().backward()
    h = 1e-2
    with torch.no_grad():
        fp = (E.res_drop_ln(xr, yr + h * v, gam, beta, lens, p, 9) * w).sum()
        fm = (E.res_drop_ln(xr, yr - h * v, gam, beta, lens, p, 9) * w).sum()
    num = ((fp - fm) / (2 * h)).item()
    ana = (yr.grad * v).sum().item()
    assert abs(num - ana) < 2e-2 * max(1.0, abs(ana)), (num, ana)
    # dropped elements get no gradient, kept ones are scaled by 1/(1-p): dy == dx * mult with mult in {0, 1/(1-p)}
    assert ((yr.grad == 0).float().mean().item() - p) < 0.03


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_wn_residual(gpu, dtype):
    from easevoice_trainer_amd.hip.enc import wn_residual, wn_residual_last

    B, T, H = 3, 41, 192
    g = torch.Generator().manual_seed(2)
    lens = torch.tensor([T, 20, 1], device=gpu, dtype=torch.int32)
    live = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).unsqueeze(-1).float()
    mk = lambda *s: torch.randn(*s, generator=g).to(gpu, dtype)
    x, rs, acc, rs_l = mk(B, T, H), mk(B, T, 2 * H), mk(B, T, H), mk(B, T, H)
    w1, w2, w3 = mk(B, T, H).float(), mk(B, T, H).float(), mk(B, T, H).float()
    ins = [t.clone().requires_grad_(True) for t in (x, rs, acc, rs_l)]
    xo, ao = wn_residual(ins[0], ins[1], ins[2], lens)
    lo = wn_residual_last(ins[3], ao, lens)
    ((xo.float() * w1).sum() + (lo.float() * w3).sum()).backward()
    ref = [t.detach().float().requires_grad_(True) for t in (x, rs, acc, rs_l)]
    xr = (ref[0] + ref[1][..., :H]) * live
    ar = ref[2] + ref[1][..., H:]
    lr = (ar + ref[3]) * live
    ((xr * w1).sum() + (lr * w3).sum()).backward()
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    for a, b, name in [(xo, xr, "x"), (lo, lr, "last")] + [(i.grad, r.grad, f"grad{k}") for k, (i, r) in enumerate(zip(ins, ref))]:
        err = (a.float() - b.float()).abs().max().item() / (b.float().abs().max().item() + 1e-6)
        assert err < tol, (name, err)
    # first layer: no accumulator yet
    xo2, ao2 = wn_residual(x, rs, None, lens)
    assert torch.equal(ao2, rs[..., H:].contiguous())


def test_relu_dropout(gpu):
    from easevoice_trainer_amd.hip import enc as E

    g = torch.Generator().manual_seed(4)
    x = torch.randn(8, 64, 256, generator=g).bfloat16().to(gpu)
    assert torch.equal(E.relu_dropout(x, 0.0, 3), torch.relu(x))
    E.seed_rng(gpu, 5)
    p = 0.1
    xr = x.clone().requires_grad_(True)
    y = E.relu_dropout(xr, p, 3)
    pos = x > 0
    kept = (y[pos] != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.01, kept
    assert torch.allclose(y[pos & (y != 0)].float(), (x[pos & (y != 0)].float() / (1 - p)), rtol=1e-2)
    assert (y[~pos] == 0).all()
    y.backward(torch.ones_like(y))
    # the gradient is non-zero exactly where the output is: same mask both ways
    assert torch.equal(xr.grad != 0, y != 0)
    assert torch.allclose(xr.grad[y != 0].float(), torch.full_like(xr.grad[y != 0].float(), 1 / (1 - p)), rtol=1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_wn_stack_node(gpu, dtype):
    """the WN stack as one autograd node (hip/wn.py::WNStackFn: in_layer conv, gate, res_skip conv, residual / skip
    bookkeeping per layer; chained gradient adds; one conditioning-gradient buffer) against oracle/s2_step.py::wn
    (modules.py:187-212 of the reference): output, dx, dg and every parameter gradient, ragged lengths."""
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.module.models import WN
    from oracle.s2_step import SD, wn

    torch.manual_seed(11)
    B, T, H, NL, GIN = 3, 77, 192, 4, 512
    m = WN(H, 5, 1, NL, gin_channels=GIN).to(gpu)
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.dim() == 1:
                p_.normal_(0, 0.05)
    bank = HC.WeightBank(m, dtype, gpu)
    bank.build_tables()
    bank.fold()
    lens = torch.tensor([T, 40, 5], device=gpu, dtype=torch.int32)
    live = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).float().unsqueeze(-1)
    x = (torch.randn(B, T, H, device=gpu) * live)
    g = torch.randn(B, GIN, device=gpu)
    wgt = torch.randn(B, T, H, device=gpu)
    if dtype in (torch.bfloat16, torch.float16):
        x, g = x.to(dtype).float(), g.to(dtype).float()
    x, g = x.detach(), g.detach()
    xg, gg = x.to(dtype).clone().requires_grad_(True), g.clone().requires_grad_(True)
    out = m(xg, live.to(dtype), g=gg, lens=lens)
    (out.float() * wgt).sum().backward()
    bank.grads()
    torch.cuda.synchronize()

    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict(keep_vars=True).items()}
    xr, gr = x.cpu().clone().requires_grad_(True), g.cpu().clone().requires_grad_(True)
    ref = wn(SD(sd), xr.transpose(1, 2), live.cpu().transpose(1, 2), gr.unsqueeze(-1), NL, hidden=H).transpose(1, 2)
    (ref * wgt.cpu()).sum().backward()
    tol = 2e-3 if dtype == torch.float32 else 6e-2

    def close(a, b, name):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-6)
        assert err < tol, f"{name}: rel err {err:.3e} (tol {tol})"

    close(out, ref, "out")
    close(xg.grad.float() * live, xr.grad * live.cpu(), "dx")
    close(gg.grad, gr.grad, "dg")
    for k, p_ in m.named_parameters():
        close(p_.grad, sd[k].grad, k)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_coupling_flip_equals_torch_composition(gpu, dtype):
    """the fused tail of a mean-only coupling layer + Flip (hip/enc.py::CouplingFlipFn) against the torch lines it replaces
    (models.py ResidualCouplingLayer.forward + Flip), values and both gradients, with the next layer's input as a second
    consumer; exact in fp32, bf16 only rounds the stored x0n / dstats"""
    from easevoice_trainer_amd.hip.enc import coupling_flip

    B, T, h = 3, 37, 96
    g = torch.Generator().manual_seed(3)
    lens = torch.tensor([37, 20, 1], dtype=torch.int32, device=gpu)
    mask = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).float().unsqueeze(-1)
    x = torch.randn(B, T, 2 * h, generator=g).to(gpu).requires_grad_(True)
    stats = torch.randn(B, T, h, generator=g).to(gpu).to(dtype).requires_grad_(True)
    wy = torch.randn(B, T, 2 * h, generator=g).to(gpu)
    wn = torch.randn(B, T, h, generator=g).to(gpu)

    def ref(x, stats):
        x0, x1 = torch.split(x, [h, h], dim=-1)
        st = (stats * mask.to(stats.dtype)).float()
        y = torch.flip(torch.cat([x0, st + x1 * mask], dim=-1), [-1])
        return y, y[..., :h].to(dtype)

    y_r, n_r = ref(x, stats)
    ((y_r * wy).sum() + (n_r.float() * wn).sum()).backward()
    gx_r, gs_r = x.grad.clone(), stats.grad.clone()
    x.grad = stats.grad = None
    y, n = coupling_flip(x, stats, lens, True)
    assert torch.equal(y, y_r) and torch.equal(n, n_r)
    ((y * wy).sum() + (n.float() * wn).sum()).backward()
    tol = 0 if dtype == torch.float32 else 2e-2
    assert (x.grad - gx_r).abs().max() <= tol * gx_r.abs().max() + 1e-6
    assert (stats.grad.float() - gs_r.float()).abs().max() <= tol * gs_r.float().abs().max() + 1e-6
    # without the second output (the block's last layer)
    x.grad = stats.grad = None
    y2, none = coupling_flip(x, stats, lens, False)
    assert none is None and torch.equal(y2, y_r)
    (y2 * wy).sum().backward()
    x0g = torch.autograd.grad((ref(x, stats)[0] * wy).sum(), x)[0]
    assert torch.allclose(x.grad, x0g, rtol=0, atol=1e-6)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-2)], ids=["f32", "bf16"])
def test_mish_and_glu_chains_equal_torch(gpu, dtype, tol):
    """the style encoder's fused element-wise chains (hip/enc.py::MishDropoutFn / GluDropoutResFn) against the torch lines
    they replace (models.py Mish + Dropout, Conv1dGLU.forward), dropout off: values and gradients"""
    from easevoice_trainer_amd.hip.enc import glu_dropout_res, mish_dropout

    g = torch.Generator().manual_seed(9)
    x = (torch.randn(4, 50, 128, generator=g) * 3).to(gpu).to(dtype).requires_grad_(True)
    w = torch.randn(4, 50, 128, generator=g).to(gpu)
    ref = (x.float() * torch.tanh(F.softplus(x.float())))
    (ref * w).sum().backward()
    gr = x.grad.clone()
    x.grad = None
    y = mish_dropout(x, 0.0, 3)
    assert y.dtype == dtype
    assert (y.float() - ref).abs().max() <= tol * ref.abs().max()
    (y.float() * w).sum().backward()
    assert (x.grad.float() - gr.float()).abs().max() <= tol * gr.float().abs().max() + 1e-6

    y32 = mish_dropout(x, 0.0, 3, torch.float32)                                  # fp32 output next to a bf16 operand
    assert y32.dtype == torch.float32 and (y32 - ref).abs().max() <= 2e-6 * ref.abs().max()
    x.grad = None
    (y32 * w).sum().backward()
    assert (x.grad.float() - gr.float()).abs().max() <= tol * gr.float().abs().max() + 1e-6

    h = torch.randn(4, 50, 256, generator=g).to(gpu).to(dtype).requires_grad_(True)
    res = torch.randn(4, 50, 128, generator=g).to(gpu).requires_grad_(True)       # fp32 residual stream
    x1, x2 = torch.split(h.float(), 128, dim=-1)
    ref = res.float() + x1 * torch.sigmoid(x2)
    (ref * w).sum().backward()
    gh, gres = h.grad.clone(), res.grad.clone()
    h.grad = res.grad = None
    y = glu_dropout_res(h, res, 0.0, 4)
    assert (y.float() - ref).abs().max() <= tol * ref.abs().max()
    (y.float() * w).sum().backward()
    assert (h.grad.float() - gh.float()).abs().max() <= tol * gh.float().abs().max() + 1e-6
    assert (res.grad.float() - gres.float()).abs().max() <= tol * gres.float().abs().max() + 1e-6


def test_mish_and_glu_dropout_masks(gpu):
    """dropout on: the kept fraction is 1 - p, kept values are scaled by 1 / (1 - p), and the backward regenerates the same
    mask (gradient zero exactly where the output was dropped)"""
    from easevoice_trainer_amd.hip.enc import glu_dropout_res, mish_dropout

    p = 0.25
    x = (torch.rand(8, 100, 128, device=gpu) + 0.5).requires_grad_(True)          # mish(x) != 0 everywhere
    y = mish_dropout(x, p, 11)
    full = mish_dropout(x.detach(), 0.0, 11)
    kept = y != 0
    assert abs(kept.float().mean().item() - (1 - p)) < 0.01
    assert torch.allclose(y[kept], full[kept] / (1 - p), rtol=1e-5)
    y.sum().backward()
    assert torch.equal(x.grad != 0, kept)
    h = (torch.rand(8, 100, 256, device=gpu) + 0.5).requires_grad_(True)
    res = torch.zeros(8, 100, 128, device=gpu)
    y = glu_dropout_res(h, res, p, 12)
    kept = y != 0
    assert abs(kept.float().mean().item() - (1 - p)) < 0.01
    y.sum().backward()
    assert torch.equal(h.grad[..., :128] != 0, kept) and torch.equal(h.grad[..., 128:] != 0, kept)
    y2 = glu_dropout_res(h.detach(), res, p, 13)                                   # another site: another mask
    assert not torch.equal(y2 != 0, kept)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)], ids=["f32", "bf16"])
def test_reparam_equals_torch_lines(gpu, dtype, tol):
    """posterior encoder tail (hip/enc.py::ReparamFn) against models.py PosteriorEncoder.forward's torch lines"""
    from easevoice_trainer_amd.hip.enc import reparam

    B, T, Cc = 3, 41, 192
    g = torch.Generator().manual_seed(2)
    lens = torch.tensor([41, 7, 30], dtype=torch.int32, device=gpu)
    mask = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).float().unsqueeze(-1)
    stats = (torch.randn(B, T, 2 * Cc, generator=g) * 0.7).to(gpu).to(dtype).requires_grad_(True)
    eps = torch.randn(B, T, Cc, generator=g).to(gpu)
    ws = [torch.randn(B, T, Cc, generator=g).to(gpu) for _ in range(3)]

    st = (stats * mask.to(dtype)).float()
    m_r, l_r = torch.split(st, Cc, dim=-1)
    z_r = (m_r + eps * torch.exp(l_r)) * mask
    ((z_r * ws[0]).sum() + (m_r * ws[1]).sum() + (l_r * ws[2]).sum()).backward()
    gr = stats.grad.clone()
    stats.grad = None
    z, m, logs = reparam(stats, eps, lens)
    for a, b in ((z, z_r), (m, m_r), (logs, l_r)):
        assert (a - b).abs().max() <= 1e-6 * b.abs().max() + 1e-6
    ((z * ws[0]).sum() + (m * ws[1]).sum() + (logs * ws[2]).sum()).backward()
    assert (stats.grad.float() - gr.float()).abs().max() <= tol * gr.float().abs().max() + 1e-6
    stats.grad = None
    z2, _, _ = reparam(stats, eps, lens)             # only z consumed: the other two gradients arrive as None
    (z2 * ws[0]).sum().backward()
    want = torch.autograd.grad((((torch.split((stats * mask.to(dtype)).float(), Cc, dim=-1)[0]
                                  + eps * torch.exp(torch.split((stats * mask.to(dtype)).float(), Cc, dim=-1)[1])) * mask)
                                * ws[0]).sum(), stats)[0]
    assert (stats.grad.float() - want.float()).abs().max() <= tol * want.float().abs().max() + 1e-6
