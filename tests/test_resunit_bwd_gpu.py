"""GPU parity of the fused backward of a HiFi-GAN ResBlock step (csrc/resunit_bwd.hip through hip/conv.py::resunit_bwd;
reference: torch.autograd over src/easevoice/module/modules.py:299-308): dx, both weight-gradient images and both bias
gradients of ONE launch against (a) the four launches it replaces (evt_conv1d_bwd_data x 2, evt_conv1d_bwd_weight x 2)
and (b) the CPU oracle (oracle/ops.py::res_unit differentiated by torch in fp32 on the same bf16-rounded operands);
the stage-mean scale folded into the load; run-to-run bit identity of the gradients."""
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu
HALF = torch.bfloat16      # tests/test_fp16_ops_gpu.py re-runs this module's cases with torch.float16

CASES = [(C, k, d, L) for C in (16, 32) for (k, d, L) in
         [(3, 1, 200), (3, 3, 64), (3, 5, 1000), (7, 1, 333), (7, 3, 640), (7, 5, 129), (11, 1, 130), (11, 3, 2048),
          (11, 5, 777)]]


def _setup(gpu, C_, k, d, Lq, nseq=2):
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.module.models import get_padding

    torch.manual_seed(C_ * 1000 + k * 10 + d)
    m = torch.nn.ModuleList([HC.EvtConv1d(C_, C_, k, dilation=d, padding=get_padding(k, d), weight_norm=True),
                             HC.EvtConv1d(C_, C_, k, dilation=1, padding=get_padding(k, 1), weight_norm=True)]).to(gpu)
    with torch.no_grad():
        for c in m:
            c.weight_g.mul_(torch.rand_like(c.weight_g) + 0.5)
            c.bias.normal_(0, 0.2)
    bank = HC.WeightBank(m, HALF, gpu)
    bank.build_tables()
    bank.fold()
    x = torch.randn(nseq, Lq, C_, device=gpu).to(HALF)
    dy = torch.randn(nseq, Lq, C_, device=gpu).to(HALF)
    return HC, m, bank, x, dy


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-9)).item()


def _where(a, b):
    """where the largest difference sits (diagnostic text for a failing run)"""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    d = (a - b).abs()
    i = torch.nonzero(d == d.max())[0].tolist()
    bad = torch.nonzero(d > 0.05 * b.abs().max())
    return f"max |diff| {d.max().item():.4g} at {i} (ref {b[tuple(i)].item():.4g}, got {a[tuple(i)].item():.4g}); " \
           f"{bad.size(0)} of {d.numel()} off by > 5 %; first bad {bad[:6].tolist()}"


def _unfused(HC, L, m, bank, xa, mid_a, dy, slope):
    """the four launches of ResUnitFn.backward before the fused kernel; returns dx, (dw1, dw2, db1, db2)"""
    s1, s2 = m[0]._slot, m[1]._slot
    nseq, lin = xa.size(0), xa.size(1)
    bank.zero_dw()
    for c in m:
        c.weight_v.grad.zero_()
        c.weight_g.grad.zero_()
        c.bias.grad.zero_()                  # in place: the bank's tables hold this tensor's address
    defer, bank.defer_n = bank.defer_n, 0
    HC._bwd_weight(s2, mid_a, dy, None, nseq, lin, 1.0, L.ACT_NONE, 1.0)
    dmid = HC._bwd_data(s2, dy, None, mid_a, None, nseq, lin, slope, L.ACT_NONE, 1.0)
    HC._bwd_weight(s1, xa, dmid, None, nseq, lin, 1.0, L.ACT_NONE, 1.0)
    dx = HC._bwd_data(s1, dmid, None, xa, dy, nseq, lin, slope, L.ACT_NONE, 1.0)
    bank.defer_n = defer
    bank.grads()
    torch.cuda.synchronize()
    g = {n_: p_.grad.clone() for c_i, c in enumerate(m) for n_, p_ in ((f"{c_i}.{n}", p) for n, p in c.named_parameters())}
    return dx, dmid, g


def _fused(HC, m, bank, xa, mid_a, dy, slope, scale=1.0):
    s1, s2 = m[0]._slot, m[1]._slot
    bank.zero_dw()
    for c in m:
        c.weight_v.grad.zero_()
        c.weight_g.grad.zero_()
        c.bias.grad.zero_()
    dx = HC.resunit_bwd(s1, s2, dy, xa, mid_a, slope, scale)
    assert dx is not None, "the fused backward must cover this case"
    bank.grads()
    torch.cuda.synchronize()
    g = {n_: p_.grad.clone() for c_i, c in enumerate(m) for n_, p_ in ((f"{c_i}.{n}", p) for n, p in c.named_parameters())}
    return dx, g


@pytest.mark.parametrize("case", CASES)
def test_fused_backward_vs_unfused_launches(gpu, case):
    from easevoice_trainer_amd.hip import lib as L
    from easevoice_trainer_amd.module.models import LRELU_SLOPE

    C_, k, d, Lq = case
    HC, m, bank, x, dy = _setup(gpu, C_, k, d, Lq)
    s1, s2 = m[0]._slot, m[1]._slot
    xa = HC._lrelu(x, LRELU_SLOPE)
    mid_a = HC._fwd(s1, xa, None, 1.0, L.ACT_LRELU, LRELU_SLOPE)
    dx_u, dmid_u, g_u = _unfused(HC, L, m, bank, xa, mid_a, dy, LRELU_SLOPE)
    dx_f, g_f = _fused(HC, m, bank, xa, mid_a, dy, LRELU_SLOPE)
    assert _rel(dx_f, dx_u) < 1e-2, "dx: " + _where(dx_f, dx_u)
    for n_ in g_u:
        assert _rel(g_f[n_], g_u[n_]) < 5e-3, f"{n_}: " + _where(g_f[n_], g_u[n_])
    # twice the same launch: the same bits (block-ordered partial rows, no atomics)
    dx_f2, g_f2 = _fused(HC, m, bank, xa, mid_a, dy, LRELU_SLOPE)
    assert torch.equal(dx_f, dx_f2)
    for n_ in g_f:
        assert torch.equal(g_f[n_], g_f2[n_]), n_


@pytest.mark.parametrize("case", [(16, 11, 5, 777), (16, 3, 1, 200), (32, 7, 3, 640), (32, 3, 5, 1000)])
def test_fused_backward_scale_on_load(gpu, case):
    """dy_scale = 1/3 (the stage mean, models.py:466) folded into the load = the same launch on a pre-scaled bf16 dy"""
    from easevoice_trainer_amd.hip import lib as L
    from easevoice_trainer_amd.module.models import LRELU_SLOPE

    C_, k, d, Lq = case
    HC, m, bank, x, dy = _setup(gpu, C_, k, d, Lq)
    xa = HC._lrelu(x, LRELU_SLOPE)
    mid_a = HC._fwd(m[0]._slot, xa, None, 1.0, L.ACT_LRELU, LRELU_SLOPE)
    pre = (dy.float() * (1.0 / 3.0)).to(HALF)
    dx_a, g_a = _fused(HC, m, bank, xa, mid_a, pre, LRELU_SLOPE)
    dx_b, g_b = _fused(HC, m, bank, xa, mid_a, dy, LRELU_SLOPE, 1.0 / 3.0)
    assert torch.equal(dx_a, dx_b)
    for n_ in g_a:
        assert torch.equal(g_a[n_], g_b[n_]), n_


@pytest.mark.parametrize("case", [(16, 11, 5, 777), (16, 7, 3, 64), (32, 7, 5, 129), (32, 3, 1, 200), (32, 11, 3, 500)])
def test_fused_backward_vs_oracle(gpu, case):
    """against torch.autograd over the oracle's res_unit in fp32 on the same bf16-rounded operands (weights folded and
    rounded as the bank does, the intermediate rounded to bf16 with the kernel's own leaky-relu branch decisions)"""
    import torch.nn.functional as F

    from easevoice_trainer_amd.hip import lib as L
    from easevoice_trainer_amd.module.models import LRELU_SLOPE, get_padding

    C_, k, d, Lq = case
    HC, m, bank, x, dy = _setup(gpu, C_, k, d, Lq)
    xa = HC._lrelu(x, LRELU_SLOPE)
    mid_a = HC._fwd(m[0]._slot, xa, None, 1.0, L.ACT_LRELU, LRELU_SLOPE)
    dx_f, g_f = _fused(HC, m, bank, xa, mid_a, dy, LRELU_SLOPE)
    xo = x.float().cpu().transpose(1, 2).requires_grad_(True)
    po = [{n_: p_.detach().cpu().clone().requires_grad_(True) for n_, p_ in c.named_parameters()} for c in m]
    ws = []
    for q in po:
        w = O.weight_norm_fold(q["weight_v"], q["weight_g"])
        ws.append(w + (w.detach().to(HALF).float() - w.detach()))
    h = F.conv1d(F.leaky_relu(xo, LRELU_SLOPE), ws[0], po[0]["bias"], padding=get_padding(k, d), dilation=d)
    gate = torch.where(mid_a.float().cpu().transpose(1, 2) > 0, 1.0, LRELU_SLOPE)
    h = h * gate
    h = h + (h.detach().to(HALF).float() - h.detach())
    yo = xo + F.conv1d(h, ws[1], po[1]["bias"], padding=get_padding(k, 1))
    yo.backward(dy.float().cpu().transpose(1, 2))
    assert _rel(dx_f.transpose(1, 2), xo.grad) < 3e-2, "dx: " + _where(dx_f.transpose(1, 2), xo.grad)
    for ci, q in enumerate(po):
        for n_ in q:
            assert _rel(g_f[f"{ci}.{n_}"], q[n_].grad) < 3e-2, f"{ci}.{n_}: " + _where(g_f[f"{ci}.{n_}"], q[n_].grad)


@pytest.mark.parametrize("case", [(16, 333), (32, 200), (16, 64), (32, 1000)])
def test_grouped_stage_vs_block_by_block(gpu, case):
    """one HiFi-GAN stage (three ResBlock1 of kernel sizes 3 / 7 / 11, dilations 1 / 3 / 5, averaged: models.py:457-466)
    through the grouped launches (hip/conv.py::ResStageFn) against the same modules run block by block, step by step
    (ResUnitFn + Add3ScaleFn): output, input gradient, every parameter gradient; and the grouped path twice: same bits"""
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.module.models import LRELU_SLOPE, ResBlock1

    C_, Lq = case
    torch.manual_seed(C_ + Lq)
    blocks = torch.nn.ModuleList([ResBlock1(C_, k, (1, 3, 5)) for k in (3, 7, 11)]).to(gpu)
    with torch.no_grad():
        for p_ in blocks.parameters():
            if p_.dim() == 1:
                p_.normal_(0, 0.2)
    bank = HC.WeightBank(blocks, HALF, gpu)
    bank.build_tables()
    bank.fold()
    x = torch.randn(2, Lq, C_, device=gpu).to(HALF)
    dy = torch.randn(2, Lq, C_, device=gpu).to(HALF)

    def run(fused):
        bank.zero_dw()
        for p_ in blocks.parameters():
            if p_.grad is not None:
                p_.grad.zero_()
        xg = x.clone().requires_grad_(True)
        if fused:
            y = HC.res_stage(xg, list(blocks), LRELU_SLOPE, 1.0 / 3.0)
            assert y is not None, "the grouped path must cover this stage"
        else:
            rs = [b(xg) for b in blocks]
            y = HC.Add3ScaleFn.apply(rs[0], rs[1], rs[2], 1.0 / 3.0)
        y.backward(dy)
        bank.grads()
        torch.cuda.synchronize()
        return y.detach(), xg.grad, {n_: p_.grad.clone() for n_, p_ in blocks.named_parameters()}

    y_u, dx_u, g_u = run(False)
    y_f, dx_f, g_f = run(True)
    assert _rel(y_f, y_u) < 1e-2, "y: " + _where(y_f, y_u)
    assert _rel(dx_f, dx_u) < 2e-2, "dx: " + _where(dx_f, dx_u)
    for n_ in g_u:
        assert _rel(g_f[n_], g_u[n_]) < 1e-2, f"{n_}: " + _where(g_f[n_], g_u[n_])
    y_2, dx_2, g_2 = run(True)
    assert torch.equal(y_f, y_2) and torch.equal(dx_f, dx_2)
    for n_ in g_f:
        assert torch.equal(g_f[n_], g_2[n_]), n_


WIDE = [(C, k, d, L) for C in (64, 128) for (k, d, L) in
        [(3, 1, 200), (3, 5, 1000), (7, 3, 640), (7, 5, 129), (11, 1, 130), (11, 3, 2048), (11, 5, 777), (11, 5, 64)]]


@pytest.mark.parametrize("case", WIDE)
def test_wide_fused_step_vs_unfused_launches(gpu, case):
    """C = 64 / 128 (csrc/resunit_wide.hip through ResUnitFn): forward (xa, mid_a, y) and backward (dx, every parameter
    gradient) of the one-launch-each-way path against the leaky-relu + convolution launches it replaces"""
    from easevoice_trainer_amd.hip import lib as L
    from easevoice_trainer_amd.module.models import LRELU_SLOPE

    C_, k, d, Lq = case
    HC, m, bank, x, dy = _setup(gpu, C_, k, d, Lq)
    s1, s2 = m[0]._slot, m[1]._slot
    assert HC._resunit_wide_params(s1, s2, x, LRELU_SLOPE) is not None, "the wide fused path must cover this case"
    xa_u = HC._lrelu(x, LRELU_SLOPE)
    mid_u = HC._fwd(s1, xa_u, None, 1.0, L.ACT_LRELU, LRELU_SLOPE)
    y_u = HC._fwd(s2, mid_u, x, 1.0, L.ACT_NONE, 1.0)
    dx_u, dmid_u, g_u = _unfused(HC, L, m, bank, xa_u, mid_u, dy, LRELU_SLOPE)

    bank.zero_dw()
    for c in m:
        c.weight_v.grad.zero_()
        c.weight_g.grad.zero_()
        c.bias.grad.zero_()
    xg = x.clone().requires_grad_(True)
    y = HC.res_unit(xg, m[0], m[1], LRELU_SLOPE)
    xa_f, mid_f = y.grad_fn.saved_tensors
    y.backward(dy)
    bank.grads()
    torch.cuda.synchronize()
    assert torch.equal(xa_f, xa_u), "lrelu(x)"
    assert _rel(mid_f, mid_u) < 1e-2, "mid_a: " + _where(mid_f, mid_u)
    assert _rel(y, y_u) < 1e-2, "y: " + _where(y, y_u)
    # the unfused backward used ITS mid_a; leaky-relu branch decisions can differ where the two intermediates round to
    # opposite sides of zero, so dx / gradients are compared loosely here and exactly-gated against the oracle below
    assert _rel(xg.grad, dx_u) < 3e-2, "dx: " + _where(xg.grad, dx_u)
    for n_ in g_u:
        got = dict((f"{ci}.{n}", p_.grad) for ci, c in enumerate(m) for n, p_ in c.named_parameters())[n_]
        assert _rel(got, g_u[n_]) < 3e-2, f"{n_}: " + _where(got, g_u[n_])


@pytest.mark.parametrize("case", [(64, 11, 5, 777), (64, 3, 1, 200), (128, 7, 3, 640), (128, 11, 5, 333)])
def test_wide_fused_step_vs_oracle(gpu, case):
    import torch.nn.functional as F

    from easevoice_trainer_amd.module.models import LRELU_SLOPE, get_padding

    C_, k, d, Lq = case
    HC, m, bank, x, dy = _setup(gpu, C_, k, d, Lq)
    bank.zero_dw()
    xg = x.clone().requires_grad_(True)
    y = HC.res_unit(xg, m[0], m[1], LRELU_SLOPE)
    xa_f, mid_f = y.grad_fn.saved_tensors
    y.backward(dy)
    bank.grads()
    torch.cuda.synchronize()
    xo = x.float().cpu().transpose(1, 2).requires_grad_(True)
    po = [{n_: p_.detach().cpu().clone().requires_grad_(True) for n_, p_ in c.named_parameters()} for c in m]
    ws = []
    for q in po:
        w = O.weight_norm_fold(q["weight_v"], q["weight_g"])
        ws.append(w + (w.detach().to(HALF).float() - w.detach()))
    h = F.conv1d(F.leaky_relu(xo, LRELU_SLOPE), ws[0], po[0]["bias"], padding=get_padding(k, d), dilation=d)
    gate = torch.where(mid_f.float().cpu().transpose(1, 2) > 0, 1.0, LRELU_SLOPE)
    agree = ((h.detach() > 0) == (mid_f.float().cpu().transpose(1, 2) > 0))
    assert agree.float().mean().item() >= 0.999, agree.float().mean().item()
    h = h * gate
    h = h + (h.detach().to(HALF).float() - h.detach())
    yo = xo + F.conv1d(h, ws[1], po[1]["bias"], padding=get_padding(k, 1))
    yo.backward(dy.float().cpu().transpose(1, 2))
    assert _rel(y.transpose(1, 2), yo) < 3e-2, "y: " + _where(y.transpose(1, 2), yo)
    assert _rel(xg.grad.transpose(1, 2), xo.grad) < 3e-2, "dx: " + _where(xg.grad.transpose(1, 2), xo.grad)
    for ci, (c, q) in enumerate(zip(m, po)):
        for n_, p_ in c.named_parameters():
            assert _rel(p_.grad, q[n_].grad) < 3e-2, f"{ci}.{n_}: " + _where(p_.grad, q[n_].grad)
