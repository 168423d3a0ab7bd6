"""GPU parity of the fused HiFi-GAN ResBlock step (csrc/resunit.hip through hip/conv.py::ResUnitFn):
y = x + c2(lrelu(c1(lrelu(x)))) (src/easevoice/module/modules.py:299-308) for the narrow vocoder stages, against
(a) the CPU oracle's convolutions (oracle/ops.py, fp32 on bf16-rounded inputs / folded weights) and (b) the three-launch
composition it replaces; forward outputs, the two saved activations, dx and all parameter gradients."""
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as O

pytestmark = pytest.mark.gpu
HALF = torch.bfloat16      # tests/test_fp16_ops_gpu.py re-runs this module's cases with torch.float16

CASES = [(C, k, d, L) for C in (16, 32) for (k, d, L) in
         [(3, 1, 200), (3, 5, 1000), (7, 1, 333), (7, 3, 64), (11, 1, 130), (11, 5, 777), (11, 3, 2048)]]


@pytest.mark.parametrize("case", CASES)
def test_fused_resblock_step(gpu, case):
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.hip import lib as L
    from easevoice_trainer_amd.module.models import LRELU_SLOPE, get_padding

    C_, k, d, Lq = case
    nseq = 2
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
    s1, s2 = m[0]._slot, m[1]._slot
    assert HC._resunit_params(s1, s2, x, LRELU_SLOPE) is not None, "the fused path must cover this case"

    # (b) the three launches it replaces
    xa_u = HC._lrelu(x, LRELU_SLOPE)
    mid_u = HC._fwd(s1, xa_u, None, 1.0, L.ACT_LRELU, LRELU_SLOPE)
    y_u = HC._fwd(s2, mid_u, x, 1.0, L.ACT_NONE, 1.0)

    xg = x.clone().requires_grad_(True)
    y = HC.res_unit(xg, m[0], m[1], LRELU_SLOPE)
    xa_f, mid_f = y.grad_fn.saved_tensors
    y.backward(dy)
    bank.grads()
    torch.cuda.synchronize()

    def rel(a, b):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        return ((a - b).abs().max() / (b.abs().max() + 1e-9)).item()

    assert torch.equal(xa_f, xa_u), "lrelu(x)"
    assert rel(mid_f, mid_u) < 1e-2 and rel(y, y_u) < 1e-2, (rel(mid_f, mid_u), rel(y, y_u))

    # (a) oracle, fp32 on the same bf16-rounded operands
    xo = x.float().cpu().transpose(1, 2).requires_grad_(True)
    po = [{n_: p_.detach().cpu().clone().requires_grad_(True) for n_, p_ in c.named_parameters()} for c in m]
    ws = []
    for q in po:
        w = O.weight_norm_fold(q["weight_v"], q["weight_g"])
        ws.append(w + (w.detach().to(HALF).float() - w.detach()))       # straight-through bf16 rounding
    h = F.conv1d(F.leaky_relu(xo, LRELU_SLOPE), ws[0], po[0]["bias"], padding=get_padding(k, d), dilation=d)
    # leaky-relu with the KERNEL's branch decisions (the sign of its stored intermediate): the folded weights are rounded
    # to bf16 on either side of an fp32 rounding difference, so a pre-activation within 1e-3 of zero may take the other
    # branch -- one such position moves a bias gradient by 0.9 |d| while leaving every forward value where it was
    gate = torch.where(mid_f.float().cpu().transpose(1, 2) > 0, 1.0, LRELU_SLOPE)
    # ... which must be the oracle's own decisions almost everywhere: a wrong sign pattern in the kernel's intermediate
    # would otherwise pass unnoticed (the positions that differ sit within rounding distance of zero)
    agree = ((h.detach() > 0) == (mid_f.float().cpu().transpose(1, 2) > 0))
    assert agree.float().mean().item() >= 0.999, agree.float().mean().item()
    assert float(h.detach()[~agree].abs().max() if (~agree).any() else 0.0) < 2e-2 * float(h.detach().abs().max())
    h = h * gate
    h = h + (h.detach().to(HALF).float() - h.detach())                  # the intermediate is stored as bf16
    yo = xo + F.conv1d(h, ws[1], po[1]["bias"], padding=get_padding(k, 1))
    yo.backward(dy.float().cpu().transpose(1, 2))
    assert rel(y.transpose(1, 2), yo) < 3e-2, rel(y.transpose(1, 2), yo)
    assert rel(xg.grad.transpose(1, 2), xo.grad) < 3e-2, rel(xg.grad.transpose(1, 2), xo.grad)
    for c, q in zip(m, po):
        for n_, p_ in c.named_parameters():
            assert rel(p_.grad, q[n_].grad) < 3e-2, (n_, rel(p_.grad, q[n_].grad))
