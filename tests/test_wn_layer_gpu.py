"""GPU: the one-launch WN layer forward (csrc/wn_layer.hip, evt_wn_layer_fwd) against the four launches it replaces
(evt_conv1d_fwd, evt_gated_act_fwd, evt_conv1d_fwd, evt_wn_residual_fwd) on the same weights (the fused launch reads
evt_frag_pack's re-ordered copy of the REG images; a refold must refresh it): every output incl. the
two tensors saved for the backward, both 16-bit builds, ragged lengths, tiles that end inside / beyond a sequence, first
(no skip sum yet) / middle / last layer, with and without conditioning.  The rounding points are the same, so the paths differ
by the order of the fp32 sums inside a convolution only: a few values may land on the neighbouring 16-bit number.
The stack-level parity against the CPU oracle is tests/test_enc_ops_gpu.py::test_wn_stack_node (it runs the fused forward)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

H = 192


def _stack(gpu, dtype, n_layers, gin):
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.module.models import WN

    m = WN(H, 5, 1, n_layers, gin_channels=gin).to(gpu)
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.dim() == 1:
                p_.normal_(0, 0.05)
    bank = HC.WeightBank(m, dtype, gpu)
    bank.build_tables()
    bank.fold()
    return m, bank


def _near(a, b, ulp, name, few=0.05, ulps=2.5):
    """equal up to the neighbouring 16-bit value on a few elements"""
    a, b = a.float(), b.float()
    scale = b.abs().max().item() + 1e-6
    d = (a - b).abs()
    assert d.max().item() <= ulps * ulp * scale, (name, d.max().item() / scale)
    assert (d > 0).float().mean().item() < few, (name, (d > 0).float().mean().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("B,T", [(16, 200), (3, 77), (2, 16), (2, 5), (1, 33)])
def test_layer_forward_equals_the_four_launches(gpu, dtype, B, T):
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.hip import lib as L
    from easevoice_trainer_amd.hip import wn as W

    torch.manual_seed(5)
    L.set_half(dtype)
    m, bank = _stack(gpu, dtype, 3, 512)
    frag_in, frag_rs = W._frag_images(tuple(c._slot for c in m.in_layers), tuple(c._slot for c in m.res_skip_layers))
    dt = L.dt_code(dtype)
    assert L.lib().evt_wn_layer_supported(dt, H, 5, 1) == 1
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    lens = torch.randint(1, T + 1, (B,), dtype=torch.int32)
    lens[0] = T
    lens = lens.to(gpu)
    live = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).unsqueeze(-1)
    for layer, with_g, with_acc in [(0, True, False), (1, True, True), (1, False, True), (2, True, True), (2, False, False)]:
        last = layer == 2
        si, sr = m.in_layers[layer]._slot, m.res_skip_layers[layer]._slot
        x = (torch.randn(B, T, H, device=gpu) * live).to(dtype)
        g = (torch.randn(B, 2 * H, device=gpu) * 0.5).to(dtype) if with_g else None
        acc = torch.randn(B, T, H, device=gpu).to(dtype) if with_acc else None
        # the four launches
        x_in_u = HC._fwd(si, x, None, 1.0, L.ACT_NONE, 1.0)
        acts_u = torch.empty(B, T, H, dtype=dtype, device=gpu)
        L.check(L.lib().evt_gated_act_fwd(dt, L.ptr(x_in_u), L.ptr(g), L.ptr(acts_u), B, T, H, L.stream_ptr()), "gate")
        rs = HC._fwd(sr, acts_u, None, 1.0, L.ACT_NONE, 1.0)
        acc_u = torch.empty(B, T, H, dtype=dtype, device=gpu)
        xo_u = None if last else torch.empty_like(acc_u)
        L.check(L.lib().evt_wn_residual_fwd(dt, L.ptr(None if last else x), L.ptr(rs), L.ptr(acc), L.ptr(lens), T, L.ptr(xo_u),
                                            L.ptr(acc_u), C.c_int64(B * T), H, int(last), L.stream_ptr()), "residual")
        # one launch
        x_in_f = torch.full((B, T, 2 * H), float("nan"), dtype=dtype, device=gpu)
        acts_f = torch.full((B, T, H), float("nan"), dtype=dtype, device=gpu)
        acc_f = torch.full((B, T, H), float("nan"), dtype=dtype, device=gpu)
        xo_f = None if last else torch.full((B, T, H), float("nan"), dtype=dtype, device=gpu)
        mi, mr = si.module, sr.module
        L.check(L.lib().evt_wn_layer_fwd(dt, L.ptr(x), L.ptr(frag_in[layer]), L.ptr(mi.bias.data), L.ptr(frag_rs[layer]),
                                         L.ptr(mr.bias.data),
                                         L.ptr(g), L.ptr(acc), L.ptr(lens), L.ptr(x_in_f), L.ptr(acts_f), L.ptr(xo_f),
                                         L.ptr(acc_f), B, T, H, 5, int(last), L.stream_ptr()), "evt_wn_layer_fwd")
        torch.cuda.synchronize()
        tag = f"layer {layer} g={with_g} acc={with_acc}"
        for a in (x_in_f, acts_f, acc_f) + (() if last else (xo_f,)):
            assert torch.isfinite(a.float()).all(), tag          # every element written
        _near(x_in_f, x_in_u, ulp, tag + " x_in")
        _near(acts_f, acts_u, ulp, tag + " acts")
        _near(acc_f, acc_u, ulp, tag + " skip sum")
        if not last:
            _near(xo_f, xo_u, ulp, tag + " x")
            assert (xo_f.float() * (~live)).abs().max().item() == 0.0      # masked rows are exact zeros
        else:
            assert (acc_f.float() * (~live)).abs().max().item() == 0.0
    L.set_half(torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("B,T", [(16, 200), (3, 77), (2, 16), (2, 5), (1, 33)])
def test_layer_backward_data_equals_the_four_launches(gpu, dtype, B, T):
    """evt_wn_layer_bwd_data against evt_wn_residual_bwd + the 1 x 1 backward-data launch + evt_gated_act_bwd + the k = 5
    backward-data launch with its add epilogue: drs (bit-equal: a masked copy), dx_in, dx, the conditioning gradient"""
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.hip import lib as L

    torch.manual_seed(6)
    L.set_half(dtype)
    m, bank = _stack(gpu, dtype, 3, 512)
    dt = L.dt_code(dtype)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    lens = torch.randint(1, T + 1, (B,), dtype=torch.int32)
    lens[0] = T
    lens = lens.to(gpu)
    for layer, with_g, with_dx in [(2, True, False), (1, True, True), (1, False, True), (0, True, True), (0, False, False)]:
        last = layer == 2
        si, sr = m.in_layers[layer]._slot, m.res_skip_layers[layer]._slot
        x_in = torch.randn(B, T, 2 * H, device=gpu).to(dtype)
        g = (torch.randn(B, 2 * H, device=gpu) * 0.5).to(dtype) if with_g else None
        dacc = torch.randn(B, T, H, device=gpu).to(dtype)
        dx_next = torch.randn(B, T, H, device=gpu).to(dtype) if (with_dx and not last) else None
        rs_w = H if last else 2 * H
        # the four launches
        drs_u = torch.empty(B, T, rs_w, dtype=dtype, device=gpu)
        dx_res = None if last else torch.empty(B, T, H, dtype=dtype, device=gpu)
        L.check(L.lib().evt_wn_residual_bwd(dt, L.ptr(dx_next), L.ptr(dacc), L.ptr(lens), T, L.ptr(dx_res), L.ptr(drs_u),
                                            C.c_int64(B * T), H, int(last), L.stream_ptr()), "residual bwd")
        dacts = HC._bwd_data(sr, drs_u, None, None, None, B, T, 1.0, L.ACT_NONE, 1.0)
        dx_in_u = torch.empty_like(x_in)
        dg_u = torch.zeros(B, 2 * H, dtype=torch.float32, device=gpu) if with_g else None
        L.check(L.lib().evt_gated_act_bwd(dt, L.ptr(x_in), L.ptr(g), L.ptr(dacts), L.ptr(dx_in_u), L.ptr(dg_u), B, T, H,
                                          L.stream_ptr()), "gate bwd")
        dx_u = HC._bwd_data(si, dx_in_u, None, None, dx_res, B, T, 1.0, L.ACT_NONE, 1.0)
        # one launch
        nan = float("nan")
        drs_f = torch.full((B, T, rs_w), nan, dtype=dtype, device=gpu)
        dx_in_f = torch.full((B, T, 2 * H), nan, dtype=dtype, device=gpu)
        dx_f = torch.full((B, T, H), nan, dtype=dtype, device=gpu)
        dg_f = torch.zeros(B, 2 * H, dtype=torch.float32, device=gpu) if with_g else None
        L.check(L.lib().evt_wn_layer_bwd_data(dt, L.ptr(dx_next), L.ptr(dacc), L.ptr(x_in), L.ptr(g), L.ptr(bank.frag(sr, "alt")),
                                              L.ptr(bank.frag(si, "alt")), L.ptr(lens), L.ptr(drs_f), L.ptr(dx_in_f),
                                              L.ptr(dx_f), L.ptr(dg_f), B, T, H, 5, int(last), L.stream_ptr()),
                "evt_wn_layer_bwd_data")
        torch.cuda.synchronize()
        tag = f"layer {layer} g={with_g} dx_next={dx_next is not None}"
        for a in (drs_f, dx_in_f, dx_f):
            assert torch.isfinite(a.float()).all(), tag
        assert torch.equal(drs_f, drs_u), tag
        _near(dx_in_f, dx_in_u, ulp, tag + " dx_in", few=0.08)
        _near(dx_f, dx_u, ulp, tag + " dx", few=0.6, ulps=4.0)      # sums over 1920 products of dx_in values that may have flipped
        if with_g:
            err = (dg_f - dg_u).abs().max().item() / (dg_u.abs().max().item() + 1e-6)
            assert err < 2e-2, (tag, err)
    L.set_half(torch.bfloat16)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_stack_fused_forward_vs_four_launch_forward(gpu, dtype):
    """the whole autograd node both ways (one launch per layer forward + one for the data half of its backward, against four +
    four): output, input / conditioning gradients and every parameter gradient"""
    from easevoice_trainer_amd.hip import lib as L
    from easevoice_trainer_amd.hip import wn as W

    L.set_half(dtype)
    torch.manual_seed(9)
    B, T, NL, GIN = 4, 120, 4, 512
    m, bank = _stack(gpu, dtype, NL, GIN)
    lens = torch.tensor([T, 64, 33, 7], device=gpu, dtype=torch.int32)
    live = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).float().unsqueeze(-1)
    x0 = (torch.randn(B, T, H, device=gpu) * live).to(dtype)
    g0 = torch.randn(B, GIN, device=gpu)
    wgt = torch.randn(B, T, H, device=gpu)
    res = {}
    try:
        for fused in (True, False):
            W.FUSED_FORWARD = W.FUSED_BACKWARD = fused
            for p_ in m.parameters():
                p_.grad = None
            bank.zero_dw()
            x, g = x0.clone().requires_grad_(True), g0.clone().requires_grad_(True)
            out = m(x, live.to(dtype), g=g, lens=lens)
            (out.float() * wgt).sum().backward()
            bank.grads()
            torch.cuda.synchronize()
            res[fused] = dict(out=out.detach().float(), dx=x.grad.float() * live, dg=g.grad.float(),
                              **{k: p_.grad.float().clone() for k, p_ in m.named_parameters()})
    finally:
        W.FUSED_FORWARD = W.FUSED_BACKWARD = True
        L.set_half(torch.bfloat16)
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3     # 16-bit roundings that flipped, carried through four layers
    for k, a in res[True].items():
        b = res[False][k]
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-6)
        assert err < tol, (k, err)


def test_fragment_images_follow_the_fold(gpu):
    """the fragment-order copies are re-made behind every fold: after an in-place weight change + fold, the fused forward
    sees the new weights (and equals the four launches again)"""
    from easevoice_trainer_amd.hip import lib as L
    from easevoice_trainer_amd.hip import wn as W

    dtype = torch.bfloat16
    L.set_half(dtype)
    torch.manual_seed(3)
    B, T = 2, 50
    m, bank = _stack(gpu, dtype, 2, 0)
    lens = torch.tensor([T, 31], device=gpu, dtype=torch.int32)
    live = (torch.arange(T, device=gpu)[None, :] < lens[:, None]).float().unsqueeze(-1)
    x = (torch.randn(B, T, H, device=gpu) * live).to(dtype)
    with torch.no_grad():
        y0 = m(x, live.to(dtype), lens=lens).float()
        for p_ in m.parameters():
            p_.mul_(1.5)
        bank.fold()
        y1 = m(x, live.to(dtype), lens=lens).float()
        W.FUSED_FORWARD = False
        try:
            y1_u = m(x, live.to(dtype), lens=lens).float()
        finally:
            W.FUSED_FORWARD = True
    assert (y1 - y0).abs().max().item() > 1e-2 * y0.abs().max().item()          # the change arrived
    assert (y1 - y1_u).abs().max().item() <= 2e-2 * y1_u.abs().max().item()
    # packed image == gather of the REG image
    s = m.in_layers[0]._slot
    lay = s.layout
    ktot = lay.reg_nchunk * lay.reg_kp * lay.reg_ck
    frag = W._frag_images(tuple(c._slot for c in m.in_layers), tuple(c._slot for c in m.res_skip_layers))[0][0]
    reg = s.reg.view(lay.d0, ktot)
    want = reg.view(lay.d0 // 16, 16, ktot // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)   # [tile][ks][g][n][8]
    torch.cuda.synchronize()
    assert torch.equal(frag, want)
