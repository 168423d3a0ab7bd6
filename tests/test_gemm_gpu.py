"""GPU parity of the dense-layer GEMMs (evt_gemm_bf16_fwd / _bwd_data / _bwd_weight through hip/linear.py) against
torch's fp32 F.linear on the CPU (the arithmetic of the reference's call sites, transformer.py:207-224,330-334,
patched_mha_with_cache.py:242,460, t2s_model.py:276,486), at the s1 layer shapes with ragged row counts, the padded
1025-entry vocabulary projection, and the relu epilogue.  fp32: 1e-3 relative; bf16: inputs rounded to bf16 on both
sides, 2e-2 of the tensor's max (fp32 accumulation over K <= 2048)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # (M, N, K, bias, relu)
    (1000, 1536, 512, True, False),      # packed in-projection, ragged M
    (4096, 512, 2048, True, False),      # linear2
    (2056, 2048, 512, True, True),       # linear1 with the relu epilogue
    (640, 512, 1024, True, False),       # bert_proj
    (777, 1025, 512, False, False),      # vocabulary projection: padded to 1152 columns
    (96, 512, 512, True, False),         # tiny M (fewer rows than a tile)
    (5000, 1536, 512, True, False),      # long reduction, ragged M: the 128 x 128 weight-gradient tile (wgrad_gemm)
    (3001, 1152, 512, False, False),     # wgrad_gemm with the padded vocabulary width
]


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}x{c[1]}x{c[2]}" for c in CASES])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_gemm_matches_torch_linear(gpu, case, dtype):
    from easevoice_trainer_amd.hip.linear import LinearBank, linear

    M, N, K, has_bias, relu = case
    g = torch.Generator().manual_seed(M + N)
    Np = (N + 127) // 128 * 128 if N % 8 else N
    store = torch.zeros(Np, K, device=gpu)                     # the padded rows the image reads must exist and be zero
    w = torch.nn.Parameter(store[:N])
    w.data.copy_(torch.randn(N, K, generator=g) * K ** -0.5)
    b = torch.nn.Parameter((torch.randn(N, generator=g) * 0.1).to(gpu)) if has_bias else None
    x = torch.randn(1, M, K, generator=g)                      # a 3-D input, like [B, L, K]
    dy = torch.randn(1, M, N, generator=g)
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    bank = LinearBank([("t", w, b)], dtype, gpu)
    bank.prepare()
    xg = x.to(gpu, dtype).requires_grad_(True)
    y = linear(xg, w, b, relu=relu)
    assert y.shape == (1, M, Np)
    if Np != N:
        assert not y[..., N:].any()                            # padding columns are exact zeros
    y[..., :N].backward(dy.to(gpu, dtype))
    # reference: fp32 on the CPU with the weights as the kernel sees them (bf16-rounded in the bf16 run)
    wr = w.detach().cpu()
    wr = wr.bfloat16().float() if dtype == torch.bfloat16 else wr
    xr = x.clone().requires_grad_(True)
    wr = wr.clone().requires_grad_(True)
    br = b.detach().cpu().clone().requires_grad_(True) if has_bias else None
    yr = F.linear(xr, wr, br)
    yr = F.relu(yr) if relu else yr
    yr.backward(dy)
    tol = 1e-3 if dtype == torch.float32 else 2e-2
    assert rel(y[..., :N], yr) < tol
    assert rel(xg.grad, xr.grad) < tol
    assert rel(w.grad, wr.grad) < tol
    if has_bias:
        assert rel(b.grad, br.grad) < tol


def test_bank_refolds_after_weight_change(gpu):
    from easevoice_trainer_amd.hip.linear import LinearBank, linear

    w = torch.nn.Parameter(torch.randn(256, 128, device=gpu) * 0.1)
    bank = LinearBank([("t", w, None)], torch.bfloat16, gpu)
    bank.prepare()
    x = torch.randn(64, 128, device=gpu).bfloat16()
    y0 = linear(x, w).float()
    with torch.no_grad():
        w.mul_(2.0)                     # versioned in-place write (load_state_dict, copy_)
    bank.prepare()
    assert rel(linear(x, w), 2 * y0) < 1e-2
    w.data.view(-1)[0] += 0.0           # raw write the bank cannot see ...
    bank.mark_dirty()                   # ... is announced, as S1Engine does after the optimiser launch
    bank.prepare()
    assert rel(linear(x, w), 2 * y0) < 1e-2
