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


def _bank(N, K, gpu, g, bias=True):
    from easevoice_trainer_amd.hip.linear import LinearBank

    w = torch.nn.Parameter((torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().float().to(gpu))
    b = torch.nn.Parameter((torch.randn(N, generator=g) * 0.1).to(gpu)) if bias else None
    bank = LinearBank([("t", w, b)], torch.bfloat16, gpu)
    bank.prepare()
    return w, b, w._evt_slot


# the last three walk SEVERAL tiles per persistent block (768 / 314 / 423 tiles on 256 CUs): the pipelined kernel's tile
# boundary (next tile's pieces in flight under the quadrant epilogues, bias swap, the last quadrant riding in the next tile's
# first phase), with 8, 2 and 3 K tiles per output tile and a ragged last token tile
@pytest.mark.parametrize("shape", [(4096, 2048, 512), (2304, 512, 2048), (3000, 1536, 512), (32768, 1536, 512),
                                   (40000, 512, 128), (36000, 768, 192)], ids=lambda s: "x".join(map(str, s)))
def test_gemm256_fused_epilogues(gpu, shape):
    """the 256 x 256 kernel's epilogues (csrc/gemm256.hip): relu + dropout in the forward store, gate + add in the
    backward-data store, against fp32 arithmetic on the CPU; the dropout mask is the one evt_relu_dropout_fwd draws for
    the same (seed, site)"""
    import ctypes as C
    from easevoice_trainer_amd.hip import enc as E, lib as L
    from easevoice_trainer_amd.hip.linear import gemm_bwd_data, gemm_fwd

    M, N, K = shape
    g = torch.Generator().manual_seed(N + K)
    w, b, slot = _bank(N, K, gpu, g)
    assert slot.fused(M, False)                      # (K = 128 / 192 outputs are not 256-wide: backward-data runs on other kernels)
    x = torch.randn(M, K, generator=g).bfloat16().to(gpu)
    E.seed_rng(gpu, 123)
    p, site = 0.25, 9
    z = (x.float().cpu() @ w.detach().cpu().t() + b.detach().cpu())                       # fp32 reference
    y0 = gemm_fwd(slot, x, relu=True)                                                       # relu only
    assert rel(y0, torch.relu(z)) < 2e-2
    # every element, not only the largest: a tile written to the wrong rows / a stale bias would hide under max-norm
    err = (y0.float().cpu() - torch.relu(z)).abs()
    assert float(err.max()) < 0.05 * float(torch.relu(z).abs().max()) and float(err.mean()) < 2e-3 * float(z.abs().mean() + 1)
    yp = gemm_fwd(slot, x)                                                                  # bias only, no activation
    assert rel(yp, z) < 2e-2
    y = gemm_fwd(slot, x, relu=True, drop=(p, site))
    # the standalone kernel on the same positions draws the same mask
    zz = torch.relu(z).bfloat16().to(gpu)
    yk = torch.empty_like(zz)
    L.check(L.lib().evt_relu_dropout_fwd(L.DT_BF16, L.ptr(zz), C.c_float(p), L.ptr(E.rng_counter(gpu)), C.c_uint32(site), None,
                                         0, 0, L.ptr(yk), C.c_int64(zz.numel()), L.stream_ptr()), "evt_relu_dropout_fwd")
    live = zz.float() > 1e-2
    assert torch.equal((y.float() > 0) & live, (yk.float() > 0) & live)
    assert rel(y, yk) < 2e-2
    kept = ((y.float() > 0) & live).float().sum() / live.float().sum()
    assert abs(kept.item() - (1 - p)) < 0.01
    # forward add-epilogue
    addt = torch.randn(M, N, generator=g).bfloat16().to(gpu)
    ya = gemm_fwd(slot, x, relu=False, add=addt)
    assert rel(ya, z + addt.float().cpu()) < 2e-2
    # backward-data: gate (derivative of relu + dropout read off the saved activation) and add (residual gradient)
    dy = torch.randn(M, N, generator=g).bfloat16().to(gpu)
    gate = (torch.randn(M, K, generator=g) * (torch.rand(M, K, generator=g) > 0.4)).clamp(min=0).bfloat16().to(gpu)
    addk = torch.randn(M, K, generator=g).bfloat16().to(gpu)
    ref = dy.float().cpu() @ w.detach().cpu()
    assert rel(gemm_bwd_data(slot, dy), ref) < 2e-2
    got = gemm_bwd_data(slot, dy, gate=gate, gate_pos=1.0 / (1 - p), add=addk)
    want = ref * (gate.float().cpu() > 0).float() / (1 - p) + addk.float().cpu()
    assert rel(got, want) < 2e-2


def _run_layer_blocks(gpu, dtype, B, seed=5):
    """one post-LN layer through auto_reg/blocks.py; returns (layer, x, lens, out, dx, {param: grad})"""
    from easevoice_trainer_amd.auto_reg.blocks import attn_block, ffn_block
    from easevoice_trainer_amd.auto_reg.t2s_model import TransformerEncoderLayer
    from easevoice_trainer_amd.hip.linear import LinearBank

    Lq, E, x_len = 1024, 512, 256
    torch.manual_seed(seed)
    layer = TransformerEncoderLayer(E, 16, 2048, 0.0).to(gpu)
    with torch.no_grad():
        for p_ in layer.parameters():
            p_.copy_((p_ + 0.02 * torch.randn_like(p_)).to(dtype).float())
    specs = [("in", layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias),
             ("out", layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias),
             ("l1", layer.linear1.weight, layer.linear1.bias), ("l2", layer.linear2.weight, layer.linear2.bias)]
    LinearBank(specs, dtype, gpu).prepare()
    layer.eval()
    x = torch.randn(B, Lq, E, device=gpu).to(dtype)
    x_lens = torch.full((B,), x_len, dtype=torch.int32, device=gpu)
    y_lens = torch.tensor([Lq - x_len, 500, 17][:B], dtype=torch.int32, device=gpu)
    wgt = torch.randn(B, Lq, E, device=gpu)
    xg = x.clone().requires_grad_(True)
    h = attn_block(xg, layer.self_attn, layer.norm1, x_lens, y_lens, x_len, 1, 0.0, 1)
    out = ffn_block(h, layer.linear1, layer.linear2, layer.norm2, 0.0, 2, 3)
    (out.float() * wgt).sum().backward()
    torch.cuda.synchronize()
    return layer, x, (x_lens, y_lens, x_len, Lq, E), wgt, out.detach(), xg.grad, {k: p_.grad.clone() for k, p_ in layer.named_parameters()}


def test_s1_blocks_fused_equals_composed(gpu, monkeypatch):
    """the fused epilogues (gate / add in the backward GEMMs' stores, 256 x 256 kernel) against the SAME blocks on the
    128 x 128 kernels + separate element-wise launches (EVT_NO_GEMM256): same bf16 operands, same roundings up to the
    staging of the gated value, so everything agrees closely -- this is the check of the fusion logic itself"""
    _l, _x, _lens, _w, out_f, dx_f, g_f = _run_layer_blocks(gpu, torch.bfloat16, 3)
    monkeypatch.setenv("EVT_NO_GEMM256", "1")
    _l, _x, _lens, _w, out_c, dx_c, g_c = _run_layer_blocks(gpu, torch.bfloat16, 3)
    monkeypatch.delenv("EVT_NO_GEMM256")
    assert rel(out_f, out_c) < 1e-2
    assert rel(dx_f, dx_c) < 2e-2
    for k in g_f:
        assert rel(g_f[k], g_c[k]) < 2e-2, k


@pytest.mark.parametrize("dtype,B", [(torch.bfloat16, 3), (torch.float32, 1)], ids=["bf16-gemm256", "f32-fallback"])
def test_s1_blocks_match_composition(gpu, dtype, B):
    """auto_reg/blocks.py (attention block / FFN block as one autograd node each, fused epilogues) against the same
    arithmetic composed from torch ops in fp32 on the CPU (transformer.py:311-334 with dropout off), with every
    parameter gradient; bf16 at 3072 rows runs the 256 x 256 kernel, fp32 the fallback composition.  The bf16 bounds on
    the FFN weights are loose: a pre-activation within bf16 rounding of zero takes the other branch of relu' than in the
    fp32 reference, a full-size (not a rounding-size) difference in a few of the 3072 summed rows."""
    from oracle import s1_step as OS

    layer, x, (x_lens, y_lens, x_len, Lq, E), wgt, out, dx, grads = _run_layer_blocks(gpu, dtype, B)
    F_ = torch.nn.functional
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.named_parameters()}
    xr = x.float().cpu().requires_grad_(True)
    qkv = F_.linear(xr, P["self_attn.in_proj_weight"], P["self_attn.in_proj_bias"])
    mask = OS.prefix_lm_mask(x_lens.cpu().long(), y_lens.cpu().long(), x_len, Lq - x_len)
    sa = F_.linear(OS.attention(qkv, mask, 16), P["self_attn.out_proj.weight"], P["self_attn.out_proj.bias"])
    h1 = F_.layer_norm(xr + sa, (E,), P["norm1.weight"], P["norm1.bias"], layer.norm1.eps)
    ff = F_.linear(torch.relu(F_.linear(h1, P["linear1.weight"], P["linear1.bias"])), P["linear2.weight"], P["linear2.bias"])
    ref = F_.layer_norm(h1 + ff, (E,), P["norm2.weight"], P["norm2.bias"], layer.norm2.eps)
    (ref * wgt.cpu()).sum().backward()
    bf = dtype == torch.bfloat16
    tol = 3e-2 if bf else 1e-3
    assert rel(out, ref) < tol
    assert rel(dx, xr.grad) < (6e-2 if bf else tol)            # through 8 bf16 GEMMs, the attention and two LayerNorms
    for k, g_ in grads.items():
        assert rel(g_, P[k].grad) < ((1e-1 if "linear" in k else 5e-2) if bf else tol), k


def test_s1_blocks_skip_frozen_weights(gpu):
    """a weight that does not require a gradient gets no gradient launch (ctx.needs_input_grad), everything else is
    unchanged bit for bit"""
    from easevoice_trainer_amd.auto_reg.blocks import attn_block, ffn_block

    layer, x, (x_lens, y_lens, x_len, Lq, E), wgt, out, dx, grads = _run_layer_blocks(gpu, torch.bfloat16, 2)
    for p_ in layer.parameters():
        p_.grad = None
    layer.linear1.weight.requires_grad_(False)
    layer.self_attn.out_proj.bias.requires_grad_(False)
    xg = x.clone().requires_grad_(True)
    h = attn_block(xg, layer.self_attn, layer.norm1, x_lens, y_lens, x_len, 1, 0.0, 1)
    out2 = ffn_block(h, layer.linear1, layer.linear2, layer.norm2, 0.0, 2, 3)
    (out2.float() * wgt).sum().backward()
    assert layer.linear1.weight.grad is None and layer.self_attn.out_proj.bias.grad is None
    assert torch.equal(out2, out) and torch.equal(xg.grad, dx)
    for k, p_ in layer.named_parameters():
        if k in ("linear1.weight", "self_attn.out_proj.bias"):
            continue
        # fp32 atomics of the split reductions: the order of the adds differs between runs
        assert rel(p_.grad, grads[k]) < 1e-4, k
