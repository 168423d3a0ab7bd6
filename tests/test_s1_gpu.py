"""GPU parity of the s1 HIP kernels (through the C ABI) and of the whole s1 micro-step.
fp32: 1e-3 relative to max-abs (north_star); bf16: 3e-2."""
import json
import math
import os

import pytest
import torch
import torch.nn.functional as F
import yaml

from oracle import s1_step as OS
from util_fill import fill_module, s1_batch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


ATTN_CASES = [  # B, H, x_len, y_len, x_lens, y_lens
    (2, 4, 24, 40, [24, 17], [40, 29]),
    (2, 16, 100, 200, [100, 63], [200, 131]),
    (1, 16, 256, 768, [256], [768]),
    (3, 2, 5, 130, [5, 1, 3], [130, 7, 64]),
    (2, 2, 130, 3, [130, 64], [3, 1]),
]


def _hash_keep(seed, bh, L_, thr16):
    """python mirror of drop_row / drop_pair / keep_lo / keep_hi in csrc/attention.hip (mod 2^32): one hash decides the
    keys kp = k & ~16 (low 16 bits) and kp + 16 (high 16 bits) of a query"""
    M, M24 = 0xFFFFFFFF, 0xFFFFFF
    q = torch.arange(L_, dtype=torch.int64)[:, None]
    k = torch.arange(L_, dtype=torch.int64)[None, :]
    a = ((((bh << 11) & M) ^ q) + ((seed * 0x9E3779B1) & M)) & M
    a = a ^ (a >> 16)
    a = (a * 0x85EBCA6B) & M          # int64 wrap-around keeps the low 32 bits exact
    a = a ^ (a >> 13)
    a = (a * 0xC2B2AE35) & M
    a = a ^ (a >> 16)
    kp = k & ~16
    x = (a + (((kp & M24) * 0x85EBCB) & M)) & M
    x = x ^ (x >> 15)
    x = ((((x & M24) * 0x2C1B3D) & M) ^ (x >> 9)) & M
    f = torch.where((k & 16) != 0, x >> 16, x & 0xFFFF)
    return f >= thr16


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", ATTN_CASES)
@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_attention_parity(gpu, case, dtype, dropout):
    from easevoice_trainer_amd.auto_reg.ops import PrefixLMAttentionFn

    B, H, x_len, y_len, xl, yl = case
    D, L_ = 32, x_len + y_len
    E = H * D
    torch.manual_seed(B * 1000 + L_)
    qkv = torch.randn(B, L_, 3 * E) * 0.8
    d_o = torch.randn(B, L_, E)
    if dtype == torch.bfloat16:
        qkv, d_o = qkv.bfloat16().float(), d_o.bfloat16().float()
    x_lens, y_lens = torch.tensor(xl), torch.tensor(yl)
    seed = 12345
    # oracle (materialised mask, plain softmax) with the same dropout mask
    qo = qkv.clone().requires_grad_(True)
    mask = OS.prefix_lm_mask(x_lens, y_lens, x_len, y_len)
    q, k, v = qo.split(E, dim=-1)
    q = q.view(B, L_, H, D).transpose(1, 2); k = k.view(B, L_, H, D).transpose(1, 2); v = v.view(B, L_, H, D).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(D)
    p = F.softmax(s.masked_fill(mask[:, None], float("-inf")), dim=-1)
    if dropout > 0:
        thr16 = int(dropout * 65536.0 + 0.5)
        keep = torch.stack([torch.stack([_hash_keep(seed, b * H + h, L_, thr16) for h in range(H)]) for b in range(B)])
        p = p * keep / (1 - thr16 / 65536.0)
    oo = torch.matmul(p, v).transpose(1, 2).reshape(B, L_, E)
    oo.backward(d_o)
    # HIP
    qg = qkv.to(gpu, dtype).requires_grad_(True)
    og = PrefixLMAttentionFn.apply(qg, x_lens.to(gpu, torch.int32), y_lens.to(gpu, torch.int32), x_len, H, dropout, seed)
    og.backward(d_o.to(gpu, dtype))
    torch.cuda.synchronize()
    tol = 1e-3 if dtype == torch.float32 else 3e-2
    assert rel(og, oo) < tol, "o"
    g_ref, g_got = qo.grad, qg.grad.float().cpu()
    for i, name in enumerate(("dq", "dk", "dv")):
        assert rel(g_got[..., i * E:(i + 1) * E], g_ref[..., i * E:(i + 1) * E]) < tol, name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_add_layernorm_and_ce(gpu, dtype):
    from easevoice_trainer_amd.auto_reg.ops import AddLayerNormFn, CrossEntropySumFn

    torch.manual_seed(3)
    rows, C = 333, 512
    x, r = torch.randn(rows, C), torch.randn(rows, C) * 0.5
    gm, bt = 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
    dy = torch.randn(rows, C)
    if dtype == torch.bfloat16:
        x, r, dy = x.bfloat16().float(), r.bfloat16().float(), dy.bfloat16().float()
    xo, ro, go, bo = [t.clone().requires_grad_(True) for t in (x, r, gm, bt)]
    yo = F.layer_norm(xo + ro, (C,), go, bo, 1e-5)
    yo.backward(dy)
    xg, rg = x.to(gpu, dtype).requires_grad_(True), r.to(gpu, dtype).requires_grad_(True)
    gg, bg = gm.to(gpu).requires_grad_(True), bt.to(gpu).requires_grad_(True)
    yg = AddLayerNormFn.apply(xg, rg, gg, bg, 1e-5)
    yg.backward(dy.to(gpu, dtype))
    tol = 1e-3 if dtype == torch.float32 else 3e-2
    assert rel(yg, yo) < tol and rel(xg.grad, xo.grad) < tol and rel(rg.grad, ro.grad) < tol
    assert rel(gg.grad, go.grad) < tol and rel(bg.grad, bo.grad) < tol
    # cross entropy (sum) + top-3 hits
    V, n = 1025, 777
    logits = torch.randn(n, V) * 2
    if dtype == torch.bfloat16:
        logits = logits.bfloat16().float()
    tg = torch.randint(0, V, (n,))
    tg[::7] = 1024
    lo = logits.clone().requires_grad_(True)
    loss_o = F.cross_entropy(lo, tg, reduction="sum")
    loss_o.backward()
    lg = logits.to(gpu, dtype).requires_grad_(True)
    loss_g, hits = CrossEntropySumFn.apply(lg, tg.to(gpu), 3, 1024)
    loss_g.backward()
    assert abs(float(loss_g) - float(loss_o)) <= (1e-4 if dtype == torch.float32 else 2e-3) * float(loss_o)
    assert rel(lg.grad, lo.grad) < tol
    lt = logits.gather(1, tg[:, None])
    keep = tg != 1024
    want = int((((logits > lt).sum(1) < 3) & keep).sum())
    assert int(hits[0]) == want and int(hits[1]) == int(keep.sum())


def test_scaled_adam_kernels_match_reference_trajectory(gpu):
    from easevoice_trainer_amd.auto_reg.optim import ScaledAdam
    from easevoice_trainer_amd.runtime import ParamArena

    gold = torch.load(os.path.join(HERE, "golden", "s1_small.pt"), weights_only=False)["scaled_adam"]

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for k, v in gold["init"].items():
                setattr(self, k, torch.nn.Parameter(v.clone()))

    h = Holder().to(gpu)
    arena = ParamArena(h, gpu)
    opt = ScaledAdam(arena, lr=0.01, betas=(0.9, 0.95), clipping_scale=2.0, clipping_update_period=4)
    params = dict(h.named_parameters())
    for step, (grads, want) in enumerate(zip(gold["grads"], gold["traj"])):
        arena.zero_grad()
        for k, g in grads.items():
            params[k].grad.copy_(g.to(gpu))
        opt.step()
        opt.param_groups[0]["lr"] = 0.002
        for k in params:
            assert torch.allclose(params[k].detach().cpu(), want[k], rtol=1e-4, atol=3e-6), (step, k)


def test_s1_model_matches_reference_fixture(gpu):
    from easevoice_trainer_amd.train.s1_engine import S1Engine

    gold = torch.load(os.path.join(HERE, "golden", "s1_small.pt"), weights_only=False)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    eng = S1Engine(cfg, gpu, torch.float32)
    fill_module(eng.model, 3)
    eng.model.eval()
    keys = json.load(open(os.path.join(HERE, "golden", "state_dict_keys.json")))["s1"]
    assert {k: list(v.shape) for k, v in eng.model.state_dict().items()} == keys
    c = gold["config"]
    b = s1_batch(c["B"], c["x_len"], c["y_len"])
    loss, acc = eng.model.forward_old(b["phoneme_ids"].to(gpu), torch.tensor(c["x_lens"]).to(gpu),
                                      b["semantic_ids"].to(gpu), torch.tensor(c["y_lens"]).to(gpu),
                                      b["bert_feature"].to(gpu))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - gold["loss"]) <= 1e-3 * gold["loss"]
    assert abs(float(acc) - gold["acc"]) < 1e-6
    params = dict(eng.model.named_parameters())
    for n, s in gold["grad_slices"].items():
        assert rel(params[n].grad.flatten()[:96], s) < 2e-3, n
    tot = {}
    for n, p in params.items():
        top = ".".join(n.split(".")[:3]) if n.startswith("h.layers") else n.split(".")[0]
        tot[top] = tot.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    for k, v in gold["grad_sumsq"].items():
        assert abs(tot[k] - v) <= 5e-3 * v, (k, tot[k], v)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_ce_rows_parity(gpu, dtype):
    """per-row cross-entropy with row-weighted gradient (the DPO branch's launch) vs torch fp32"""
    from easevoice_trainer_amd.auto_reg.ops import CrossEntropyRowsFn

    torch.manual_seed(4)
    V, n = 1025, 515
    logits = torch.randn(n, V) * 2
    if dtype == torch.bfloat16:
        logits = logits.bfloat16().float()
    tg = torch.randint(0, V, (n,))
    tg[::5] = 1024
    w = torch.randn(n)
    lo = logits.clone().requires_grad_(True)
    row_o = F.cross_entropy(lo, tg, reduction="none")
    (row_o * w).sum().backward()
    lg = logits.to(gpu, dtype).requires_grad_(True)
    row_g, hits = CrossEntropyRowsFn.apply(lg, tg.to(gpu), 3, 1024)
    (row_g * w.to(gpu)).sum().backward()
    tol = 1e-3 if dtype == torch.float32 else 3e-2
    assert row_g.dtype == torch.float32 and rel(row_g, row_o) < (1e-5 if dtype == torch.float32 else 2e-3)
    assert rel(lg.grad, lo.grad) < tol
    keep = tg != 1024
    lt = logits.gather(1, tg[:, None])
    assert int(hits[0]) == int((((logits > lt).sum(1) < 3) & keep).sum()) and int(hits[1]) == int(keep.sum())


def test_s1_dpo_matches_reference_fixture(gpu):
    """Text2SemanticDecoder.forward (DPO) on the GPU vs the reference's own module: same rejected sequences for the same
    torch seed, loss / accuracy / gradients within fp32 kernel tolerance (dropout off, as in the fixture)"""
    from easevoice_trainer_amd.train.s1_engine import S1Engine

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    eng = S1Engine(cfg, gpu, torch.float32)
    fill_module(eng.model, 3)
    eng.model.eval()
    for gold in torch.load(os.path.join(HERE, "golden", "s1_dpo.pt"), weights_only=False)["cases"]:
        c = gold["config"]
        b = s1_batch(c["B"], c["x_len"], c["y_len"])
        eng.model.zero_grad(set_to_none=True)
        eng.arena.zero_grad()        # inside an engine the GEMM weight gradients accumulate straight into the flat arena
        torch.manual_seed(c["seed"])
        loss, acc = eng.model.forward(b["phoneme_ids"].to(gpu), torch.tensor(c["x_lens"]).to(gpu),
                                      b["semantic_ids"].to(gpu), torch.tensor(c["y_lens"]).to(gpu),
                                      b["bert_feature"].to(gpu))
        loss.backward()
        torch.cuda.synchronize()
        assert abs(float(loss) - gold["loss"]) <= 1e-3 * gold["loss"], c["seed"]
        assert abs(float(acc) - gold["acc"]) < 1e-6
        params = dict(eng.model.named_parameters())

        def grad_of(p_):             # autograd's tensor, or the parameter's slot in the gradient arena (hip/linear.py)
            return p_.grad if p_.grad is not None else p_._evt_grad_view

        for n, s in gold["grad_slices"].items():
            assert rel(grad_of(params[n]).flatten()[:96], s) < 3e-3, (c["seed"], n)
        tot = {}
        for n, p in params.items():
            top = ".".join(n.split(".")[:3]) if n.startswith("h.layers") else n.split(".")[0]
            tot[top] = tot.get(top, 0.0) + float(grad_of(p).double().pow(2).sum())
        for k, v in gold["grad_sumsq"].items():
            assert abs(tot[k] - v) <= 5e-3 * v, (c["seed"], k, tot[k], v)


def test_s1_engine_dpo_micro_steps(gpu):
    """bf16 engine with train.if_dpo: five micro-steps (one optimiser step), finite loss above the plain one"""
    from easevoice_trainer_amd.train.s1_engine import S1Engine

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    cfg["train"]["if_dpo"] = True
    torch.manual_seed(0)
    eng = S1Engine(cfg, gpu, torch.bfloat16)
    b = {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in s1_batch(4, 32, 96).items()}
    b["phoneme_ids_len"] = torch.tensor([32, 20, 32, 11], device=gpu)
    b["semantic_ids_len"] = torch.tensor([96, 70, 50, 96], device=gpu)
    before = eng.arena.param.clone()
    stepped = []
    for i in range(5):
        loss, acc, st = eng.micro_step(b, i)
        stepped.append(st)
        assert torch.isfinite(loss) and float(loss) > 0
    assert stepped == [False, False, False, False, True]
    assert not torch.equal(before, eng.arena.param)
