"""WeightBank bookkeeping on the GPU (hip/conv.py): gradient tables that follow replaced .grad tensors, the deferred
weight-gradient queue cleared by zero_dw(), extra slabs that start as zeros."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bank(gpu, cin=64, cout=64, k=3):
    from easevoice_trainer_amd.hip import conv as HC

    torch.manual_seed(5)
    m = torch.nn.ModuleList([HC.EvtConv1d(cin, cout, k, padding=k // 2, weight_norm=True)]).to(gpu)
    bank = HC.WeightBank(m, torch.bfloat16, gpu)
    bank.build_tables()
    bank.fold()
    return HC, m, bank


def test_bias_gradient_follows_a_replaced_grad_tensor(gpu):
    """the device tables hold bias.grad's address (fused bias gradients are folded into it by evt_wn_grad_multi); after
    zero_grad(set_to_none=True) / a caller assigning a new .grad, grads() must rebuild them instead of writing into the
    old storage"""
    HC, m, bank = _bank(gpu)
    conv = m[0]
    x = torch.randn(4, 640, 64, device=gpu).bfloat16().requires_grad_(True)
    dy = torch.randn(4, 640, 64, device=gpu).bfloat16()

    def run():
        bank.zero_dw()
        y = conv(x)
        y.backward(dy)
        bank.grads()
        torch.cuda.synchronize()

    run()
    ref = conv.bias.grad.clone()
    assert ref.abs().max() > 0
    old = conv.bias.grad
    for p_ in conv.parameters():
        p_.grad = None                                  # what zero_grad(set_to_none=True) does
    run()
    assert conv.bias.grad is not None and conv.bias.grad.data_ptr() != old.data_ptr()
    assert torch.allclose(conv.bias.grad, ref, rtol=1e-5, atol=1e-5), (conv.bias.grad - ref).abs().max()
    assert torch.allclose(old, ref), "the old tensor must not have been written again"


def test_zero_dw_drops_a_stale_deferred_queue(gpu):
    """a backward that never reached grads() leaves queued weight-gradient launches behind; the next step's zero_dw()
    must not let them land in the fresh gradient images"""
    HC, m, bank = _bank(gpu)
    conv = m[0]
    bank.defer_n = 64
    x = torch.randn(2, 320, 64, device=gpu).bfloat16().requires_grad_(True)
    conv(x).backward(torch.randn(2, 320, 64, device=gpu).bfloat16())
    assert bank._deferred, "the launch must have been queued"
    bank.zero_dw()
    assert not bank._deferred and not bank._held and bank._deferred_bytes == 0
    bank.grads()
    torch.cuda.synchronize()
    assert float(conv.weight_v.grad.abs().max()) == 0.0


def test_extra_slabs_start_as_zeros(gpu):
    HC, m, bank = _bank(gpu)
    assert bank.dw_extra_arena is not None and float(bank.dw_extra_arena.abs().max()) == 0.0
    assert float(bank.db_part_arena.abs().max()) == 0.0


FOLD_CASES = [  # cin, cout, k, stride, groups, transposed, weight_norm
    (64, 64, 3, 1, 1, False, True), (1024, 1024, 5, 1, 1, False, True), (128, 512, 5, 3, 1, False, True),
    (192, 384, 5, 1, 1, False, True), (256, 128, 16, 8, 1, True, True), (16, 16, 11, 1, 1, False, True),
    (64, 256, 41, 4, 16, False, True), (192, 1, 7, 1, 1, False, False), (1, 16, 15, 1, 1, False, True),
    (96, 192, 1, 1, 1, False, False), (100, 72, 3, 1, 1, False, True),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_grouped_fold_writes_the_same_images(gpu, dtype, monkeypatch):
    """evt_wn_fold_groups (eight rows per block, 16-byte pieces) against evt_wn_fold_multi (a row per block) on one bank of
    layers of every geometry the models use -- stride 1 / polyphase ALT, ConvTranspose, grouped, 1-row and 1-column layers,
    widths that are not multiples of 8 / 32, with and without weight norm: REG and ALT images BIT-identical, padding
    included (the grouped kernel's norm is accumulated in the row kernel's order)"""
    from easevoice_trainer_amd.hip import conv as HC

    torch.manual_seed(11)
    mods = []
    for cin, cout, k, stride, groups, transposed, wn in FOLD_CASES:
        pad = (k - stride) // 2 if transposed else k // 2
        mods.append(HC.EvtConv1d(cin, cout, k, stride=stride, padding=pad, groups=groups, transposed=transposed,
                                 weight_norm=wn))
    m = torch.nn.ModuleList(mods).to(gpu)
    bank = HC.WeightBank(m, dtype, gpu)
    bank.build_tables()
    monkeypatch.setenv("EVT_FOLD_GROUPS", "0")
    bank.reg_arena.fill_(7.0); bank.alt_arena.fill_(7.0)         # padding the kernels do not write stays 7 in both runs
    bank.fold()
    torch.cuda.synchronize()
    reg0, alt0 = bank.reg_arena.clone(), bank.alt_arena.clone()
    monkeypatch.setenv("EVT_FOLD_GROUPS", "1")
    bank.reg_arena.fill_(7.0); bank.alt_arena.fill_(7.0)
    bank.fold()
    torch.cuda.synchronize()
    # the grouped kernel also writes the zero padding of the segments it stages (taps beyond k, columns beyond the
    # parameter's): where the row kernel left the fill value, the image is never read -- compare what the row kernel wrote
    wrote = reg0 != 7.0
    assert torch.equal(bank.reg_arena[wrote], reg0[wrote])
    pad = bank.reg_arena[~wrote]
    assert bool(((pad == 0) | (pad == 7.0)).all())
    assert torch.equal(bank.alt_arena, alt0)
    # and a slot-aligned sub-range takes the grouped path too
    lo, hi = bank.rows_of([m[1], m[2]])
    bank.reg_arena.fill_(7.0); bank.alt_arena.fill_(7.0)
    bank.fold(lo, hi)
    torch.cuda.synchronize()
    s1, s2 = bank.slots[1], bank.slots[2]
    for s in (s1, s2):
        ro = s.reg.data_ptr() - bank.reg_arena.data_ptr()
        ro //= bank.reg_arena.element_size()
        w = wrote[ro: ro + s.layout.reg_elems]
        assert torch.equal(s.reg[w], reg0[ro: ro + s.layout.reg_elems][w])
    assert float(bank.slots[0].reg.float().min()) == 7.0          # other layers untouched
