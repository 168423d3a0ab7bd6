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
