"""GPU: the IEEE-half build of the library (libevt_hip_f16.so: the same sources compiled with -DEVT_HALF_F16, the compute
type of the reference's fp16_run mode, src/train/sovits.py:459-525) through the SAME parity cases the bfloat16 build is
tested with -- every convolution family (naive, implicit GEMM, ring, deep, narrow, halo weight gradients), the fused
ResBlock kernels forward and backward (narrow and wide), relative / plain attention, the encoder element-wise ops, the
discriminators' generator step, the period fold and the masked KL -- against the same CPU oracle, with half-rounded inputs
and weights.  The case bodies live in the per-family test modules; here they are called with torch.float16 (module-level
HALF switched for the hard-wired ones).  Tolerances: the modules' 16-bit bounds, tightened where the module keeps a
per-dtype table (half carries 11 significant bits against bfloat16's 8)."""
import pytest
import torch

import test_conv_gpu as TC
import test_disc_gen_gpu as TD
import test_enc_ops_gpu as TE
import test_kl_gpu as TK
import test_mha_gpu as TM
import test_mpd_fold_gpu as TF
import test_resunit_bwd_gpu as TRB
import test_resunit_gpu as TR

pytestmark = pytest.mark.gpu
F16 = torch.float16


@pytest.fixture
def f16(monkeypatch):
    from easevoice_trainer_amd.hip import lib as L

    for mod in (TC, TR, TRB):
        monkeypatch.setattr(mod, "HALF", F16)
    L.set_half(F16)
    assert L.lib().evt_half_dtype() == L.DT_F16
    yield
    L.set_half(torch.bfloat16)


@pytest.mark.parametrize("impl", [1, 0], ids=["naive", "auto"])
def test_conv_families_f16(gpu, f16, impl):
    for ci in range(len(TC.CASES)):
        TC.test_conv_parity(gpu, ci, impl, F16)


def test_conv_tuned_paths_f16(gpu, f16):
    for ci in range(len(TC.NARROW_CASES)):
        TC.test_conv_narrow_parity(gpu, ci)
    for ci in range(len(TC.RING_CASES)):
        TC.test_conv_ring_parity(gpu, ci)
    for ci in range(len(TC.DEEP_CASES)):
        TC.test_conv_deep_parity(gpu, ci)
    for ci in TC._halo_case_ids():
        TC.test_wgrad_halo_parity(gpu, ci)


def test_fused_resblock_kernels_f16(gpu, f16):
    for case in TR.CASES:
        TR.test_fused_resblock_step(gpu, case)
    for case in [(16, 11, 5, 777), (16, 7, 3, 64), (32, 7, 5, 129), (32, 3, 1, 200), (32, 11, 3, 500)]:
        TRB.test_fused_backward_vs_oracle(gpu, case)
    for case in [(16, 11, 5, 777), (32, 3, 5, 1000)]:
        TRB.test_fused_backward_scale_on_load(gpu, case)
    for case in [(16, 333), (32, 200)]:
        TRB.test_grouped_stage_vs_block_by_block(gpu, case)
    for case in [(64, 11, 5, 777), (64, 3, 1, 200), (128, 7, 3, 640), (128, 11, 5, 333)]:
        TRB.test_wide_fused_step_vs_oracle(gpu, case)


def test_attention_f16(gpu, f16):
    for shape in [(2, 37, 2, 96), (3, 200, 2, 96), (2, 130, 4, 64), (1, 70, 2, 32)]:
        TM.test_relattn_parity(gpu, shape, F16)
    for shape, scale in [((2, 200, 60, 4, 128), None), ((3, 70, 33, 4, 128), None), ((2, 45, 130, 2, 32), None)]:
        TM.test_mha_core_no_window(gpu, shape, scale, F16)
    TM.test_relattn_dropout_consistency(gpu, F16)
    for packed in (False, True):
        TM.test_self_attention_block_parity(gpu, (3, 200, 2, 96), packed, F16)
    TM.test_mrte_block_parity(gpu, F16)


def test_encoder_ops_f16(gpu, f16):
    for shape in [(3, 37, 192), (2, 5, 512), (2, 9, 768)]:
        TE.test_res_ln_no_dropout(gpu, F16, 3e-3, shape)
    TE.test_wn_residual(gpu, F16)
    TE.test_wn_stack_node(gpu, F16)
    TE.test_coupling_flip_equals_torch_composition(gpu, F16)
    TE.test_mish_and_glu_chains_equal_torch(gpu, F16, 3e-3)
    TE.test_reparam_equals_torch_lines(gpu, F16, 2e-3)


def test_discriminator_losses_fold_f16(gpu, f16):
    TD.test_generator_step_through_discriminators(gpu, F16, 6e-3)
    for T in (20480, 1003, 77, 24):
        TF.test_fold_equals_torch_composition(gpu, T, F16)
    for layout in ("channels_last_views", "bct_contiguous"):
        TK.test_masked_kl_matches_reference(gpu, layout, F16, 173)


def test_a_bf16_tensor_is_refused_while_the_f16_build_is_selected(gpu, f16):
    """the two builds read the same 16 bits differently: a tensor of the other 16-bit type must be stopped in the binding"""
    from easevoice_trainer_amd.hip import lib as L

    with pytest.raises(L.EvtError):
        L.dt_code(torch.bfloat16)
    L.set_half(torch.bfloat16)
    with pytest.raises(L.EvtError):
        L.dt_code(F16)
