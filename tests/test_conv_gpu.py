"""GPU parity: fused Conv1d / ConvTranspose1d HIP kernels (through the C ABI) vs the CPU oracle.

fp32 tolerance 1e-3 relative to the tensor's max-abs (north_star: 1e-3 relative fp32); bf16 runs are
checked at 3e-2 against the fp32 oracle evaluated on bf16-rounded inputs.
"""
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu

# (cin, cout, k, stride, pad, dil, groups, transposed, wn, L, nseq)  — shapes of the real model, short L
CASES = [
    # HiFi-GAN ResBlock convs (modules.py:226-296): C in {256,128,64,32,16}, k in {3,7,11}, d in {1,3,5}
    (32, 32, 3, 1, 1, 1, 1, False, True, 100, 3),
    (32, 32, 11, 1, 25, 5, 1, False, True, 150, 2),
    (16, 16, 7, 1, 9, 3, 1, False, True, 130, 2),
    (16, 16, 11, 1, 5, 1, 1, False, True, 70, 2),
    (64, 64, 7, 1, 3, 1, 1, False, True, 96, 2),
    (128, 128, 3, 1, 3, 3, 1, False, True, 80, 2),
    (256, 256, 11, 1, 5, 1, 1, False, True, 40, 2),
    # conv_pre 192->512 k7, WN in_layer 192->384 k5, FFN 192->768 k3, 1x1 res_skip
    (192, 512, 7, 1, 3, 1, 1, False, False, 32, 2),
    (192, 384, 5, 1, 2, 1, 1, False, True, 50, 2),
    (192, 768, 3, 1, 1, 1, 1, False, False, 37, 2),
    (192, 384, 1, 1, 0, 1, 1, False, True, 45, 2),
    (96, 192, 1, 1, 0, 1, 1, False, False, 45, 2),
    # 1x1 projections that used to be vendor GEMMs: flow post 192->96, the style encoder's linears (704->128, 128->512)
    # and the weight-normed conditioning layers applied to a [B, 512] vector as ONE sequence of B rows
    (192, 96, 1, 1, 0, 1, 1, False, False, 45, 2),
    (192, 576, 1, 1, 0, 1, 1, False, False, 200, 3),     # packed q|k|v projection, 600 rows (ragged 64-row stage)
    (704, 128, 1, 1, 0, 1, 1, False, False, 40, 2),
    (128, 512, 1, 1, 0, 1, 1, False, False, 33, 2),
    (512, 1536, 1, 1, 0, 1, 1, False, True, 16, 1),
    (512, 512, 1, 1, 0, 1, 1, False, False, 16, 1),
    (512, 1536, 1, 1, 0, 1, 1, False, True, 5, 1),       # fewer than 16 rows (a batch of 5): rows16 zero-fills the tile
    # ups (models.py:424-436): (k,u,p) = (16,10,3) (16,8,4) (8,2,3) (2,2,0)
    (64, 32, 16, 10, 3, 1, 1, True, True, 12, 2),
    (64, 32, 16, 8, 4, 1, 1, True, True, 12, 2),
    (32, 16, 8, 2, 3, 1, 1, True, True, 33, 2),
    (32, 16, 2, 2, 0, 1, 1, True, True, 33, 2),
    # DiscriminatorP (k5,s3) and (k5,s1), DiscriminatorS dense k5 (models.py:487-536,571)
    (32, 128, 5, 3, 2, 1, 1, False, True, 100, 4),
    (128, 64, 5, 3, 2, 1, 1, False, True, 23, 6),
    (64, 64, 5, 1, 2, 1, 1, False, True, 23, 6),
    (64, 128, 5, 1, 2, 1, 1, False, True, 37, 5),      # short sequences -> split-unit tiling (bf16)
    (32, 64, 3, 1, 1, 1, 1, False, True, 51, 7),
    (128, 128, 5, 3, 2, 1, 1, False, True, 152, 9),    # strided, output length 51
    # degenerate / grouped shapes -> direct kernels: conv_post 16->1 k7 no bias, D first layers, D post,
    # DiscriminatorS grouped k41 s4
    (16, 1, 7, 1, 3, 1, 1, False, False, 90, 2),
    (1, 32, 5, 3, 2, 1, 1, False, True, 200, 4),
    (1, 16, 15, 1, 7, 1, 1, False, True, 120, 2),
    (64, 1, 3, 1, 1, 1, 1, False, True, 40, 3),
    (1024, 1, 3, 1, 1, 1, 1, False, True, 127, 4),
    (16, 64, 41, 4, 20, 1, 4, False, True, 300, 2),
    (64, 256, 41, 4, 20, 1, 16, False, True, 1000, 3),
    (256, 1024, 41, 4, 20, 1, 64, False, True, 130, 2),
    (1024, 1024, 41, 4, 20, 1, 256, False, True, 90, 2),
]
FUSIONS = [
    dict(in_slope=1.0, out_act=0, out_slope=1.0, res=False),
    dict(in_slope=0.1, out_act=0, out_slope=1.0, res=True),
    dict(in_slope=1.0, out_act=1, out_slope=0.1, res=False),
    dict(in_slope=0.01, out_act=2, out_slope=1.0, res=False),
]


HALVES = (torch.bfloat16, torch.float16)
HALF = torch.bfloat16      # the 16-bit type of the hard-wired cases; tests/test_fp16_ops_gpu.py re-runs them with torch.float16


def _run_case(gpu, case, fusion, dtype, impl, report=None):
    from easevoice_trainer_amd.hip import conv as HC

    cin, cout, k, stride, pad, dil, groups, transposed, wn, Lin, nseq = case
    torch.manual_seed(hash(case) % 100000)
    m = HC.EvtConv1d(cin, cout, k, stride, pad, dil, groups, bias=(cout != 1 or wn), transposed=transposed,
                     weight_norm=wn)
    if wn:
        with torch.no_grad():
            m.weight_g.mul_(torch.rand_like(m.weight_g) + 0.5)
    x = torch.randn(nseq, cin, Lin)  # reference layout
    lout = m.lout(Lin)
    res = torch.randn(nseq, cout, lout) if fusion["res"] else None
    dy = torch.randn(nseq, cout, lout)
    if dtype in HALVES:
        x, dy = x.to(dtype).float(), dy.to(dtype).float()
        res = res.to(dtype).float() if res is not None else None

    # ---- oracle (CPU fp32, [B,C,L]) ----
    xo = x.clone().requires_grad_(True)
    ro = res.clone().requires_grad_(True) if res is not None else None
    po = {n_: p.detach().clone().requires_grad_(True) for n_, p in m.named_parameters()}
    w = O.weight_norm_fold(po["weight_v"], po["weight_g"]) if wn else po["weight"]
    if dtype in HALVES:
        w = w + (w.detach().to(dtype).float() - w.detach())  # straight-through 16-bit rounding of weights
    yo = O.conv_block(xo, w, po.get("bias"), ro, stride=stride, pad=pad, dil=dil, groups=groups,
                      transposed=transposed, in_slope=fusion["in_slope"], out_act=fusion["out_act"],
                      out_slope=fusion["out_slope"])
    if fusion["out_act"] == 1:
        # the library takes the leaky-relu derivative from the STORED activation output; an output that underflows the
        # storage type to zero (IEEE half: |y| < 3e-8; met once in 1.2 M elements, tools/exp/diag_ring_f16.py) has lost its
        # sign -- as it has in the reference's fp16 autocast, whose saved pre-activation is half as well -- and an output
        # within the accumulation noise of zero (~1e-6 of a sum of 640 products) may have either.  Such elements (about one
        # in ten thousand) get no upstream gradient here, so that the comparison is about the kernels and not a coin toss.
        dy = dy.masked_fill(yo.detach().abs() < 1e-4, 0.0)
    yo.backward(dy)

    # ---- HIP path (channels-last) ----
    m = m.to(gpu)
    bank = HC.WeightBank(m, dtype, gpu, impl=impl)
    bank.build_tables()
    bank.fold()
    xg = x.transpose(1, 2).contiguous().to(gpu, dtype).requires_grad_(True)
    rg = res.transpose(1, 2).contiguous().to(gpu, dtype).requires_grad_(True) if res is not None else None
    yg = m(xg, rg, fusion["in_slope"], fusion["out_act"], fusion["out_slope"])
    yg.backward(dy.transpose(1, 2).contiguous().to(gpu, dtype))
    bank.grads()
    torch.cuda.synchronize()

    tol = {torch.float32: 1e-3, torch.bfloat16: 3e-2, torch.float16: 6e-3}[dtype]

    def close(a, b, name):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        scale = b.abs().max().item() + 1e-6
        err = (a - b).abs().max().item() / scale
        if report is not None:
            report.append((name, err))
            return
        assert err < tol, f"{name}: rel err {err:.3e} (tol {tol}) case={case} fusion={fusion} dtype={dtype} impl={impl}"

    close(yg.transpose(1, 2), yo, "y")
    close(xg.grad.transpose(1, 2), xo.grad, "dx")
    if rg is not None:
        close(rg.grad.transpose(1, 2), ro.grad, "dres")
    for n_, p in m.named_parameters():
        close(p.grad, po[n_].grad, f"d{n_}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("impl", [1, 0], ids=["naive", "auto"])
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_conv_parity(gpu, ci, impl, dtype):
    case = CASES[ci]
    for fusion in FUSIONS:
        if fusion["res"] and case[7]:
            continue
        _run_case(gpu, case, fusion, dtype, impl)


# wide layers of DiscriminatorP (models.py:487-536) at sizes that select the LDS-DMA GEMM path (conv_deep):
# forward, backward-data (stride 1 and polyphase) and the pre-multiplied activation derivative
DEEP_CASES = [
    (512, 1024, 5, 3, 2, 1, 1, False, True, 225, 40),
    (1024, 1024, 5, 1, 2, 1, 1, False, True, 37, 90),     # sequences shorter than a 128-position tile
    (128, 512, 5, 3, 2, 1, 1, False, True, 207, 130),
    (256, 256, 7, 1, 9, 3, 1, False, True, 700, 20),      # dilated, long sequences
    (32, 128, 5, 3, 2, 1, 1, False, True, 1200, 64),      # K side of 32 channels -> 32-channel stages
    (1024, 1024, 5, 1, 2, 1, 1, False, True, 127, 40),    # 5080 positions: enough 128 x 128 tiles to fill the chip
]


# latency-bound mid-size layers (WN in/res_skip, FFN) -> the 64 x 64 ring-pipelined variant
RING_CASES = [
    (192, 384, 5, 1, 2, 1, 1, False, True, 200, 16),
    (192, 384, 1, 1, 0, 1, 1, False, True, 200, 16),
    (192, 768, 3, 1, 1, 1, 1, False, False, 61, 16),     # ragged last position tile
    (768, 192, 3, 1, 1, 1, 1, False, False, 60, 16),
    (128, 64, 5, 3, 2, 1, 1, False, True, 310, 24),       # strided forward, polyphase backward-data
    (192, 576, 1, 1, 0, 1, 1, False, False, 200, 16),     # the packed q|k|v projection (9 output tiles of 64)
]


# late vocoder stages (C = 16 / 32, stride 1, dilated): the weights-in-registers persistent kernel
NARROW_CASES = [
    (16, 16, 11, 1, 25, 5, 1, False, True, 700, 8),
    (16, 16, 3, 1, 1, 1, 1, False, True, 1100, 4),
    (16, 16, 7, 1, 9, 3, 1, False, True, 333, 14),        # ragged tiles, several sequences
    (32, 32, 7, 1, 9, 3, 1, False, True, 600, 8),
    (32, 32, 3, 1, 1, 1, 1, False, True, 2100, 2),
]


@pytest.mark.parametrize("ci", range(len(NARROW_CASES)))
def test_conv_narrow_parity(gpu, ci):
    from easevoice_trainer_amd.hip import conv as HC

    case = NARROW_CASES[ci]
    for fusion in (FUSIONS[0], FUSIONS[2], FUSIONS[1]):
        HC.set_trace([])
        try:
            _run_case(gpu, case, fusion, HALF, 0)
            tags = {(r[1], r[0].split(",")[0]) for r in HC.TRACE}
        finally:
            HC.set_trace(None)
        if fusion["in_slope"] == 1.0:
            assert ("fwd", "conv_narrow<bf16") in tags, tags
        assert ("bwd_data", "conv_narrow<bf16") in tags, tags


@pytest.mark.parametrize("ci", range(len(RING_CASES)))
def test_conv_ring_parity(gpu, ci):
    from easevoice_trainer_amd.hip import conv as HC

    case = RING_CASES[ci]
    for fusion in (FUSIONS[2], FUSIONS[0]):
        HC.set_trace([])
        try:
            _run_case(gpu, case, fusion, HALF, 0)
            tags = {(r[1], r[0]) for r in HC.TRACE}
        finally:
            HC.set_trace(None)
        assert ("fwd", "conv_ring<bf16, 64, 64, 64, x4>") in tags, tags
        if case[0] % 64 == 0 and case[1] % 64 == 0:
            assert ("bwd_data", "conv_ring<bf16, 64, 64, 64, x4>") in tags, tags
        assert any(k == "bwd_weight" and t.startswith(("wgrad_ring<", "wgrad_halo<")) for k, t in tags), tags


@pytest.mark.parametrize("ci", range(len(DEEP_CASES)))
def test_conv_deep_parity(gpu, ci):
    from easevoice_trainer_amd.hip import conv as HC

    case = DEEP_CASES[ci]
    for fusion in (FUSIONS[2], FUSIONS[0]):
        HC.set_trace([])
        try:
            _run_case(gpu, case, fusion, HALF, 0)
            tags = {(r[1], r[0]) for r in HC.TRACE}
        finally:
            HC.set_trace(None)
        deep = {"conv_deep<bf16, 128, 128, 64>", "conv_deep32<bf16, 128, 128, 32>"}
        assert any(k == "fwd" and t in deep for k, t in tags), tags
        if case[0] % 128 == 0:
            assert any(k == "bwd_data" and t in deep for k, t in tags), tags
        if case[1] % 128 == 0 and case[0] % 64 == 0:
            assert any(k == "bwd_weight" and t.startswith(("wgrad_deep<", "wgrad_halo<")) for k, t in tags), tags


# the vocoder's upsamplers at the step's own geometry (models.py:424-436, 457-460: leaky-relu(0.1) on load): the input is
# activated once (evt_conv1d_wants_plain_x) and the polyphase forward then runs on the LDS-DMA kernels
UPS_CASES = [
    (512, 256, 16, 10, 3, 1, 1, True, True, 32, 16),
    (256, 128, 16, 8, 4, 1, 1, True, True, 320, 4),
    (128, 64, 8, 2, 3, 1, 1, True, True, 640, 4),
    (128, 64, 8, 2, 3, 1, 1, True, True, 333, 3),          # ragged: 999 positions per phase
]


@pytest.mark.parametrize("ci", range(len(UPS_CASES)))
def test_upsampler_preactivated_parity(gpu, ci):
    from easevoice_trainer_amd.hip import conv as HC

    case = UPS_CASES[ci]
    fusion = dict(in_slope=0.1, out_act=0, out_slope=1.0, res=False)
    for plain in (True, False):
        old = HC.PLAIN_X
        HC.PLAIN_X = plain
        HC.set_trace([])
        try:
            _run_case(gpu, case, fusion, HALF, 0)
            tags = {(r[1], r[0]) for r in HC.TRACE}
        finally:
            HC.set_trace(None)
            HC.PLAIN_X = old
        fast = ("conv_ring<bf16, 64, 64, 64, x4>", "conv_deep<bf16, 128, 128, 64>", "conv_deep32<bf16, 128, 128, 32>")
        assert any(k == "fwd" and t in fast for k, t in tags) == plain, tags
        assert (("elt", "lrelu_kernel") in tags) == plain, tags


# stride-1 layers with 3 / 5 / 7 / 11 taps and sequences long enough for sequence-local K stages -> wgrad_halo (one staged
# window of x serves every tap): vocoder stages (dilated), WN in_layers, encoder FFN; tails that are not multiples of 64 / 32
HALO_CASES = [
    (128, 128, 11, 1, 5, 1, 1, False, True, 640, 4),
    (128, 128, 11, 1, 25, 5, 1, False, True, 320, 3),      # dilation 5: 114 window rows
    (128, 128, 7, 1, 9, 3, 1, False, True, 300, 5),        # tail of 44 positions
    (64, 64, 11, 1, 15, 3, 1, False, True, 512, 4),
    (64, 64, 3, 1, 1, 1, 1, False, True, 1000, 2),         # tail of 40
    (256, 256, 11, 1, 5, 1, 1, False, True, 320, 4),
    (256, 256, 3, 1, 5, 5, 1, False, True, 320, 16),
    (192, 384, 5, 1, 2, 1, 1, False, True, 200, 16),       # WN in_layer: tail of 8 (second half of the stage skipped)
    (192, 768, 3, 1, 1, 1, 1, False, False, 200, 16),      # FFN
    (768, 192, 3, 1, 1, 1, 1, False, False, 200, 16),
    (512, 256, 7, 1, 3, 1, 1, False, True, 128, 12),       # 128-channel tiles (EVT_HALO_MA=4 forces them elsewhere)
    (32, 32, 11, 1, 25, 5, 1, False, True, 1536, 6),       # 32 dy channels: the A tile has the window's geometry
    (64, 32, 7, 1, 3, 1, 1, False, True, 1000, 4),
]


# the dispatcher keeps the halo kernel to 64 dy channels (where it is the fastest); EVT_HALO_ALL=1 -- read once per process,
# hence the child process -- runs it on every shape it supports
def test_wgrad_halo_all_shapes_in_child_process(gpu):
    import os
    import subprocess
    import sys

    if os.environ.get("EVT_HALO_ALL") is not None:
        pytest.skip("this is the child")
    env = dict(os.environ, EVT_HALO_ALL="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k",
                        "test_wgrad_halo_parity or test_wgrad_slabs_are_deterministic"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def _halo_case_ids():
    """the cases the dispatcher sends to wgrad_halo in this process: 64 (and 32) dy channels, all of them under
    EVT_HALO_ALL=1 (the child process of the test above)"""
    import os

    every = os.environ.get("EVT_HALO_ALL") is not None
    return [i for i, c in enumerate(HALO_CASES) if every or c[1] in (32, 64)]


@pytest.mark.parametrize("ci", _halo_case_ids())
def test_wgrad_halo_parity(gpu, ci):
    from easevoice_trainer_amd.hip import conv as HC

    case = HALO_CASES[ci]
    for fusion in (FUSIONS[0], FUSIONS[2]):
        HC.set_trace([])
        try:
            _run_case(gpu, case, fusion, HALF, 0)
            tags = {(r[1], r[0]) for r in HC.TRACE}
        finally:
            HC.set_trace(None)
        assert any(k == "bwd_weight" and t.startswith("wgrad_halo<") for k, t in tags), tags


def _wgrad_twice(gpu, case, env):
    """dW / dbias of one convolution from two separate backward passes (fresh slabs each) + one accumulated pair"""
    import os

    from easevoice_trainer_amd.hip import conv as HC

    cin, cout, k, stride, pad, dil, groups, transposed, wn, Lin, nseq = case
    old = {k_: os.environ.get(k_) for k_ in env}
    os.environ.update(env)
    try:
        torch.manual_seed(5)
        m = HC.EvtConv1d(cin, cout, k, stride, pad, dil, groups, bias=True, transposed=transposed, weight_norm=wn).to(gpu)
        bank = HC.WeightBank(m, HALF, gpu)
        bank.build_tables()
        bank.fold()
        x = torch.randn(nseq, Lin, cin, device=gpu).to(HALF)
        dy = torch.randn(nseq, m.lout(Lin), cout, device=gpu).to(HALF)
        outs = []
        for reps in (1, 1, 2):
            for p in m.parameters():
                p.grad = None if p.grad is None else p.grad.zero_()
            bank.zero_dw()
            for _ in range(reps):
                xg = x.clone().requires_grad_(True)
                m(xg).backward(dy)
            bank.grads()
            torch.cuda.synchronize()
            outs.append({n_: p.grad.detach().clone() for n_, p in m.named_parameters()})
        return outs
    finally:
        for k_, v in old.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v


@pytest.mark.parametrize("ci", [0, 2, 5, 7, 10])
def test_wgrad_slabs_are_deterministic(gpu, ci):
    """split-K through slabs: two runs give the SAME BITS (fp32 atomics do not: the order of the additions is the order
    in which blocks finish), a second backward into the same slabs accumulates (2x), and the result equals the atomic
    path up to summation order"""
    case = HALO_CASES[ci]
    a, b, twice = _wgrad_twice(gpu, case, {"EVT_WGRAD_PARTS": "1"})
    for n_ in a:
        assert torch.equal(a[n_], b[n_]), (n_, (a[n_] - b[n_]).abs().max())
        ref = a[n_].abs().max().item() + 1e-6
        assert (twice[n_] - 2 * a[n_]).abs().max().item() / ref < 1e-5, n_
    c = _wgrad_twice(gpu, case, {"EVT_WGRAD_PARTS": "0"})[0]
    for n_ in a:
        ref = a[n_].abs().max().item() + 1e-6
        assert (a[n_] - c[n_]).abs().max().item() / ref < 1e-4, n_


@pytest.mark.parametrize("case", [(512, 1024, 5, 3, 2, 1, 1, False, True, 225, 40), (1024, 1024, 5, 1, 2, 1, 1, False, True, 37, 90),
                                  (192, 384, 1, 1, 0, 1, 1, False, True, 200, 16), (128, 64, 5, 3, 2, 1, 1, False, True, 310, 24),
                                  # the two-launch reductions (scratch rows + fold.hip)
                                  (16, 16, 11, 1, 25, 5, 1, False, True, 2100, 8), (32, 32, 7, 1, 9, 3, 1, False, True, 1500, 8),
                                  (16, 1, 7, 1, 3, 1, 1, False, True, 3000, 4), (1024, 1, 3, 1, 1, 1, 1, False, True, 127, 24),
                                  (1, 32, 5, 3, 2, 1, 1, False, True, 3000, 12), (1, 16, 15, 1, 7, 1, 1, False, True, 2000, 6)],
                         ids=["deep_s3", "deep_short", "ring_k1", "ring_s3", "narrow16", "narrow32", "cout1_16", "cout1_1024",
                              "cin1_32", "cin1_16"])
def test_wgrad_deep_ring_slabs_are_deterministic(gpu, case):
    a, b, twice = _wgrad_twice(gpu, case, {"EVT_WGRAD_PARTS": "1"})
    c = _wgrad_twice(gpu, case, {"EVT_WGRAD_PARTS": "0"})[0]
    for n_ in a:
        assert torch.equal(a[n_], b[n_]), (n_, (a[n_] - b[n_]).abs().max())
        ref = a[n_].abs().max().item() + 1e-6
        assert (twice[n_] - 2 * a[n_]).abs().max().item() / ref < 1e-5, n_
        assert (a[n_] - c[n_]).abs().max().item() / ref < 1e-4, n_
