"""CPU check of the conv kernels' ADDRESSING: a numpy emulation of the prepared-weight images
(wn_fold_kernel's reg_idx/alt_idx) and of conv_igemm / conv_wgrad's index maps, driven by the same
launch geometry the C++ host code derives, compared with the CPU oracle.  It pins the derivations
(tap flip, polyphase split, offsets) without a GPU; the HIP kernels themselves are tested with -m gpu.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from easevoice_trainer_amd.hip import lib as L
from oracle import ops as O


def layout(dt, cin, cout, k, stride, pad, dil, groups, transposed):
    p = L.ConvParams(dt, 1, 64 + k * dil, cin, cout, k, stride, pad, dil, groups, int(transposed), 1.0, 0, 1.0, 0)
    lay = L.WLayout()
    assert L.lib().evt_conv1d_layout(C.byref(p), C.byref(lay)) == 0
    return lay


def make_images(w, lay):
    """w: [d0][d1][k] -> (REG, ALT) flat float arrays, following elementwise.hip reg_idx / alt_idx"""
    d0, d1, k = w.shape
    reg = np.zeros(lay.reg_elems, np.float64)
    alt = np.zeros(lay.alt_elems, np.float64)
    J = (k + lay.stride - 1) // lay.stride
    for a in range(d0):
        for b in range(d1):
            for kk in range(k):
                ch, cc = divmod(b, lay.reg_ck)
                reg[((a * lay.reg_nchunk + ch) * lay.reg_kp + kk) * lay.reg_ck + cc] = w[a, b, kk]
                ch, cc = divmod(a, lay.alt_ck)
                if lay.stride == 1:
                    t = k - 1 - kk
                    alt[((b * lay.alt_nchunk + ch) * lay.alt_kp + t) * lay.alt_ck + cc] = w[a, b, kk]
                else:
                    ph, j = kk % lay.stride, kk // lay.stride
                    jp = J - 1 - j
                    alt[(((ph * d1 + b) * lay.alt_nchunk + ch) * lay.alt_kp + jp) * lay.alt_ck + cc] = w[a, b, kk]
    return reg, alt


def igemm_emulate(x, wimg, A, B, CK, nchunk, KHp, nphase, wps, s_in, dil, off_in, s_out, off_out, off_out_phase, Q,
                  Lout):
    """x [nseq][Lin][B] -> y [nseq][Lout][A] with conv_igemm's index maps"""
    nseq, Lin, _ = x.shape
    y = np.zeros((nseq, Lout, A))
    W = wimg.reshape(nphase, -1)[:, : A * nchunk * KHp * CK].reshape(nphase, A, nchunk, KHp, CK) if nphase > 1 else \
        wimg[: A * nchunk * KHp * CK].reshape(1, A, nchunk, KHp, CK)
    for ph in range(nphase):
        for q in range(Q):
            orow = q * s_out + off_out + ph * off_out_phase
            if orow < 0 or orow >= Lout:
                continue
            for tap in range(KHp):
                r = q * s_in + tap * dil + off_in
                if r < 0 or r >= Lin:
                    continue
                xv = x[:, r, :].reshape(nseq, nchunk, CK)
                y[:, orow, :] += np.einsum("snc,anc->sa", xv, W[ph, :, :, tap, :])
    return y


CASES = [
    (32, 32, 11, 1, 25, 5, False), (16, 16, 7, 1, 9, 3, False), (64, 32, 3, 1, 1, 1, False),
    (32, 64, 5, 3, 2, 1, False), (32, 32, 5, 1, 2, 1, False), (16, 32, 4, 2, 1, 1, False),
    (64, 32, 16, 10, 3, 1, True), (32, 16, 16, 8, 4, 1, True), (32, 16, 8, 2, 3, 1, True), (32, 16, 2, 2, 0, 1, True),
]


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("case", CASES)
def test_igemm_index_math(case, dt):
    cin, cout, k, stride, pad, dil, transposed = case
    torch.manual_seed(1)
    nseq, Lin = 2, 37
    lay = layout(dt, cin, cout, k, stride, pad, dil, 1, transposed)
    d0, d1 = (cin, cout) if transposed else (cout, cin)
    w = torch.randn(d0, d1, k, dtype=torch.float64)
    x = torch.randn(nseq, cin, Lin, dtype=torch.float64, requires_grad=True)
    y_ref = O.conv_block(x, w, stride=stride, pad=pad, dil=dil, transposed=transposed)
    Lout = y_ref.size(2)
    dy = torch.randn_like(y_ref)
    (dx_ref,) = torch.autograd.grad(y_ref, x, dy)
    reg, alt = make_images(w.numpy(), lay)
    xn = x.detach().numpy().transpose(0, 2, 1)
    dyn = dy.numpy().transpose(0, 2, 1)
    J = (k + stride - 1) // stride

    # ---- forward (evt_conv1d_fwd) ----
    if not transposed:
        y = igemm_emulate(xn, reg, cout, cin, lay.reg_ck, lay.reg_nchunk, lay.reg_kp, 1, 0, stride, dil, -pad, 1, 0, 0,
                          Lout, Lout)
    elif stride == 1:
        y = igemm_emulate(xn, alt, cout, cin, lay.alt_ck, lay.alt_nchunk, lay.alt_kp, 1, 0, 1, 1, pad - (k - 1), 1, 0,
                          0, Lout, Lout)
    else:
        y = igemm_emulate(xn, alt, cout, cin, lay.alt_ck, lay.alt_nchunk, lay.alt_kp, stride, 0, 1, 1, -(J - 1),
                          stride, -pad, 1, (Lout - 1 + pad) // stride + 1, Lout)
    np.testing.assert_allclose(y, y_ref.detach().numpy().transpose(0, 2, 1), atol=1e-9)

    # ---- backward-data (evt_conv1d_bwd_data) ----
    if not transposed:
        if stride == 1:
            dx = igemm_emulate(dyn, alt, cin, cout, lay.alt_ck, lay.alt_nchunk, lay.alt_kp, 1, 0, 1, dil,
                               pad - (k - 1) * dil, 1, 0, 0, Lin, Lin)
        else:
            dx = igemm_emulate(dyn, alt, cin, cout, lay.alt_ck, lay.alt_nchunk, lay.alt_kp, stride, 0, 1, 1, -(J - 1),
                               stride, -pad, 1, (Lin - 1 + pad) // stride + 1, Lin)
    else:
        dx = igemm_emulate(dyn, reg, cin, cout, lay.reg_ck, lay.reg_nchunk, lay.reg_kp, 1, 0, stride, 1, -pad, 1, 0, 0,
                           Lin, Lin)
    np.testing.assert_allclose(dx, dx_ref.numpy().transpose(0, 2, 1), atol=1e-9)

    # ---- backward-weight (conv_wgrad): dW[a][chunk][t][cc] += A[q][a] * B[q*s + t*dil - pad][b] ----
    wl = w.clone().requires_grad_(True)
    y2 = O.conv_block(x.detach(), wl, stride=stride, pad=pad, dil=dil, transposed=transposed)
    (dw_ref,) = torch.autograd.grad(y2, wl, dy)
    Aop, Bop = (dyn, xn) if not transposed else (xn, dyn)
    CA, CB = Aop.shape[2], Bop.shape[2]
    dW = np.zeros((CA, lay.reg_nchunk, lay.reg_kp, lay.reg_ck))
    for q in range(Aop.shape[1]):
        for t in range(k):
            r = q * stride + t * dil - pad
            if 0 <= r < Bop.shape[1]:
                dW[:, :, t, :] += np.einsum("sa,snc->anc", Aop[:, q, :], Bop[:, r, :].reshape(nseq, lay.reg_nchunk, lay.reg_ck))
    got = dW[:, :, :k, :].transpose(0, 1, 3, 2).reshape(CA, CB, k)
    np.testing.assert_allclose(got, dw_ref.numpy(), atol=1e-9)


def test_library_exports_header_symbols():
    """every function declared in include/evt.h is exported by the built library (no compute calls)"""
    import os
    import re

    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "evt.h")).read()
    names = set(re.findall(r"\b(evt_[a-z0-9_]+)\s*\(", hdr))
    assert names, "no declarations parsed"
    import torch

    # both builds of the sources export exactly the header: libevt_hip.so (bfloat16) and libevt_hip_f16.so (IEEE half)
    for half, code, tag in ((torch.bfloat16, 1, "half = bf16"), (torch.float16, 2, "half = f16")):
        L.set_half(half)
        lib = L.lib()
        missing = [n for n in sorted(names) if not hasattr(lib, n)]
        assert not missing, f"declared in evt.h but not exported by the {half} build: {missing}"
        assert lib.evt_half_dtype() == code and tag in lib.evt_version().decode()
    L.set_half(torch.bfloat16)


def test_workspace_bytes_query():
    """evt_workspace_bytes: host-only arithmetic, no launch"""
    import ctypes as C

    lib = L.lib()

    def ws(op, *dims):
        arr = (C.c_int64 * max(len(dims), 1))(*dims)
        return lib.evt_workspace_bytes(op, arr, len(dims))

    assert ws(0) == 0                                               # EVT_WS_NONE
    assert ws(1, 16, 20480, 2048, 640, 128) == 4 * lib.evt_mel_workspace_floats(16, 20480, 2048, 640, 128) > 0
    assert ws(2, 32, 16, 1024) == 4 * 32 * 16 * 1024                # attention backward delta
    assert ws(3, 16, 2, 200) == 4 * 16 * 2 * 200                    # relative attention backward delta
    assert ws(4) == 8                                               # masked KL: (sum, live frames)
    assert ws(1, 16, 20480) == -1 and ws(99) == -1                  # wrong arity / unknown op


def test_wn_layer_lds_swizzles():
    """csrc/wn_layer.hip: the MFMA B-operand reads (ds_read_b128 of lane (n, g) at row r + n + tap, 16-byte slot 4 ch + g) are
    conflict-free with the kernels' slot swizzles at every row offset, tap and chunk -- the bank arithmetic of
    MI355X_MICROARCH.md (LDS): a b128 read is served in four groups of 16 lanes, bank = (byte address / 4) mod 64, lanes of a
    group that touch one bank at different addresses serialise.  (The first version of the kernels used swizzles that are
    conflict-free for "16 consecutive lanes" -- not the hardware's groups -- and the counters showed a third of the LDS cycles
    as conflicts.)  The same replay confirms resunit_wide's swizzles."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
              list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]

    def cycles(addr_of_lane):
        tot = 0
        for grp in groups:
            banks = {}
            for lane in grp:
                a = addr_of_lane(lane)
                for b in range(4):
                    banks.setdefault((a // 4 + b) % 64, set()).add(a)
            tot += max(len(v) for v in banks.values())
        return tot

    def worst(pitch, swz, chunks, taps, dil=1):
        w = 0
        for base in range(16):
            for tap in range(taps):
                for ch in range(chunks):
                    def addr(lane):
                        row = base + (lane & 15) + tap * dil
                        return row * pitch + swz(row, ch * 4 + (lane >> 4)) * 16
                    w = max(w, cycles(addr))
        return w

    assert worst(384, lambda r, s: s ^ (r & 7), 6, 5) == 4                  # wn_layer_fwd: wslot, 192 channels
    assert worst(768, lambda r, s: s ^ ((r & 7) << 1), 12, 5) == 4          # wn_layer_bwd: wslot2, 384 channels
    assert worst(384, lambda r, s: s ^ ((r >> 1) & 7), 6, 5) > 4            # the first version's choice was not
    for d in (1, 3, 5):                                                     # resunit_wide, dilated taps
        assert worst(128, lambda r, s: s ^ (((r >> 1) & 3) << 1), 2, 11, d) == 4
        assert worst(256, lambda r, s: s ^ ((r & 7) << 1), 4, 11, d) == 4
    # the swizzles keep a slot inside its row
    assert all(0 <= (s ^ (r & 7)) < 24 for r in range(64) for s in range(24))
    assert all(0 <= (s ^ ((r & 7) << 1)) < 48 for r in range(64) for s in range(48))
