"""Per-shape timing of the fused conv kernels at the BASELINE s2 shapes (B=16): forward, backward-data,
backward-weight.  Prints a table (ms, TFLOP/s, algorithmic GB/s).  Run on the GPU box."""
import argparse
import ctypes as C
import json
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easevoice_trainer_amd.hip import conv as HC, lib as L  # noqa: E402


def time_fn(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--impl", type=int, default=0)
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--wonly", action="store_true", help="time the weight gradient only")
    a = ap.parse_args()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = torch.device("cuda:0")
    B = a.B
    shapes = []
    # HiFi-GAN resblock convs: (C, L) per stage, k in {3,7,11}, d in {1,3,5}
    for C_, Lx in [(256, 320), (128, 2560), (64, 5120), (32, 10240), (16, 20480)]:
        for k, d in [(3, 1), (7, 3), (11, 5), (11, 1)]:
            shapes.append((f"res C{C_} L{Lx} k{k} d{d}", B, Lx, C_, C_, k, 1, (k * d - d) // 2, d, 1, False, 0.1, 0))
            # the same layer as ResUnitFn runs it: operands activated by the neighbouring epilogues, plain here
            shapes.append((f"plain C{C_} L{Lx} k{k} d{d}", B, Lx, C_, C_, k, 1, (k * d - d) // 2, d, 1, False, 1.0, 0))
    ups = [(512, 256, 16, 10, 3, 32), (256, 128, 16, 8, 4, 320), (128, 64, 8, 2, 3, 2560), (64, 32, 2, 2, 0, 5120),
           (32, 16, 2, 2, 0, 10240)]
    for ci, co, k, u, p, Lx in ups:
        shapes.append((f"up {ci}->{co} k{k} s{u}", B, Lx, ci, co, k, u, p, 1, 1, True, 0.1, 0))
    # DiscriminatorP p=2 (real+fake => 2B*p sequences)
    H = 10240
    for ci, co, s in [(32, 128, 3), (128, 512, 3), (512, 1024, 3), (1024, 1024, 1)]:
        Hin = H = (H + 2 * 2 - 5) // 3 + 1 if ci == 32 else H
        shapes.append((f"dP2 {ci}->{co} s{s} H{Hin}", 2 * B * 2, Hin, ci, co, 5, s, 2, 1, 1, False, 1.0, 1))
        H = (Hin + 4 - 5) // s + 1
    # DiscriminatorP p=11 deep layer (short sequences)
    shapes.append(("dP11 1024->1024 s1 H23", 2 * B * 11, 23, 1024, 1024, 5, 1, 2, 1, 1, False, 1.0, 1))
    shapes.append(("dP11 512->1024 s3 H69", 2 * B * 11, 69, 512, 1024, 5, 3, 2, 1, 1, False, 1.0, 1))
    shapes.append(("dP5 512->1024 s3 H152", 2 * B * 5, 152, 512, 1024, 5, 3, 2, 1, 1, False, 1.0, 1))
    shapes.append(("dP11 128->512 s3 H207", 2 * B * 11, 207, 128, 512, 5, 3, 2, 1, 1, False, 1.0, 1))
    shapes.append(("dS 1024->1024 k5 L80", 2 * B, 80, 1024, 1024, 5, 1, 2, 1, 1, False, 1.0, 1))
    shapes.append(("dS grouped 16->64 k41 s4 g4", 2 * B, 20480, 16, 64, 41, 4, 20, 1, 4, False, 1.0, 1))
    shapes.append(("dS grouped 256->1024 k41 s4 g64", 2 * B, 1280, 256, 1024, 41, 4, 20, 1, 64, False, 1.0, 1))
    shapes.append(("WN in 192->384 k5 T200", B, 200, 192, 384, 5, 1, 2, 1, 1, False, 1.0, 0))
    shapes.append(("WN rs 192->384 k1 T200", B, 200, 192, 384, 1, 1, 0, 1, 1, False, 1.0, 0))
    shapes.append(("FFN 192->768 k3 T200", B, 200, 192, 768, 3, 1, 1, 1, 1, False, 1.0, 1))
    shapes.append(("FFN 768->192 k3 T200", B, 200, 768, 192, 3, 1, 1, 1, 1, False, 1.0, 0))
    shapes.append(("FFN 192->768 k3 T60", B, 60, 192, 768, 3, 1, 1, 1, 1, False, 1.0, 1))
    shapes.append(("FFN 768->192 k3 T60", B, 60, 768, 192, 3, 1, 1, 1, 1, False, 1.0, 0))
    # the s1 transformer's Linear layers as k = 1 convolutions over [32, 1024, C] (what-if: GEMMs on the conv kernels)
    for ci, co in [(512, 1536), (512, 512), (512, 2048), (2048, 512)]:
        shapes.append((f"s1 lin {ci}->{co}", 32, 1024, ci, co, 1, 1, 0, 1, 1, False, 1.0, 0))
    shapes.append(("conv_post 16->1 k7", B, 20480, 16, 1, 7, 1, 3, 1, 1, False, 0.01, 2))
    # first / last layers of the period discriminators (Cin = 1 / Cout = 1): [real ; fake] x period sequences
    shapes.append(("dP2 first 1->32 k5 s3", 2 * B * 2, 10240, 1, 32, 5, 3, 2, 1, 1, False, 1.0, 1))
    shapes.append(("dP7 first 1->32 k5 s3", 2 * B * 7, 2926, 1, 32, 5, 3, 2, 1, 1, False, 1.0, 1))
    shapes.append(("dP2 last 1024->1 k3", 2 * B * 2, 127, 1024, 1, 3, 1, 1, 1, 1, False, 1.0, 0))
    shapes.append(("dP7 last 1024->1 k3", 2 * B * 7, 37, 1024, 1, 3, 1, 1, 1, 1, False, 1.0, 0))

    if a.only:
        shapes = [sh for sh in shapes if any(tok in sh[0] for tok in a.only.split(','))]
    mods = nn.ModuleList()
    for (_, nseq, Lx, ci, co, k, s, p, d, g, tr, slope, oact) in shapes:
        mods.append(HC.EvtConv1d(ci, co, k, s, p, d, g, bias=True, transposed=tr, weight_norm=True))
    mods = mods.to(dev)
    bank = HC.WeightBank(mods, dtype, dev, impl=a.impl)
    bank.build_tables()
    bank.async_wgrad = False      # per-call timings below are taken on the current stream
    bank.defer_n = 0
    bank.fold()
    torch.cuda.synchronize()
    rows = []
    sz = 2 if dtype == torch.bfloat16 else 4
    for m, (name, nseq, Lx, ci, co, k, s, p, d, g, tr, slope, oact) in zip(mods, shapes):
        x = torch.randn(nseq, Lx, ci, device=dev).to(dtype)
        slot = m._slot
        y = HC._fwd(slot, x, None, slope, oact, 0.1)
        dy = torch.randn_like(y)
        lout = y.size(1)
        macs = nseq * (lout if not tr else Lx) * ci * co * k / g
        bytes_act = (x.numel() + y.numel()) * sz
        t_f = 1e9 if a.wonly else time_fn(lambda: HC._fwd(slot, x, None, slope, oact, 0.1), iters=a.iters)
        boact, by = oact, y
        if oact and L.lib().evt_conv1d_wants_plain_dy(C.byref(slot.params(nseq, Lx, slope, oact, 0.1))):
            boact, by = 0, None        # the autograd node pre-multiplies dy by the activation derivative (evt_dact_mul)
        t_d = 1e9 if a.wonly else time_fn(lambda: HC._bwd_data(slot, dy, by, x, None, nseq, Lx, slope, boact, 0.1), iters=a.iters)

        def wg():
            slot.wg_used, slot.wg_dirty = 0, False          # as in a step: the first (only) launch into this convolution's slabs
            HC._bwd_weight(slot, x, dy, by, nseq, Lx, slope, boact, 0.1)

        t_w = time_fn(wg, iters=a.iters)
        row = dict(name=name, gmac=macs / 1e9, fwd_ms=t_f, bwdd_ms=t_d, bwdw_ms=t_w,
                   fwd_tflops=2 * macs / t_f / 1e9, bwdd_tflops=2 * macs / t_d / 1e9, bwdw_tflops=2 * macs / t_w / 1e9,
                   fwd_gbs=bytes_act / t_f / 1e6)
        rows.append(row)
        print(f"{name:36s} {macs/1e9:8.2f} GMAC  fwd {t_f:8.3f} ms {row['fwd_tflops']:7.1f} TF {row['fwd_gbs']:7.0f} GB/s"
              f" | bwd-d {t_d:8.3f} ms {row['bwdd_tflops']:7.1f} TF | bwd-w {t_w:8.3f} ms {row['bwdw_tflops']:7.1f} TF",
              flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
