"""Timing of csrc/gemm256.hip at the s1 layer shapes, with its ablation variants (evt_debug_gemm256_variant), next to the
128 x 128 kernels of conv_deep.hip (EVT_NO_GEMM256 in a second process).  HIP events over `--iters` launches each."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(32768, 1536, 512), (32768, 512, 512), (32768, 2048, 512), (32768, 512, 2048), (16384, 1536, 512)]
VARIANTS = {0: "full", 8: "no stores", 2: "no DMA after prologue", 1: "no MFMA", 4: "no fragment reads", 5: "no MFMA, no reads",
            6: "no DMA, no reads", 9: "no MFMA, no stores", 16: "stores folded onto 256 rows (L2-resident output)",
            17: "no MFMA, stores folded"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--variants", default="0,8,2,1,4,5,6,9")
    args = ap.parse_args()
    from easevoice_trainer_amd.hip import lib as L
    from easevoice_trainer_amd.hip.linear import LinearBank, gemm_fwd

    dev = torch.device("cuda:0")
    lib = L.lib()
    lib.evt_debug_gemm256_variant.restype = None
    out = {}
    for M, N, K in SHAPES:
        w = torch.nn.Parameter(torch.randn(N, K, device=dev) * K ** -0.5)
        b = torch.nn.Parameter(torch.randn(N, device=dev) * 0.1)
        bank = LinearBank([("t", w, b)], torch.bfloat16, dev)
        bank.prepare()
        slot = w._evt_slot
        x = torch.randn(M, K, device=dev).bfloat16()
        row = {}
        for v in [int(t) for t in args.variants.split(",")]:
            lib.evt_debug_gemm256_variant(v)
            for _ in range(3):
                gemm_fwd(slot, x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                gemm_fwd(slot, x)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            row[VARIANTS.get(v, str(v))] = dict(us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1))
        lib.evt_debug_gemm256_variant(0)
        out[f"{M}x{N}x{K}"] = dict(fused=slot.fused(M, False), **row)
        print(f"{M}x{N}x{K}", json.dumps(out[f"{M}x{N}x{K}"]), flush=True)


if __name__ == "__main__":
    main()
