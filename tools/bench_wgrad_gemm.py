"""Timing of the s1 dense-layer weight gradient (evt_gemm_bf16_bwd_weight -> wgrad_gemm in csrc/conv_deep.hip) at the layer
shapes of the s1 transformer; HIP events over --iters launches.  usage: python tools/bench_wgrad_gemm.py [--iters N]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(32768, 1536, 512), (32768, 512, 512), (32768, 2048, 512), (32768, 512, 2048), (16384, 1536, 512)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--no-bias", action="store_true", help="weight gradient only (no fused column sums of dy)")
    args = ap.parse_args()
    from easevoice_trainer_amd.hip.linear import LinearBank, gemm_bwd_weight

    dev = torch.device("cuda:0")
    tot = 0.0
    for M, N, K in SHAPES:
        w = torch.nn.Parameter(torch.randn(N, K, device=dev) * K ** -0.5)
        b = torch.nn.Parameter(torch.randn(N, device=dev) * 0.1)
        bank = LinearBank([("t", w, b)], torch.bfloat16, dev)
        bank.prepare()
        slot = w._evt_slot
        w._evt_grad_view = torch.zeros(N, K, device=dev)          # the engine's arena views: no allocation in the timed loop
        b._evt_grad_view = torch.zeros(N, device=dev)
        x = torch.randn(M, K, device=dev).bfloat16()
        dy = torch.randn(M, N, device=dev).bfloat16()
        for _ in range(3):
            gemm_bwd_weight(slot, x, dy, want_bias=not args.no_bias)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            gemm_bwd_weight(slot, x, dy, want_bias=not args.no_bias)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        tot += us
        print(f"{M}x{N}x{K}", json.dumps(dict(us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1))), flush=True)
    print("sum_us", round(tot, 1))


if __name__ == "__main__":
    main()
