"""Times the s1 attention kernels alone (evt_attn_prefixlm_fwd / _bwd through the C ABI) at the BASELINE config-3 shape
(B x 16 heads x (256 + 768) x 32, dropout 0.1) for both kernel variants (evt_debug_attn_variant).  Kernel durations come
from torch.profiler's kernel records (the tracer rocprofv3 uses), wall times from HIP events.
usage: python tools/bench_attn.py [--batch 32] [--iters 10]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easevoice_trainer_amd.hip import lib as L                      # noqa: E402
from easevoice_trainer_amd.auto_reg.ops import PrefixLMAttentionFn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dropout", type=float, default=0.1)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, H, D, xl, yl = args.batch, 16, 32, 256, 768
    Lq, E = xl + yl, 16 * 32
    torch.manual_seed(0)
    qkv = (torch.randn(B, Lq, 3 * E, device=dev) * 0.8).bfloat16().requires_grad_(True)
    d_o = torch.randn(B, Lq, E, device=dev).bfloat16()
    x_lens = torch.full((B,), xl, dtype=torch.int32, device=dev)
    y_lens = torch.full((B,), yl, dtype=torch.int32, device=dev)
    flops_fwd = 4.0 * B * H * Lq * Lq * D          # no causal skipping credited
    out = {}
    for joint in (1, 0):
        L.lib().evt_debug_attn_variant(joint)
        for _ in range(3):
            o = PrefixLMAttentionFn.apply(qkv, x_lens, y_lens, xl, H, args.dropout, 7)
            o.backward(d_o)
        torch.cuda.synchronize()
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
            for _ in range(args.iters):
                o = PrefixLMAttentionFn.apply(qkv, x_lens, y_lens, xl, H, args.dropout, 7)
                o.backward(d_o)
            torch.cuda.synchronize()
        rec = {}
        for ev in prof.key_averages():
            if "attn" in ev.key:
                import re
                name = re.search(r"attn_\w+", ev.key).group(0)
                rec[name] = round(ev.device_time_total / max(ev.count, 1), 1)
        fwd_us = next((v for k, v in rec.items() if "fwd" in k), None)
        out["joint" if joint else "split"] = {
            "kernel_us": rec,
            "fwd_tflops_uncredited": round(flops_fwd / (fwd_us * 1e-6) / 1e12, 1) if fwd_us else None,
            "sum_us": round(sum(rec.values()), 1),
        }
    print(json.dumps({"shape": f"B={B} H={H} L={Lq} D={D} dropout={args.dropout}", **out}))


if __name__ == "__main__":
    main()
