"""What the vendor library does at the s1 layer shapes (torch.mm -> hipBLASLt / rocBLAS, bf16): a yardstick for
csrc/gemm256.hip (NT: x . W^T) and wgrad_gemm (TN: dy^T . x), nothing the product calls."""
import json

import torch

SHAPES = [(32768, 1536, 512), (32768, 512, 512), (32768, 2048, 512), (32768, 512, 2048)]


def t(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = torch.device("cuda:0")
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = torch.randn(N, K, device=dev).bfloat16()
        dy = torch.randn(M, N, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        dw = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
        dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        r = {}
        for name, fn in (("fwd x.W^T", lambda: torch.mm(x, w.t(), out=out)), ("bwd_data dy.W", lambda: torch.mm(dy, w, out=dx)),
                         ("wgrad dy^T.x", lambda: torch.mm(dy.t(), x, out=dw))):
            us = t(fn)
            r[name] = dict(us=round(us, 1), tflops=round(fl / us / 1e6, 1))
        print(f"{M}x{N}x{K}", json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
