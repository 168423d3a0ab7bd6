"""linear1 forward of the s1 blocks ([32768, 2048, 512], bias + relu + dropout epilogue): us per launch.
   EVT_GEMM256_PIPE=0 python tools/exp/time_gemm_drop.py   (round-3 kernel)   /   default: the pipelined kernel"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from easevoice_trainer_amd.hip import enc as E
from easevoice_trainer_amd.hip.linear import LinearBank, gemm_fwd
dev = torch.device("cuda:0")
E.seed_rng(dev, 1)
for M, N, K in [(32768, 2048, 512), (32768, 1536, 512)]:
    w = torch.nn.Parameter(torch.randn(N, K, device=dev) * K ** -0.5); b = torch.nn.Parameter(torch.randn(N, device=dev) * 0.1)
    bank = LinearBank([("t", w, b)], torch.bfloat16, dev); bank.prepare(); slot = w._evt_slot
    x = torch.randn(M, K, device=dev).bfloat16()
    for name, kw in (("bias", {}), ("bias+relu", dict(relu=True)), ("bias+relu+dropout", dict(relu=True, drop=(0.1, 7)))):
        for _ in range(3): gemm_fwd(slot, x, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): gemm_fwd(slot, x, **kw)
        e1.record(); torch.cuda.synchronize()
        print(f"{M}x{N}x{K} {name:20s} {e0.elapsed_time(e1) * 1e3 / 30:7.1f} us", flush=True)
