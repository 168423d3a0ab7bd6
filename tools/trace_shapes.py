"""Per-(kernel, shape) time table of one s2 training step: the conv entry points are bracketed with HIP events
(hip/conv.py TRACE) and grouped by the launched kernel instantiation and the layer geometry.  Development tool —
shows which layers a kernel's aggregate time in `rocprofv3 --stats` comes from.

    python tools/trace_shapes.py [--top 40]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--clip-seconds", type=int, default=4)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import bench
    from easevoice_trainer_amd.hip import conv as HC

    world, rank, local = bench.init_dist(1)
    res, eng, step = bench.run_s2(args, world, rank, local)
    HC.set_trace([])
    n = 2
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    rec = HC.TRACE
    HC.set_trace(None)
    agg = {}
    for tag, kind, flops, nbytes, e0, e1, shape, _m in rec:
        a = agg.setdefault((tag, kind, shape), [0.0, 0, 0.0, 0.0])
        a[0] += e0.elapsed_time(e1)
        a[1] += 1
        a[2] += flops
        a[3] += nbytes
    tot = sum(a[0] for a in agg.values()) / n
    print(f"step {res['ms_per_step']:.1f} ms; conv entry points {tot:.1f} ms/step over {len(rec) // n} launches")
    print(f"{'ms/step':>8} {'calls':>5} {'us/call':>8} {'TF':>7} {'GB/s':>7}  kind        kernel | shape")
    for (tag, kind, shape), a in sorted(agg.items(), key=lambda kv: -kv[1][0])[: args.top]:
        sec = a[0] / 1e3
        print(f"{a[0] / n:8.3f} {a[1] // n:5d} {a[0] * 1e3 / a[1]:8.1f} {a[2] / sec / 1e12:7.1f} {a[3] / sec / 1e9:7.0f}  "
              f"{kind:<11} {tag} | {shape}")


if __name__ == "__main__":
    main()
