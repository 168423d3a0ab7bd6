"""Which lines of the product launch torch-native / vendor kernels in one s2 step: torch.profiler with Python stacks,
device time of every aten op grouped by the innermost frames inside easevoice_trainer_amd/.  Development tool.

    python tools/trace_glue.py [--top 60]
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--clip-seconds", type=int, default=4)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    args.graphs = 0
    import bench

    world, rank, local = bench.init_dist(1)
    res, eng, step = bench.run_s2(args, world, rank, local)
    n = 2
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                                with_stack=True) as prof:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    out = os.path.join(ROOT, "gpurun_out", "glue_stacks.txt")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    prof.export_stacks(out, metric="self_cuda_time_total")
    agg = collections.defaultdict(float)
    for line in open(out):
        line = line.rstrip()
        if not line:
            continue
        stack, val = line.rsplit(" ", 1)
        frames = stack.split(";")
        leaf = frames[-1]
        mine = [f for f in frames if "easevoice_trainer_amd" in f]
        where = " <- ".join(f.split("easevoice_trainer_amd/")[-1] for f in mine[-3:][::-1]) or "(no product frame)"
        agg[(leaf[-40:], where)] += float(val)
    tot = sum(agg.values()) / n
    print(f"step {res['ms_per_step']:.1f} ms; stacks with device time: {tot / 1e3:.2f} ms/step")
    for (leaf, where), v in sorted(agg.items(), key=lambda kv: -kv[1])[: args.top]:
        print(f"{v / n:8.1f} us/step  {leaf:<40s} {where}")


if __name__ == "__main__":
    main()
