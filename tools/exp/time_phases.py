"""untraced per-phase times of the graph-replayed s2 step: HIP events around each of the step's graph replays"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import argparse
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--clip-seconds", type=int, default=4)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=4)
args, _ = ap.parse_known_args()
for k, v in dict(workload="s2", graphs=1, no_extras=True, dp_program=0, gpus=1, s1_batch=32).items():
    setattr(args, k, v)
world, rank, local = bench.init_dist(1)
res, eng, step = bench.run_s2(args, world, rank, local)
ent = next(e for e in eng._graph_cache.values() if e["graphs"] is not None)
n = len(ent["graphs"])
tot = [0.0] * n
reps = 20
for _ in range(reps):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i, (g, after) in enumerate(zip(ent["graphs"], ent["after"])):
        g.replay()
        if after is not None:
            after()
        evs[i + 1].record()
    torch.cuda.synchronize()
    for i in range(n):
        tot[i] += evs[i].elapsed_time(evs[i + 1])
print("phases ms:", [round(t / reps, 3) for t in tot], "sum", round(sum(tot) / reps, 3))
