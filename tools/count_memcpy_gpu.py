"""GPU development tool: which product lines issue same-dtype contiguous copies (hipMemcpyAsync: `__amd_rocclr_copyBuffer`
in the kernel stats), fills and other aten element-wise ops in one eager s2 step -- TorchDispatchMode with the innermost
frame under easevoice_trainer_amd/ (ops issued by the autograd engine itself carry no Python frame)."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name().split(".")[0].replace("aten::", "")
        kind = None
        if name in ("copy_", "clone", "contiguous", "_to_copy"):
            src = args[1] if name == "copy_" else args[0]
            dst_dtype = (args[0].dtype if name == "copy_" else (kwargs or {}).get("dtype", src.dtype)) or src.dtype
            if isinstance(src, torch.Tensor) and src.is_cuda and src.dtype == dst_dtype and src.is_contiguous():
                kind = f"memcpy:{name}"
            elif isinstance(src, torch.Tensor) and src.is_cuda:
                kind = f"cast/strided:{name}"
            elif isinstance(src, torch.Tensor) and not src.is_cuda and name == "copy_" and args[0].is_cuda:
                kind = "memcpy:h2d"
        elif name in ("fill_", "zero_", "zeros", "zeros_like", "add", "add_", "mul", "sum", "cat", "stack", "index_select"):
            kind = name
        if kind is not None:
            where = None
            for fr in reversed(traceback.extract_stack()):
                if "easevoice_trainer_amd" in fr.filename:
                    where = f"{fr.filename.split('easevoice_trainer_amd/')[-1]}:{fr.lineno} {(fr.line or '').strip()[:70]}"
                    break
            self.agg[(kind, where or "(autograd engine)")] += 1
        return func(*args, **(kwargs or {}))


def main():
    import argparse

    import bench
    args = argparse.Namespace(batch=16, clip_seconds=4, dtype="bf16", steps=1, warmup=2, graphs=0)
    world, rank, local = bench.init_dist(1)
    res, eng, step = bench.run_s2(args, world, rank, local)
    c = Counter()
    with c:
        step()
    torch.cuda.synchronize()
    by_kind = collections.Counter()
    for (kind, where), n in c.agg.items():
        by_kind[kind] += n
    print(dict(by_kind))
    for (kind, where), n in sorted(c.agg.items(), key=lambda kv: -kv[1])[:70]:
        print(f"{n:4d}  {kind:<22s} {where}")


if __name__ == "__main__":
    main()
