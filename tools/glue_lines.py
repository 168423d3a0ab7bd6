"""GPU development tool: which product lines are behind the torch (aten) kernels of one eager s2 step -- forward AND backward.
A TorchDispatchMode counts every non-view aten op; forward ops are attributed to the innermost frame under
easevoice_trainer_amd/, backward ops to the forward line that created the autograd node (anomaly mode keeps that
traceback in node.metadata), custom Function backward code to its own frame.

    python tools/glue_lines.py [--top 70]
"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SKIP = {"aten::view", "aten::_unsafe_view", "aten::transpose", "aten::t", "aten::unsqueeze", "aten::squeeze", "aten::expand",
        "aten::slice", "aten::select", "aten::detach", "aten::alias", "aten::permute", "aten::as_strided", "aten::split",
        "aten::split_with_sizes", "aten::unbind", "aten::reshape", "aten::empty", "aten::empty_like", "aten::empty_strided",
        "aten::narrow", "aten::_reshape_alias", "aten::unfold", "aten::lift_fresh", "aten::is_same_size", "aten::size",
        "aten::stride", "aten::sym_size", "aten::view_as", "aten::chunk", "aten::record_stream", "aten::is_pinned",
        "aten::new_empty", "aten::new_empty_strided", "aten::_local_scalar_dense", "aten::unsafe_split", "aten::unflatten",
        "aten::flatten", "aten::movedim", "aten::swapaxes", "aten::is_nonzero", "aten::_has_compatible_shallow_copy_type",
        "aten::set_", "aten::resize_", "aten::result_type", "aten::sym_numel", "aten::sym_stride", "aten::sym_storage_offset",
        "aten::numel", "aten::dim", "aten::is_contiguous", "aten::unsafe_chunk", "aten::view_as_real", "aten::view_as_complex"}


DEPTH = 1


def _pkg_frame(frames):
    out = []
    for fr in reversed(frames):
        fn = fr.filename if hasattr(fr, "filename") else fr[0]
        if "easevoice_trainer_amd" in fn and "/tools/" not in fn:
            line = fr.line if hasattr(fr, "line") else ""
            out.append(f"{fn.split('easevoice_trainer_amd/')[-1]}:{fr.lineno} {(line or '').strip()[:100]}")
            if len(out) >= DEPTH:
                break
    return "\n          <- ".join(out) if out else None


def _from_text(tb_lines):
    """anomaly mode stores the forward traceback as formatted text lines"""
    found = []
    text = "".join(tb_lines) if isinstance(tb_lines, (list, tuple)) else str(tb_lines)
    cur = None
    for ln in text.splitlines():
        s = ln.strip()
        if s.startswith('File "'):
            cur = s
        elif cur is not None:
            if "easevoice_trainer_amd" in cur and "/tools/" not in cur:
                try:
                    fn = cur.split('"')[1].split("easevoice_trainer_amd/")[-1]
                    no = cur.split("line ")[1].split(",")[0]
                    found.append(f"{fn}:{no} {s[:100]}")
                except Exception:
                    pass
            cur = None
    if not found:
        return None
    return "\n          <- ".join(reversed(found[-DEPTH:]))


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name().split(".")[0]
        if name not in SKIP:
            where = _pkg_frame(traceback.extract_stack())
            phase = "fwd"
            node = torch._C._current_autograd_node()
            if node is not None:
                phase = "bwd"
                inner = where if where and ("backward" in where or "hip/" in where) else None
                tb = node.metadata.get("traceback_") if hasattr(node, "metadata") else None
                src = _from_text(tb) if tb else None
                where = inner or (f"[{node.name()}] " + (src or "?"))
            self.agg[(phase, where or "(no product frame)", name)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--clip-seconds", type=int, default=4)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--depth", type=int, default=1, help="package frames shown per op (innermost first)")
    args = ap.parse_args()
    args.graphs = 0
    global DEPTH
    DEPTH = args.depth
    import bench

    world, rank, local = bench.init_dist(1)
    res, eng, step = bench.run_s2(args, world, rank, local)
    torch.cuda.synchronize()
    c = Counter()
    with torch.autograd.detect_anomaly(check_nan=False):
        with c:
            step()
    torch.cuda.synchronize()
    rows = collections.defaultdict(collections.Counter)
    for (phase, where, name), n in c.agg.items():
        rows[(phase, where)][name] += n
    tot = sum(sum(v.values()) for v in rows.values())
    print(f"{tot} aten ops (views excluded) in one eager s2 step; fwd {sum(sum(v.values()) for k, v in rows.items() if k[0] == 'fwd')}, "
          f"bwd {sum(sum(v.values()) for k, v in rows.items() if k[0] == 'bwd')}")
    for (phase, where), v in sorted(rows.items(), key=lambda kv: -sum(kv[1].values()))[:args.top]:
        print(f"{sum(v.values()):5d} {phase}  {where}\n             " + ", ".join(f"{k.replace('aten::', '')} x{n}" for k, n in v.most_common(8)))


if __name__ == "__main__":
    main()
