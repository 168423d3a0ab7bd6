"""CPU development tool: which product lines issue torch (aten) ops in one s2 generator forward + D forward -- the launches
that are NOT HIP kernels of the library.  Runs the modules under tests/cpu_emu.py (HIP entry points replaced by oracle
ops; ops issued from inside those stand-ins are not counted) with a TorchDispatchMode that attributes every aten op to the
innermost frame under easevoice_trainer_amd/.  (torch.profiler's with_stack gives no Python frames on this ROCm build.)

    python tools/count_aten_ops.py
"""
import collections
import json
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SKIP = {"aten::view", "aten::_unsafe_view", "aten::transpose", "aten::t", "aten::unsqueeze", "aten::squeeze", "aten::expand",
        "aten::slice", "aten::select", "aten::detach", "aten::alias", "aten::permute", "aten::as_strided", "aten::split",
        "aten::split_with_sizes", "aten::unbind", "aten::reshape", "aten::empty", "aten::empty_like", "aten::empty_strided",
        "aten::narrow", "aten::_reshape_alias", "aten::unfold", "aten::lift_fresh", "aten::is_same_size", "aten::size",
        "aten::stride", "aten::sym_size", "aten::view_as", "aten::chunk"}


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.name().split(".")[0]
        if name not in SKIP:
            st = traceback.extract_stack()
            where, emu = None, False
            for fr in reversed(st):
                fn = fr.filename
                if "cpu_emu" in fn or "/oracle/" in fn:
                    emu = True
                    break
                if "easevoice_trainer_amd" in fn:
                    where = f"{fn.split('easevoice_trainer_amd/')[-1]}:{fr.lineno} {fr.line.strip()[:90]}"
                    break
            if not emu:
                self.agg[(where or "(autograd engine / backward)", name)] += 1
        return func(*args, **(kwargs or {}))


def main():
    from cpu_emu import cpu_emulation
    from util_fill import fill_module, s2_batch
    from easevoice_trainer_amd.module import losses as PL, mel_processing as PM, models

    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    torch.set_num_threads(8)
    with cpu_emulation():
        net_g = models.SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
        net_d = models.MultiPeriodDiscriminator(False)
        fill_module(net_g, 1)
        fill_module(net_d, 2)
        for m in net_g.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        b = s2_batch(2, 100, 40)
        spec = PM.spectrogram_torch(b["wav"].squeeze(1), 2048, 32000, 640, 2048)
        c = Counter()
        with c:
            out = net_g(b["ssl"], spec, b["lengths"], b["text"], b["text_lengths"], eps=b["eps"], ids_slice=b["ids_slice"])
            y_hat = out[0]
            (y_hat.float().pow(2).mean() + sum(t.float().pow(2).mean() for t in out[5])).backward()
    rows = collections.defaultdict(collections.Counter)
    for (where, name), n in c.agg.items():
        rows[where][name] += n
    tot = sum(sum(v.values()) for v in rows.values())
    print(f"{tot} aten launches (views excluded) in one generator forward + backward")
    for where, v in sorted(rows.items(), key=lambda kv: -sum(kv[1].values())):
        print(f"{sum(v.values()):5d}  {where}\n         " + ", ".join(f"{k.replace('aten::', '')} x{n}" for k, n in v.most_common(8)))


if __name__ == "__main__":
    main()
