"""Times one ResBlock step at the B=16 vocoder shapes: forward fused (csrc/resunit.hip) vs the three launches it replaces,
backward fused (csrc/resunit_bwd.hip: data + weight + bias gradients) vs the four launches it replaces.
usage: python tools/bench_resunit.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easevoice_trainer_amd.hip import conv as HC   # noqa: E402
from easevoice_trainer_amd.hip import lib as L    # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    wide_fwd = "--wide-fwd" in sys.argv       # only the fused forward of the wide shapes (tile-shape experiments)
    shapes = ((64, 5120), (128, 2560)) if wide_fwd else ((16, 20480), (32, 10240), (64, 5120), (128, 2560))
    for C_, Lq in shapes:
        for k, d in (((3, 1), (7, 3), (11, 5)) if wide_fwd else
                     ((3, 1), (3, 5), (7, 1), (7, 3), (7, 5), (11, 1), (11, 3), (11, 5))):
            pad = lambda kk, dd: (kk * dd - dd) // 2
            m = torch.nn.ModuleList([HC.EvtConv1d(C_, C_, k, dilation=d, padding=pad(k, d), weight_norm=True),
                                     HC.EvtConv1d(C_, C_, k, dilation=1, padding=pad(k, 1), weight_norm=True)]).to(dev)
            bank = HC.WeightBank(m, torch.bfloat16, dev)
            bank.build_tables()
            bank.fold()
            x = torch.randn(16, Lq, C_, device=dev).bfloat16()
            s1, s2 = m[0]._slot, m[1]._slot

            def unfused():
                xa = HC._lrelu(x, 0.1)
                mid = HC._fwd(s1, xa, None, 1.0, L.ACT_LRELU, 0.1)
                return HC._fwd(s2, mid, x, 1.0, L.ACT_NONE, 1.0)

            def fused():
                with torch.no_grad():
                    return HC.res_unit(x, m[0], m[1], 0.1)

            dy = torch.randn(16, Lq, C_, device=dev).bfloat16()
            xa = HC._lrelu(x, 0.1)
            mid_a = HC._fwd(s1, xa, None, 1.0, L.ACT_LRELU, 0.1)
            bank.defer_n = 0

            def bwd_unfused():
                HC._bwd_weight(s2, mid_a, dy, None, 16, Lq, 1.0, L.ACT_NONE, 1.0)
                dmid = HC._bwd_data(s2, dy, None, mid_a, None, 16, Lq, 0.1, L.ACT_NONE, 1.0)
                HC._bwd_weight(s1, xa, dmid, None, 16, Lq, 1.0, L.ACT_NONE, 1.0)
                return HC._bwd_data(s1, dmid, None, xa, dy, 16, Lq, 0.1, L.ACT_NONE, 1.0)

            def bwd_fused():
                if C_ <= 32:
                    return HC.resunit_bwd(s1, s2, dy, xa, mid_a, 0.1, 1.0)
                xg = x.clone().requires_grad_(True)
                y = HC.res_unit(xg, m[0], m[1], 0.1)       # fwd + bwd through the autograd node (wide path)
                y.backward(dy)
                return xg.grad

            out = dict(unfused=0.0, bwd_unfused=0.0, bwd_fused=0.0)
            for name, fn in ((("fused", fused),) if wide_fwd else
                             (("unfused", unfused), ("fused", fused), ("bwd_unfused", bwd_unfused), ("bwd_fused", bwd_fused))):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                out[name] = e0.elapsed_time(e1) / 20 * 1e3
            print(f"C={C_:3d} L={Lq:6d} k={k:2d} d={d}: fwd unfused {out['unfused']:6.1f} us  fused {out['fused']:6.1f} us | "
                  f"bwd unfused {out['bwd_unfused']:6.1f} us  fused {out['bwd_fused']:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
