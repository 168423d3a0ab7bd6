"""GPU development tool: forward of one WN stack (posterior encoder shape: 16 layers, H = 192, k = 5, B x T = 16 x 200) with the
one-launch layer (csrc/wn_layer.hip) and with the four launches per layer, alternating in one process, each as a replayed HIP graph; us per layer.
--flush: a 512 MB fill between the stacks, so every layer's weights are a first touch as in the training step.
Tile / ring variants: EVT_WN_NT=1|2|3, EVT_WN_RING=4|8 (read once per process).

    python tools/bench_wn.py [--dtype bf16|f16] [--flush] [--iters 30]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--flush", action="store_true")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--backward", action="store_true", help="forward + backward of the stack (weight gradients included)")
    args = ap.parse_args()
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.hip import lib as L
    from easevoice_trainer_amd.hip import wn as W
    from easevoice_trainer_amd.module.models import WN

    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    L.set_half(dtype)
    torch.manual_seed(0)
    H, B, T, NL = 192, args.batch, args.frames, args.layers
    m = WN(H, 5, 1, NL, gin_channels=512).to(dev)
    bank = HC.WeightBank(m, dtype, dev)
    bank.build_tables()
    bank.fold()
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    live = torch.ones(B, T, 1, device=dev, dtype=dtype)
    x = torch.randn(B, T, H, device=dev).to(dtype)
    g = torch.randn(B, 512, device=dev)
    junk = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev) if args.flush else None
    times = {True: [], False: []}
    graphs = {}
    wgt = torch.randn(B, T, H, device=dev).to(dtype)
    bank.defer_n = 0                     # weight gradients in line: one stream, what the captured graph replays

    def run(xin, glb):
        out = W.wn_stack(xin, glb, lens, m.in_layers, m.res_skip_layers, H)
        if args.backward:
            out.backward(wgt)
        return out
    with torch.set_grad_enabled(args.backward):
        g_lbh = m.cond_layer(g).to(dtype).view(B, NL, 2 * H).transpose(0, 1).contiguous().detach()
        if args.backward:
            x.requires_grad_(True)
            g_lbh.requires_grad_(True)
        for fused in (True, False):          # one captured graph per variant: replay has no host time between the launches
            W.FUSED_FORWARD = W.FUSED_BACKWARD = fused
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    run(x, g_lbh)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                out = run(x, g_lbh)
            graphs[fused] = (gr, out.detach())
        W.FUSED_FORWARD = W.FUSED_BACKWARD = True
        for gr, _ in graphs.values():
            gr.replay()
        torch.cuda.synchronize()
        d = (graphs[True][1].float() - graphs[False][1].float()).abs().max().item()
        print(f"max |one launch - four launches| of the stack output: {d:.4g} (output max {graphs[False][1].float().abs().max().item():.4g})")
    with torch.no_grad():
        for it in range(args.iters + 3):
            for fused in (True, False):
                if junk is not None:
                    junk.fill_(float(it))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graphs[fused][0].replay()
                e1.record()
                torch.cuda.synchronize()
                if it >= 3:
                    times[fused].append(e0.elapsed_time(e1) * 1e3 / NL)
    for fused in (True, False):
        v = sorted(times[fused])
        print(f"{'one launch ' if fused else 'four launches'} per layer: median {v[len(v) // 2]:7.2f} us  min {v[0]:7.2f} us   "
              f"[{args.dtype}, {B} x {T}, {NL} layers, flush={args.flush}, EVT_WN_NT={os.environ.get('EVT_WN_NT', '1')}, "
              f"EVT_WN_RING={os.environ.get('EVT_WN_RING', '8')}; graph replay{', forward + backward' if args.backward else ''}]")


if __name__ == "__main__":
    main()
