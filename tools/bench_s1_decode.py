"""Decoding throughput of the s1 model: tokens/s of infer_panel_naive for one sequence, HIP-graph
replay vs eager launches, with the algorithmic HBM bytes per token (every block matrix once + the key/value cache read).

    python tools/bench_s1_decode.py [--tokens 512] [--x-len 96] [--prompt 128] [--dtype bf16]
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime loads: easevoice_trainer_amd/__init__.py

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=512)
    ap.add_argument("--x-len", type=int, default=96)
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rows", type=int, default=0, help="also time infer_panel_batch_infer with this many texts (<= 4)")
    args = ap.parse_args()
    from easevoice_trainer_amd.train.s1_engine import S1Engine

    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    torch.manual_seed(1234)
    m = S1Engine(cfg, dev, dtype).model
    m.eval()
    with torch.no_grad():
        m.ar_predict_layer.weight[-1].zero_()      # EOS logit 0: never the arg-max of 1025 random logits
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 732, (1, args.x_len), generator=g).to(dev)
    bert = torch.randn(1, 1024, args.x_len, generator=g).to(dev)
    prompts = torch.randint(0, 1024, (1, args.prompt), generator=g).to(dev)
    # a noise table that never lets EOS win keeps every run at exactly --tokens steps
    noise = torch.empty(args.tokens + 2, 1025).exponential_(1, generator=g)
    noise[:, 1024] = 1e30
    noise = noise.to(dev)
    esz = 2 if dtype == torch.bfloat16 else 4
    E, nl = 512, 24
    w_bytes = (nl * (3 * E * E + E * E + 8 * E * E) + 1025 * E) * esz
    out = {}
    for mode in ("1", "0"):
        os.environ["EVT_DECODE_GRAPH"] = mode
        times = []
        for rep in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y, idx = m.infer_panel_naive(x, None, prompts, bert, top_k=15, top_p=1, early_stop_num=args.tokens, noise=noise,
                                         repetition_penalty=1.35)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            assert y.size(1) == args.prompt + args.tokens, (y.shape, idx)
        t = sorted(times[1:])[len(times[1:]) // 2]
        out["graph" if mode == "1" else "eager"] = dict(seconds=round(t, 4), tokens_per_s=round((args.tokens + 1) / t, 1),
                                                        us_per_token=round(1e6 * t / (args.tokens + 1), 1))
    if args.rows:
        os.environ["EVT_DECODE_GRAPH"] = "1"
        R = args.rows
        xs = [x[0][: args.x_len - 7 * r].contiguous() for r in range(R)]         # different text lengths: padded batch
        berts = [bert[0][:, : args.x_len - 7 * r].contiguous() for r in range(R)]
        times = []
        for rep in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ys, idxs = m.infer_panel_batch_infer(xs, None, prompts.expand(R, -1).contiguous(), berts, top_k=15, top_p=1,
                                                 early_stop_num=args.tokens, noise=noise, repetition_penalty=1.35)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            assert all(y.numel() == args.prompt + args.tokens for y in ys), [y.shape for y in ys]
        t = sorted(times[1:])[len(times[1:]) // 2]
        out[f"batch{R}_graph"] = dict(seconds=round(t, 4), tokens_per_s=round(R * (args.tokens + 1) / t, 1),
                                      us_per_step=round(1e6 * t / (args.tokens + 1), 1))
    L_avg = args.x_len + args.prompt + args.tokens / 2
    cache_bytes = nl * 2 * L_avg * E * esz
    per_tok = w_bytes + cache_bytes
    gps = per_tok / (out["graph"]["us_per_token"] * 1e-6) / 1e9
    print(json.dumps(dict(workload=f"s1 decode, 1 sequence, x_len={args.x_len}, prompt={args.prompt}, {args.tokens} tokens, "
                                   f"{args.dtype}, top_k=15 (prompt pass included)",
                          **out, algorithmic_bytes_per_token=int(per_tok), achieved_GBps=round(gps, 1),
                          hbm_peak_GBps=8000, frac=round(gps / 8000, 4))))


if __name__ == "__main__":
    main()
