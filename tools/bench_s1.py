"""bench.py --workload s1: the s1 AR text->semantic GPT micro-step (forward_old + backward, ScaledAdam every 4th
micro-batch as in the reference) at BASELINE configs[2]: batch 32, x_len 256 + y_len 768 = 1024, bf16.
metric: tokens/sec = N * B * 1024 / micro-step time."""
import os
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, world, rank, local):
    from easevoice_trainer_amd.train.s1_engine import S1Engine

    dev = torch.device("cuda", local)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(cfg["train"]["seed"])
    reducer = None
    if world > 1:
        from easevoice_trainer_amd.dist import GradReducer

        reducer = GradReducer(world)
    eng = S1Engine(cfg, dev, dtype, reducer=reducer)
    if world > 1:
        reducer.broadcast_params(eng.arena.param)
    B = 32 if args.batch == 16 else args.batch   # bench.py's default --batch is the s2 one
    x_len, y_len = 256, 768
    g = torch.Generator().manual_seed(1234 + rank)
    batch = dict(phoneme_ids=torch.randint(0, 732, (B, x_len), generator=g).to(dev),
                 phoneme_ids_len=torch.full((B,), x_len, dtype=torch.long, device=dev),
                 semantic_ids=torch.randint(0, 1024, (B, y_len), generator=g).to(dev),
                 semantic_ids_len=torch.full((B,), y_len, dtype=torch.long, device=dev),
                 bert_feature=torch.randn(B, 1024, x_len, generator=g).to(dev))
    idx = 0
    for _ in range(args.warmup):
        loss, acc, _ = eng.micro_step(batch, idx)
        idx += 1
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, acc, _ = eng.micro_step(batch, idx)
        idx += 1
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    tok = world * B * (x_len + y_len)
    return {
        "metric": "tokens/sec (s1)", "value": tok / (dt / args.steps), "unit": "tokens/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"s1 AR text->semantic GPT micro-step (forward_old + backward, ScaledAdam every 4th "
                               f"micro-batch), batch={B}/GPU, x_len=256 + y_len=768, configs/gpt.yaml, attention dropout 0.1",
                   "global_batch": world * B, "seq_len": x_len + y_len, "parallelism": f"dp{world}"},
        "loss_per_token_last": float(loss) / (B * y_len), "top3_acc_last": float(acc),
    }
