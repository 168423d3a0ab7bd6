"""The s1 leg of bench.py: the AR text->semantic GPT micro-step (forward_old + backward, ScaledAdam every 4th micro-batch
as in the reference) at BASELINE configs[2]: batch 32, x_len 256 + y_len 768 = 1024, bf16.
metric: tokens/sec = N * B * 1024 / micro-step time.  Roofline: the attention forward kernel against the dense bf16
MFMA peak on the flops it EXECUTES ("attention FLOPs executed / (time x peak)"): the prefix-LM mask lets a text
row see the x text columns only and an audio row the text plus its own past, so of the L^2 score elements of a (batch,
head) x^2 + y*x + y(y+1)/2 are computed (53 % at 256 + 768) -- the kernel skips the other tiles.  `frac_credited` keeps the
full-square accounting (4 * L^2 * D * H * B) for comparison with earlier rounds.  Durations from torch.profiler's kernel
records; north_star's "attention at batch 16" point is timed too; the GEMM kernels get their own TFLOP/s (`gemm`)."""
import os
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MFMA_BF16_PEAK_TF = 2500.0
MFMA_F32_PEAK_TF = 157.3


def _batch(B, x_len, y_len, dev, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(phoneme_ids=torch.randint(0, 732, (B, x_len), generator=g).to(dev),
                phoneme_ids_len=torch.full((B,), x_len, dtype=torch.long, device=dev),
                semantic_ids=torch.randint(0, 1024, (B, y_len), generator=g).to(dev),
                semantic_ids_len=torch.full((B,), y_len, dtype=torch.long, device=dev),
                bert_feature=torch.randn(B, 1024, x_len, generator=g).to(dev))


def _pmc_util():
    """MFMA utilisation of the three attention kernels from the committed PMC pass (profiles/attn_pmc.json, written by
    tools/pmc_summary.py --json from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...` run); None when
    the file is absent"""
    import json
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "attn_pmc.json")))
    except Exception:
        return None


def attention_roofline(eng, batch, B, Lq, dtype_name, n_micro=2, x_len=256):
    """per-kernel records of `n_micro` micro-steps -> roofline object of the attention forward kernel (+ the two
    backward kernels and the GEMM kernels in `also`)"""
    from tools.bench_extras import kernel_profile, short_name

    m = eng.config["model"]
    H, E, nl = m["head"], m["hidden_dim"], m["n_layer"]
    D = E // H

    def run():
        for i in range(n_micro):
            eng.micro_step(batch, 1 + i)      # indices 1..: no optimiser step inside the profiled region

    # per-kernel durations are taken with the weight-gradient side stream OFF: a kernel that shares the chip with another
    # stream's kernel has a longer record without being slower (the timed steps of the bench line keep the side stream)
    side, eng.bank.side = eng.bank.side, None
    try:
        kernels, _ = kernel_profile(run)
    except Exception as e:
        eng.bank.side = side
        return dict(bound="mfma", achieved=None, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s", frac=None, traffic=None,
                    note=f"profiler unavailable: {e!r}")
    eng.bank.side = side

    def pick(sub):
        v = [(k, c) for k, c in kernels.items() if sub in k]
        return (sum(c[0] for _, c in v), sum(c[1] for _, c in v)) if v else (0, 0.0)

    peak = MFMA_BF16_PEAK_TF if dtype_name == "bf16" else MFMA_F32_PEAK_TF
    y_len = Lq - x_len
    scores_exec = x_len * x_len + y_len * x_len + y_len * (y_len + 1) // 2     # per (batch, head): what the mask leaves
    flops_full = 4.0 * Lq * Lq * D * H * B                 # QK^T and PV over the full square, per layer
    flops_fwd = 4.0 * scores_exec * D * H * B              # ... over the computed score elements
    calls, us = pick("attn_fwd")
    total_us = sum(c[1] for c in kernels.values())
    out = dict(bound="mfma", peak=peak, unit="TFLOP/s", traffic=None, kernel="attn_fwd_" + ("bf16" if dtype_name == "bf16" else "f32"),
               timing="torch.profiler kernel records (roctracer, the clock rocprofv3 uses)",
               flops_per_launch=flops_fwd, flops_per_launch_full_square=flops_full, executed_share=flops_fwd / flops_full,
               causal_skip_credited=False, batch=B, seq_len=Lq, head_dim=D, heads=H)
    if calls:
        avg = us / calls
        out.update(achieved=flops_fwd / (avg * 1e-6) / 1e12, frac=flops_fwd / (avg * 1e-6) / 1e12 / peak,
                   frac_credited=flops_full / (avg * 1e-6) / 1e12 / peak, avg_launch_us=avg,
                   launches_per_micro_step=calls / n_micro)
    else:
        out.update(achieved=None, frac=None)
    pmc = _pmc_util()
    if pmc is not None:
        out["mfma_util_pmc"] = pmc
    also = {}
    for name, mult in (("attn_bwd_dq", 2.0), ("attn_bwd_dkv", 2.0)):     # each recomputes S and does two more products
        c, u = pick(name)
        if c:
            also[name] = dict(avg_us=round(u / c, 1), tflops=round(mult * flops_fwd / (u / c * 1e-6) / 1e12, 1))
    att_us = sum(pick(n)[1] for n in ("attn_fwd", "attn_bwd_dq", "attn_bwd_dkv", "attn_delta"))
    # the Linear layers run on the k = 1 members of the conv family (csrc/gemm.hip); anything named Cijk_* would be a
    # vendor GEMM
    is_gemm = lambda k: any(t in short_name(k) for t in ("conv_deep", "conv_ring", "wgrad_gemm", "wgrad_ring", "wgrad_deep",
                                                          "gemm_bf16", "conv_igemm", "rows16_gemm", "gemm256"))
    gemm = sorted(((k, c) for k, c in kernels.items() if k.startswith("Cijk_") or is_gemm(k)), key=lambda kv: -kv[1][1])
    out["also"] = also
    out["attention_ms_per_micro_step"] = round(att_us / 1e3 / n_micro, 3)
    gemm_ms = sum(c[1] for _, c in gemm) / 1e3 / n_micro
    out["gemm_ms_per_micro_step"] = round(gemm_ms, 3)
    # forward MACs of the dense layers: 24 blocks x 12 d^2 per token, the vocabulary projection on the y tokens, bert_proj
    # on the x tokens; forward + backward-data + backward-weight = 3 x 2 flops per MAC
    macs = B * (Lq * nl * 12 * E * E + y_len * 1025 * E + x_len * 1024 * E)
    if gemm_ms > 0:
        tf = 6.0 * macs / (gemm_ms * 1e-3) / 1e12
        out["gemm"] = dict(tflops=round(tf, 1), frac=round(tf / peak, 4), gflop_per_micro_step=round(6.0 * macs / 1e9, 1),
                           kernels={short_name(k)[:40]: round(c[1] / 1e3 / n_micro, 3) for k, c in gemm[:6]})
    out["vendor_gemm_kernels"] = sum(1 for k, _ in gemm if k.startswith("Cijk_"))
    out["gpu_kernel_ms_per_micro_step"] = round(total_us / 1e3 / n_micro, 3)
    out["gpu_kernels_top"] = [dict(kernel=short_name(k)[:60], calls=c[0] // n_micro, avg_us=round(c[1] / c[0], 1),
                                   ms=round(c[1] / 1e3 / n_micro, 3))
                              for k, c in sorted(kernels.items(), key=lambda kv: -kv[1][1])[:8]]
    return out


def run(args, world, rank, local, extras=True):
    from easevoice_trainer_amd.train.s1_engine import S1Engine

    on_gpu = torch.cuda.is_available()
    dev = torch.device("cuda", local) if on_gpu else torch.device("cpu")     # cpu: the emulated dry run of the test tier
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    if os.environ.get("EVT_BENCH_TINY") == "1":                              # dry runs only: a 2-layer toy of the model
        cfg["model"].update(hidden_dim=64, embedding_dim=64, head=4, n_layer=2, linear_units=256)
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16      # --dtype f16 is the s2 leg's fp16_run mode
    dtype_name = "f32" if args.dtype == "f32" else "bf16"
    torch.manual_seed(cfg["train"]["seed"])
    reducer = None
    if world > 1 or os.environ.get("EVT_DP_FORCE", "0") == "1":      # forced: one-rank collectives (bench.py --dp-program 2)
        from easevoice_trainer_amd.dist import GradReducer

        reducer = GradReducer(world)
    eng = S1Engine(cfg, dev, dtype, reducer=reducer)
    if reducer is not None:
        reducer.broadcast_params(eng.arena.param)
    B = args.s1_batch
    x_len, y_len = (256, 768) if os.environ.get("EVT_BENCH_TINY") != "1" else (8, 24)
    batch = _batch(B, x_len, y_len, dev, 1234 + rank)
    idx = 0
    for _ in range(max(args.warmup, 1)):
        loss, acc, _ = eng.micro_step(batch, idx)
        idx += 1
    if world > 1:
        torch.distributed.barrier()
    if reducer is not None:
        reducer.reset_stats()
        reducer.timing = on_gpu
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, acc, _ = eng.micro_step(batch, idx)
        idx += 1
    if world > 1:
        torch.distributed.barrier()
    sync()
    dt = time.perf_counter() - t0
    comm = reducer.comm_report(args.steps) if reducer is not None else None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    tok = world * B * (x_len + y_len)
    res = {
        "metric": "tokens/sec (s1)", "value": tok / (dt / args.steps), "unit": "tokens/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": f"s1 AR text->semantic GPT micro-step (forward_old + backward, ScaledAdam every 4th "
                               f"micro-batch), batch={B}/GPU, x_len=256 + y_len=768, configs/gpt.yaml, attention dropout 0.1",
                   "global_batch": world * B, "seq_len": x_len + y_len,
                   "parallelism": (reducer.describe() + f", {1 + len(eng._cuts)} pieces per optimiser step, the first "
                                   "ones under the last micro-batch's backward") if reducer is not None else "dp1"},
        "loss_per_token_last": float(loss) / (B * y_len), "top3_acc_last": float(acc),
    }
    if comm is not None:
        res["comm"] = comm       # per MICRO-step on rank 0 (one exchange every fourth): collectives, MiB, exposed wait
    if extras:
        try:
            res["roofline"] = attention_roofline(eng, batch, B, x_len + y_len, dtype_name)
            if B != 16:       # north_star quotes MFMA utilisation of the attention "at batch 16"
                b16 = _batch(16, x_len, y_len, dev, 99)
                eng.micro_step(b16, 1)
                r16 = attention_roofline(eng, b16, 16, x_len + y_len, dtype_name)
                res["roofline"]["at_batch_16"] = {k: r16.get(k) for k in ("achieved", "frac", "frac_credited", "avg_launch_us",
                                                                          "also", "attention_ms_per_micro_step")}
        except Exception as e:
            res["roofline_error"] = repr(e)
        if world == 1 and rank == 0:
            from tools.bench_extras import cpu_baseline_s1

            res["cpu_baseline"] = cpu_baseline_s1()
    del eng
    if on_gpu:
        torch.cuda.empty_cache()
    return res
