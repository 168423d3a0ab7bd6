"""bench.py — the s2 SoVITS GAN training step (generator + discriminators, both AdamW updates) on synthetic
fixed-shape batches, one process per GPU.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run with one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.
Workload (BASELINE.json configs[1]): batch 16 per GPU, 4 s clips (T = 200 frames, 128000
samples, text 60), bf16 compute, random-init weights of configs/s2.json, synthetic data.  Weak scaling: the per-GPU
batch is fixed, gradients are all-reduced over RCCL.
metric: audio-seconds/sec trained (s2) = N * B * clip_seconds / step_time (`value`); the s1 half of the combined
BASELINE metric (tokens/sec of the AR GPT micro-step at batch 32 x 1024, configs[2]) is measured in the same run and
reported as the `s1` sub-object with its own roofline (attention kernel vs the MFMA peak) and CPU baseline.
Started bare with --gpus N > 1 (no WORLD_SIZE) it launches its own N ranks.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime loads: easevoice_trainer_amd/__init__.py

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def init_dist(n_gpus):
    """one rank per GPU over RCCL.  EVT_BENCH_BACKEND=gloo is the dry-run switch of the test tiers (a box with fewer
    GPUs than ranks: the ranks share device LOCAL_RANK % device_count, which RCCL refuses; no GPU at all: the s1 leg on
    the CPU with emulated launches, tests/bench_dryrun_worker.py) -- same spawn / rendezvous / broadcast / step / JSON
    path, different transport."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("EVT_BENCH_BACKEND", "nccl")
    if torch.cuda.is_available():
        if backend != "nccl":
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    if world > 1 or os.environ.get("EVT_DP_FORCE", "0") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return world, rank, local


def synth_s2_batch(B, T, t_text, device, seed):
    g = torch.Generator().manual_seed(seed)
    wav = (torch.rand(B, 1, T * 640, generator=g) - 0.5).to(device)
    ssl = torch.randn(B, 768, T, generator=g).to(device)
    text = torch.randint(0, 732, (B, t_text), generator=g).to(device)
    lengths = torch.full((B,), T, dtype=torch.long, device=device)
    tl = torch.full((B,), t_text, dtype=torch.long, device=device)
    return wav, ssl, text, lengths, tl


def _lanes_note(eng):
    """which independent sub-models of the step run on branch streams (hip/disc.py switches)"""
    from easevoice_trainer_amd.hip import disc as HD

    on = []
    if HD.MPD_STREAMS > 1:
        on.append(f"sub-discriminators on {HD.MPD_STREAMS} streams")
    if HD.ENC_STREAM and not eng.net_g.split_backward:
        on.append("prior encoder + quantizer + mel spectrograms on a lane")
    if HD.DEC_STREAM:
        on.append("two of three wide-stage ResBlocks on a lane")
    return "; ".join(on) if on else "one stream"


def run_s2(args, world, rank, local):
    from easevoice_trainer_amd.module.mel_processing import spectrogram_torch
    from easevoice_trainer_amd.train.s2_engine import S2Engine

    dev = torch.device("cuda", local)
    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    torch.manual_seed(hps["train"]["seed"])
    reducer = None
    if world > 1 or os.environ.get("EVT_DP_FORCE", "0") == "1":
        from easevoice_trainer_amd.dist import GradReducer

        reducer = GradReducer(world)
    eng = S2Engine(hps, dev, dtype, reducer=reducer)
    # frozen quantizer codebook: synthetic N(0,1) codes; marks it initialised
    cb = eng.net_g.quantizer.vq.layers[0]._codebook
    cb.embed.normal_(generator=None)
    cb.inited.fill_(1.0)
    if reducer is not None:
        reducer.broadcast_params(eng.rt_g.arena.param)
        reducer.broadcast_params(eng.rt_d.arena.param)
    eng.build_optimizers()
    use_graphs = bool(getattr(args, "graphs", 0))
    if use_graphs:
        # fixed-shape batches: after two eager steps the step is captured once and replayed as HIP graphs (three on one GPU;
        # nine smaller ones with the collectives between them when data-parallel)
        eng.enable_graphs(warmup_steps=2)
    B, T, t_text = args.batch, args.clip_seconds * 50, 60
    wav, ssl, text, lengths, tl = synth_s2_batch(B, T, t_text, dev, 1234 + rank)
    spec = spectrogram_torch(wav.squeeze(1), 2048, 32000, 640, 2048)

    def step():
        return eng.step(ssl, spec, lengths, wav, text, tl)

    for _ in range(max(args.warmup, 4 if use_graphs else 0)):   # graph mode: 2 eager + capture + 1 replay first
        out = step()
    if world > 1:
        torch.distributed.barrier()
    if reducer is not None:
        reducer.reset_stats()
        reducer.timing = dev.type == "cuda"        # HIP events around every wait for the side stream: the exposed part
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    comm = None
    if reducer is not None:
        comm = reducer.comm_report(args.steps)      # this rank's view; the line is rank 0's
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    losses = dict(disc=float(out.disc), gen=float(out.gen), fm=float(out.fm), mel=float(out.mel), kl=float(out.kl))
    finite = all(v == v and abs(v) != float("inf") for v in losses.values())
    res = {
        "metric": "audio-seconds/sec trained (s2)", "value": world * B * args.clip_seconds / (dt / args.steps),
        "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"s2 SoVITS generator+discriminator GAN step, batch={B}/GPU, {args.clip_seconds} s 32 kHz "
                               f"clips (T={T} frames), configs/s2.json, random-init weights",
                   "global_batch": world * B, "parallelism": (reducer.describe(eng.exchange_ranges()) if reducer is not None else "dp1")
                                  + (", cut (data-parallel) program without collectives" if eng.cut_only else ""),
                   "launch": (f"hip-graph replay ({len(eng._program())} graphs/step"
                              f"{', gradient reductions between them' if reducer is not None else ''})") if eng.graphs_enabled else "eager",
                   "streams": _lanes_note(eng)},
        "generated_seconds_per_sec": world * B * 0.64 / (dt / args.steps),
        "losses_last_step": losses, "losses_finite": finite,
    }
    if eng.scaler.enabled:
        res["grad_scaler"] = dict(eng.scaler.state_dict(), optimizer_steps_applied=[eng.optim_d.sync_step_count(),
                                                                                   eng.optim_g.sync_step_count()])
    if comm is not None:
        # gradient exchange per step on rank 0: collectives issued, MiB moved, and how long the compute stream stood
        # waiting for them (what the overlap with the backward did not hide)
        res["comm"] = comm
    return res, eng, step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="both", choices=["both", "s2", "s1"],
                    help="both (default): the s2 line with the s1 leg as its `s1` sub-object")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"],
                    help="f16: the reference's fp16_run mode (IEEE-half build of the library + device-side GradScaler), s2 leg")
    ap.add_argument("--batch", type=int, default=16, help="s2 batch per GPU (BASELINE config 2: 16)")
    ap.add_argument("--s1-batch", type=int, default=32, help="s1 batch per GPU (BASELINE config 3: 32)")
    ap.add_argument("--clip-seconds", type=int, default=4)
    ap.add_argument("--graphs", type=int, default=1, help="1: replay the s2 step as HIP graphs (default), 0: eager launches")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline / cpu_baseline legs (use under rocprofv3)")
    ap.add_argument("--dp-program", type=int, default=0, choices=[0, 1, 2],
                    help="one GPU only. 1: run the data-parallel (cut, nine-graph) program without collectives -- the cost of "
                         "the decomposition itself next to the default three-phase program; 2: the same with the collectives "
                         "issued on a one-rank RCCL group (side stream, between the graph replays)")
    args = ap.parse_args()
    if args.dp_program and args.gpus == 1:
        os.environ["EVT_DP_CUT" if args.dp_program == 1 else "EVT_DP_FORCE"] = "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare (`python bench.py --gpus N`): become the launcher, one rank per GPU over RCCL on 127.0.0.1
        from easevoice_trainer_amd.dist import spawn_ranks

        codes = spawn_ranks([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], list(range(args.gpus)))
        sys.exit(max(int(c or 0) != 0 for c in codes))
    world, rank, local = init_dist(args.gpus)
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} is running with WORLD_SIZE={world}: launch one rank per GPU "
                         f"(torch.distributed.run --nproc-per-node {args.gpus}) or start it bare so that it spawns them")
    res = None
    if args.workload in ("both", "s2"):
        res, eng, step_fn = run_s2(args, world, rank, local)
        if world > 1:
            # the roofline / cpu_baseline legs are single-GPU properties and run extra, individually timed steps; with
            # several ranks a leg that fails on ONE rank would leave the others waiting in a collective -- they are
            # reported by the N = 1 run only
            res["roofline_note"] = "kernel roofline and cpu_baseline are reported by the N = 1 run"
        elif not args.no_extras:
            try:
                from tools import bench_extras

                res.update(bench_extras.s2_extras(args, eng, world, rank, step_fn))
            except Exception as e:  # the headline number must still be printed
                res["extras_error"] = repr(e)
        del eng, step_fn
        torch.cuda.empty_cache()
    if args.workload in ("both", "s1"):
        from tools import bench_s1

        try:
            s1 = bench_s1.run(args, world, rank, local, extras=not args.no_extras and world == 1)
        except Exception as e:
            if res is None:
                raise
            s1 = {"error": repr(e)}
        if res is None:
            res = s1
        else:
            res["metric"] = "audio-seconds/sec trained (s2) [value]; tokens/sec (s1) in `s1`"
            res["s1"] = s1
    if rank == 0:
        # RCCL prints its version banner through C stdio, which is block-buffered on a pipe and would otherwise come out at
        # exit, AFTER the line below: flush it first, so that the JSON line is the last line of stdout
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    sys.stdout.flush()
    os._exit(0) if os.environ.get("EVT_BENCH_HARD_EXIT") == "1" else None


if __name__ == "__main__":
    main()
