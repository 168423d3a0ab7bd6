"""BASELINE configs[4]: s1 and s2 fine-tuned side by side on ONE node -- s1 (AR text->semantic GPT) data-parallel on some
GPUs, s2 (SoVITS generator + discriminators) data-parallel on the others -- one RCCL bootstrap, two sub-communicators
(dist.split_subgroups), synthetic batches for a fixed wall-clock budget.

    python tools/launch_joint.py --s1-gpus 0-1 --s2-gpus 2-3-4-5-6-7 --minutes 10

What the reference does for this: nothing in one process group -- `easy_mode` runs the two trainers one after the other
(src/cmd/easy_mode.py:94-129), each with its own DDP world (src/train/sovits.py:319-322, src/train/gpt.py:147-162).  Here
the two jobs share the node: every rank joins one world (the bootstrap), both sub-groups are created by every rank (the
collective contract of new_group), and from then on a rank only ever talks inside its own group: the s1 ranks reduce
their 310 MB arena once per optimiser step, the s2 ranks their D / G arenas every step, on disjoint sets of xGMI links.
A job stops when its group's first rank sees the deadline and says so in a one-int broadcast (every rank of a group runs
the same number of steps -- a rank that stopped alone would leave the others waiting in a collective).  Rank 0 prints ONE
JSON line: tokens/s of the s1 group, audio-s/s of the s2 group, over the common wall time.

Started bare it spawns one process per listed GPU (dist.spawn_ranks); under torchrun it takes the ranks from the
environment.  `make_job` is the seam the CPU test uses to stand in for the engines (tests/joint_worker.py)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class S1Job:
    """S1Engine on fixed-shape synthetic micro-batches (BASELINE config 3 shapes); units = tokens"""

    def __init__(self, dev, reducer, rank_in_group, args):
        import yaml
        from easevoice_trainer_amd.train.s1_engine import S1Engine
        from tools.bench_s1 import _batch

        cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
        torch.manual_seed(cfg["train"]["seed"])
        self.eng = S1Engine(cfg, dev, torch.bfloat16, reducer=reducer)
        self.B, self.x_len, self.y_len = args.s1_batch, 256, 768
        self.batch = _batch(self.B, self.x_len, self.y_len, dev, 1234 + rank_in_group)
        self.idx = 0
        self.units_per_step = self.B * (self.x_len + self.y_len)
        self.unit = "tokens"

    def params(self):
        return [self.eng.arena.param]

    def step(self):
        self.eng.micro_step(self.batch, self.idx)
        self.idx += 1


class S2Job:
    """S2Engine on fixed-shape synthetic batches (BASELINE config 2 shapes); units = audio seconds"""

    def __init__(self, dev, reducer, rank_in_group, args):
        from bench import synth_s2_batch
        from easevoice_trainer_amd.module.mel_processing import spectrogram_torch
        from easevoice_trainer_amd.train.s2_engine import S2Engine

        hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
        torch.manual_seed(hps["train"]["seed"])
        self.eng = S2Engine(hps, dev, torch.bfloat16, reducer=reducer)
        cb = self.eng.net_g.quantizer.vq.layers[0]._codebook
        cb.embed.normal_()
        cb.inited.fill_(1.0)
        B, T = args.s2_batch, args.clip_seconds * 50
        self.wav, self.ssl, self.text, self.lengths, self.tl = synth_s2_batch(B, T, 60, dev, 1234 + rank_in_group)
        self.spec = spectrogram_torch(self.wav.squeeze(1), 2048, 32000, 640, 2048)
        self.units_per_step = B * args.clip_seconds
        self.unit = "audio-s"
        self._built = False

    def params(self):
        return [self.eng.rt_g.arena.param, self.eng.rt_d.arena.param]

    def step(self):
        if not self._built:            # after the parameter broadcast
            self.eng.build_optimizers()
            self.eng.enable_graphs(warmup_steps=2)
            self._built = True
        self.eng.step(self.ssl, self.spec, self.lengths, self.wav, self.text, self.tl)


def make_job(role, dev, reducer, rank_in_group, args):
    return (S1Job if role == "s1" else S2Job)(dev, reducer, rank_in_group, args)


def run_rank(args, make=make_job):
    """one rank of the joint run; returns the dict rank 0 prints (None on the other ranks)"""
    import torch.distributed as dist
    from easevoice_trainer_amd.dist import GradReducer, parse_gpu_ids, split_subgroups

    s1_ids, s2_ids = parse_gpu_ids(args.s1_gpus), parse_gpu_ids(args.s2_gpus)
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    if world != len(s1_ids) + len(s2_ids):
        raise SystemExit(f"launch_joint: WORLD_SIZE={world} but {len(s1_ids)} + {len(s2_ids)} GPUs are listed")
    on_gpu = torch.cuda.is_available() and args.backend == "nccl"
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count() if args.backend != "nccl" else local
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device("cpu")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    ranks_a = list(range(len(s1_ids)))                                  # ranks are handed out in listing order: s1 first
    ranks_b = list(range(len(s1_ids), world))
    ga, gb = split_subgroups(ranks_a, ranks_b)
    role, group, ranks = ("s1", ga, ranks_a) if rank in ranks_a else ("s2", gb, ranks_b)
    reducer = GradReducer(len(ranks), group=group) if len(ranks) > 1 else None
    job = make(role, dev, reducer, ranks.index(rank), args)
    if reducer is not None:
        for flat in job.params():
            reducer.broadcast_params(flat, src=ranks[0])
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    for _ in range(args.warmup):
        job.step()
    sync()
    dist.barrier()                                                      # the one world-wide meeting: a common start
    t0 = time.perf_counter()
    deadline = t0 + args.minutes * 60.0
    flag = torch.zeros(1, dtype=torch.int32, device=dev if on_gpu else "cpu")
    steps = 0
    while True:
        job.step()
        steps += 1
        if steps % args.check_every == 0:
            if rank == ranks[0]:
                sync()
                flag.fill_(1 if time.perf_counter() >= deadline else 0)
            if len(ranks) > 1:
                dist.broadcast(flag, src=ranks[0], group=group)
            if int(flag.item()) == 1:
                break
    sync()
    dt = time.perf_counter() - t0
    # [s1 units, s1 seconds, s2 units, s2 seconds] summed over the world; each group's first rank reports its time
    rep = torch.zeros(4, dtype=torch.float64, device=dev if on_gpu else "cpu")
    o = 0 if role == "s1" else 2
    rep[o] = steps * job.units_per_step
    rep[o + 1] = dt if rank == ranks[0] else 0.0
    dist.all_reduce(rep)
    out = None
    if rank == 0:
        r = rep.tolist()
        out = {"config": f"joint s1+s2 on one node: s1 dp{len(ranks_a)} + s2 dp{len(ranks_b)}, synthetic batches, "
                         f"{args.minutes} min budget", "backend": args.backend,
               "s1": {"n_gpus": len(ranks_a), "tokens_per_sec": r[0] / max(r[1], 1e-9), "seconds": r[1]},
               "s2": {"n_gpus": len(ranks_b), "audio_seconds_per_sec": r[2] / max(r[3], 1e-9), "seconds": r[3]}}
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return out


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--s1-gpus", default="0-1")
    ap.add_argument("--s2-gpus", default="2-3-4-5-6-7")
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--check-every", type=int, default=8, help="steps between deadline checks (one 4-byte broadcast each)")
    ap.add_argument("--s1-batch", type=int, default=32)
    ap.add_argument("--s2-batch", type=int, default=16)
    ap.add_argument("--clip-seconds", type=int, default=4)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    return ap


def main():
    args = parser().parse_args()
    if "WORLD_SIZE" not in os.environ:
        from easevoice_trainer_amd.dist import parse_gpu_ids, spawn_ranks

        ids = parse_gpu_ids(args.s1_gpus) + parse_gpu_ids(args.s2_gpus)
        codes = spawn_ranks([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], ids)
        sys.exit(max(int(c or 0) != 0 for c in codes))
    run_rank(args)


if __name__ == "__main__":
    main()
