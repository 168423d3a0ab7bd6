"""Data-parallel gradient exchange over RCCL/xGMI for the flat gradient arenas (runtime.ParamArena).

What the reference does (src/train/sovits.py:321-322, Lightning DDP in src/train/gpt.py:147-162): torch DDP with
default 25 MiB buckets over several hundred gradient tensors; the discriminator's reducer fires a second, discarded
time during the generator backward; s1 all-reduces on each of the 4 accumulation micro-batches.

MI355X-first: gradients of a model are ONE contiguous fp32 buffer whose sub-models are contiguous ranges, so the
exchange is a handful of large collectives issued on a side HIP stream as soon as the backward that produced a range
is done (train/s2_engine.py::_program drives this; the results equal the plain two-reduction step, tested on the GPU
over gloo and on the CPU):
  * s2 discriminators: in the D step the six sub-discriminators are differentiated one after the other (their graphs
    are disjoint there); the ~31 MB range of sub-discriminator i is reduced while sub-discriminator i-1 runs its
    backward; the D optimiser waits for the side stream;
  * s2 generator: the autograd graph is cut at the vocoder's inputs; the backward through D and `dec` runs first, the
    vocoder's 58 MB range is reduced under the flow / encoder backward, the remaining ~146 MB after it; nothing is
    reduced twice (the reference's DDP reduces D a second, discarded time during the generator backward);
  * s1: ONE reduction of the 310 MB arena per optimiser step, i.e. per four micro-batches (~0.2 s of compute);
  * with HIP-graph replay the pieces are separate graphs and the collectives are issued between the replays;
  * bucket size defaults to 64 MiB: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so per-collective latency
    dominates small buckets; reduce-scatter + all-gather of a large bucket uses every link at once;
  * averaging (1/world) is folded into the AdamW launch (grad_scale), not a separate pass over the buffer.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, world_size: int, bucket_bytes: int = 64 << 20, group=None, rsag=None, rsag_min_bytes: int = 8 << 20,
                 force=None):
        """rsag (default: EVT_DP_RSAG = auto | 0 | 1): a bucket as reduce-scatter + all-gather instead of one all-reduce.
        On xGMI every GPU has a direct link to each of the other seven: a ring all-reduce is bound by ONE link per hop,
        while the two halves of reduce-scatter / all-gather move 1/world of the bucket to / from every peer at once; below a
        few MiB the second collective's latency costs more than the links give, so `auto` keeps all-reduce for buckets
        under rsag_min_bytes (and for worlds of two, where there is one link either way).  The results are the same sums
        (per element one reduction over the ranks either way); equality-tested against all_reduce on gloo
        (tests/test_host_cpu.py) and issued on RCCL by the one-rank self-test (tests/test_zz_rccl_selftest_gpu.py).
        Default auto: reduce-scatter + all-gather from 3 ranks up -- but only after `verify_rsag()` has run BOTH forms
        on the real group over a small integer-valued buffer and every rank saw identical sums (the pair has
        never met a multi-rank RCCL communicator; a mismatch or an exception on any rank puts all ranks back on
        all-reduce, with a warning).  The check runs once, at the first bucket that would take the pair -- before any
        graph capture, since the first steps of a shape are eager.  EVT_DP_RSAG=0 is the all-reduce fallback,
        EVT_DP_RSAG=1 forces the pair without the check.
        force (default: EVT_DP_FORCE=1): issue the collectives even in a world of one -- a one-rank RCCL group runs the
        same launches on the same side stream between the same graph replays, so the stream / replay ordering of the
        data-parallel program can be exercised (and its cost timed) on a single GPU: bench.py --dp-program 2.
        The reducer keeps count of what it issued (collectives, bytes, which kind) and -- with timing on -- of how long
        the compute stream had to wait for the side stream: bench.py prints both."""
        self.world = world_size
        self.group = group
        self.force = (os.environ.get("EVT_DP_FORCE", "0") == "1") if force is None else bool(force)
        self.bucket_elems = max(1, bucket_bytes // 4)
        mode = os.environ.get("EVT_DP_RSAG", "auto") if rsag is None else rsag
        self.rsag_mode = {True: "1", False: "0"}.get(mode, str(mode))
        if self.rsag_mode not in ("0", "1", "auto"):
            raise ValueError(f"EVT_DP_RSAG / rsag must be 0, 1 or auto, got {mode!r}")
        self.rsag_min_elems = max(1, rsag_min_bytes // 4)
        self._pending = []
        self._stream = None
        self._shards = {}
        self.stats = {"all_reduce": 0, "rs_ag": 0, "bytes": 0}
        self.timing = False
        self._waits = []
        self.rsag_verified = None     # auto mode: None = not checked yet, True / False = result of verify_rsag()

    @property
    def rsag(self):
        return self.rsag_mode != "0"

    @property
    def active(self):
        """whether collectives are issued at all: more than one rank, or a forced one-rank run"""
        return self.world > 1 or self.force

    def _use_rsag(self, n):
        if self.rsag_mode == "1":
            return n >= self.world
        return (self.rsag_mode == "auto" and self.rsag_verified is not False and self.world > 2
                and n >= max(self.world, self.rsag_min_elems))

    def _rs_ag(self, body, chunk):
        # the reduced shard lands in a buffer of its own (an output aliasing the input is something only gloo has run), the
        # all-gather then writes every rank's shard back over the bucket
        key = (chunk, body.dtype, body.device)
        mine = self._shards.get(key)
        if mine is None:
            mine = self._shards[key] = torch.empty(chunk, dtype=body.dtype, device=body.device)
        dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_gather_into_tensor(body, mine, group=self.group)

    def verify_rsag(self, device):
        """auto mode, once per reducer: reduce-scatter + all-gather against all_reduce on THIS group.  Small integers
        (every partial sum exact in fp32), a different vector per rank, a length that is not a multiple of the world.
        The verdict is itself reduced (MIN over ranks), so all ranks take the same branch afterwards."""
        if self.rsag_verified is not None:
            return self.rsag_verified
        ok = 1.0
        try:
            rank = dist.get_rank(self.group)
            n = 4096 * self.world
            idx = torch.arange(n, device=device, dtype=torch.float32)
            a = ((idx * 7 + rank * 13) % 251) - 125.0
            b = a.clone()
            dist.all_reduce(a, op=dist.ReduceOp.SUM, group=self.group)
            self._rs_ag(b, n // self.world)
            if not torch.equal(a, b):
                ok = 0.0
        except Exception as e:  # noqa: BLE001 -- any failure of the pair means: do not use it
            import warnings

            warnings.warn(f"GradReducer: reduce-scatter + all-gather self-check raised {e!r}")
            ok = 0.0
        flag = torch.tensor([ok], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        self.rsag_verified = bool(flag.item() == 1.0)
        self._shards.pop((4096, torch.float32, device), None)
        if not self.rsag_verified:
            import warnings

            warnings.warn("GradReducer: reduce-scatter + all-gather did not reproduce all_reduce on this group; "
                          "using all_reduce for every bucket (EVT_DP_RSAG=0 behaviour)")
        return self.rsag_verified

    def plan(self, n):
        """what all_reduce() issues for a flat range of n fp32 elements: [(bytes, "all-reduce" | "reduce-scatter + all-gather")]"""
        return [((e - b) * 4, "reduce-scatter + all-gather" if self._use_rsag(e - b) else "all-reduce")
                for b, e in self.buckets(n)]

    def _sum_bucket(self, t):
        """sum the 1-D contiguous bucket `t` over the group, in place"""
        self.stats["bytes"] += t.numel() * t.element_size()
        if self.rsag_mode == "auto" and self.rsag_verified is None and self._use_rsag(t.numel()):
            self.verify_rsag(t.device)
        if not self._use_rsag(t.numel()):
            self.stats["all_reduce"] += 1
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return
        self.stats["rs_ag"] += 1
        w = self.world
        chunk = t.numel() // w
        body = t[: chunk * w]
        self._rs_ag(body, chunk)
        if chunk * w < t.numel():                           # fewer than `world` leftover elements
            dist.all_reduce(t[chunk * w:], op=dist.ReduceOp.SUM, group=self.group)

    def describe(self, ranges=None):
        """how gradients are exchanged, for the bench line's config.parallelism.  ranges: [(name, elements)] of the flat
        ranges the step reduces one by one -> the plan per range (MiB, kind) is spelt out."""
        kind = {"0": "all-reduce", "1": "reduce-scatter + all-gather",
                "auto": (f"reduce-scatter + all-gather for buckets >= {self.rsag_min_elems * 4 >> 20} MiB from 3 ranks up "
                         "(after a start-up equality check against all-reduce on the group), all-reduce otherwise")}
        s = f"dp{self.world}, {kind[self.rsag_mode]}, buckets of {self.bucket_elems * 4 >> 20} MiB"
        if self.force and self.world == 1:
            s += ", one-rank collectives forced"
        if ranges:
            parts = []
            for name, n in ranges:
                pl = self.plan(n)
                kinds = sorted({k for _b, k in pl})
                parts.append(f"{name} {n * 4 / (1 << 20):.1f} MiB = {len(pl)} x {' / '.join(kinds)}")
            s += "; per range: " + "; ".join(parts)
        return s

    def _side_stream(self, device):   # (a library-owned HIP stream, not one of torch's 32 pooled ones: hip/lib.py::role_stream)
        if self._stream is None and device.type == "cuda":
            from .hip import lib as L

            self._stream = L.role_stream(device, "comm")
        return self._stream

    def buckets(self, n):
        b = self.bucket_elems
        return [(i, min(n, i + b)) for i in range(0, n, b)]

    def all_reduce(self, flat: torch.Tensor, async_op: bool = False, average: bool = False):
        """sum `flat` (1-D contiguous) over the group, in place.  async_op=True: enqueue on a side stream and return;
        call wait() before the optimiser reads the buffer.  average=True: the 1/world pass rides on the same stream right
        behind the collectives (off the compute stream when async)."""
        if not self.active:
            return
        assert flat.dim() == 1 and flat.is_contiguous()
        if flat.device.type == "cuda" and async_op:
            side = self._side_stream(flat.device)
            side.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(side):
                for b, e in self.buckets(flat.numel()):
                    self._sum_bucket(flat[b:e])
                if average:
                    flat.mul_(1.0 / self.world)
            self._pending.append(flat)
            return
        for b, e in self.buckets(flat.numel()):
            self._sum_bucket(flat[b:e])
        if average:
            flat.mul_(1.0 / self.world)

    def wait(self):
        if self._stream is not None and self._pending:
            cur = torch.cuda.current_stream()
            if self.timing and not torch.cuda.is_current_stream_capturing():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self._stream)
                e1.record(cur)
                self._waits.append((e0, e1))
            else:
                cur.wait_stream(self._stream)
        self._pending.clear()

    def comm_report(self, steps: int):
        """{collectives, MiB and exposed wait per step}: the wait is what the compute stream stood still for at wait()
        (HIP events around it; needs timing = True and a synchronised device)"""
        ms = sum(a.elapsed_time(b) for a, b in self._waits)
        n = max(1, steps)
        out = {"all_reduce_per_step": self.stats["all_reduce"] / n, "rs_ag_per_step": self.stats["rs_ag"] / n,
               "mib_per_step": self.stats["bytes"] / n / (1 << 20),
               "exposed_wait_ms_per_step": (ms / n) if self._waits else None, "waits_per_step": len(self._waits) / n}
        return out

    def reset_stats(self):
        self.stats = {"all_reduce": 0, "rs_ag": 0, "bytes": 0}
        self._waits = []

    def broadcast_params(self, flat: torch.Tensor, src: int = 0):
        """one-time parameter broadcast from rank `src` (DDP's wrap-time broadcast)"""
        if not self.active:
            return
        for b, e in self.buckets(flat.numel()):
            dist.broadcast(flat[b:e], src=src, group=self.group)

    def all_reduce_scalars(self, t: torch.Tensor):
        """batched metric reduction (the reference does three separate sync_dist scalar all-reduces per step)"""
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.div_(self.world)
        return t


def init_process_group_from_env(backend=None, gpu_ids=None):
    """one process per GPU; rendezvous over 127.0.0.1 (single node), RCCL when a GPU is present.
    Returns (world, rank, local device index).  A single-process run without a launcher takes its device from
    `gpu_ids` (the reference exports CUDA_VISIBLE_DEVICES=gpu_ids, src/train/sovits.py:168: gpu_ids="2" trains on
    card 2, not on card 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        if "LOCAL_RANK" in os.environ:
            local = int(os.environ["LOCAL_RANK"])
        else:
            ids = parse_gpu_ids(gpu_ids) if gpu_ids is not None else []
            local = ids[0] if ids else 0
            if len(ids) > 1:
                import warnings

                warnings.warn(f"gpu_ids={gpu_ids!r} names {len(ids)} GPUs but this is a single process (no launcher: "
                              f"WORLD_SIZE=1): training on GPU {local} only -- start it through cmd/train_*.py, which "
                              "spawns one rank per listed GPU")
            # a scheduler (or the reference's own CUDA_VISIBLE_DEVICES=gpu_ids export, src/train/sovits.py:168) may have
            # narrowed the visible devices already: id 2 with one visible device is that device
            nvis = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if nvis and local >= nvis:
                narrowed = any(os.environ.get(v) for v in ("CUDA_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES",
                                                           "ROCR_VISIBLE_DEVICES"))
                if not narrowed:
                    # nothing narrowed the visible devices: "2" on a 2-GPU box is a mistake, and folding it onto card 0
                    # would put two independent jobs on one device without a word
                    raise ValueError(f"gpu_ids={gpu_ids!r} names GPU {local}, this node has {nvis} (and no "
                                     "*_VISIBLE_DEVICES narrowing that would explain the id)")
                local = local % nvis
        return world, int(os.environ.get("RANK", "0")), local
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return world, rank, local


def split_subgroups(ranks_a, ranks_b):
    """BASELINE config 5 (s1 on some ranks, s2 on the others): two sub-communicators from one world.  Every rank
    must call this with the same arguments."""
    ga = dist.new_group(ranks=list(ranks_a))
    gb = dist.new_group(ranks=list(ranks_b))
    return ga, gb


def parse_gpu_ids(gpu_ids: str):
    """"0-1-2" (SovitsTrainParams / GPTTrainParams.gpu_ids, src/train/sovits.py:47) -> [0, 1, 2]"""
    return [int(t) for t in str(gpu_ids).replace(",", "-").split("-") if t.strip() != ""]


def spawn_ranks(argv, gpu_ids, poll_s=0.2):
    """One process per listed GPU on this node: re-runs `argv` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
    MASTER_PORT set (LOCAL_RANK = the GPU id, so "0-2" uses devices 0 and 2).  Every child inherits stdout/stderr: only
    rank 0 prints protocol lines (the trainers print on rank 0, the cmd entry points answer on rank 0).  When a child
    exits non-zero the others are terminated (by PID) -- a rank stuck in a collective would otherwise wait forever.
    Returns the list of exit codes.  This is what `mp.spawn` / Lightning's launcher do for the reference
    (src/train/sovits.py:199-211, src/train/gpt.py:147-162), as a plain process group."""
    import socket
    import subprocess
    import time

    ids = list(gpu_ids)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    stamp = os.environ.get("EVT_RUN_STAMP") or time.strftime("%Y%m%d-%H%M%S")     # one run name for all ranks (train/helper.py)
    for rank, dev in enumerate(ids):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(dev), WORLD_SIZE=str(len(ids)), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   EVT_SPAWNED="1", EVT_RUN_STAMP=stamp)
        procs.append(subprocess.Popen(list(argv), env=env))
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, pr in enumerate(procs):
            if codes[i] is None:
                codes[i] = pr.poll()
        if any(c not in (None, 0) for c in codes):
            for i, pr in enumerate(procs):
                if codes[i] is None:
                    pr.terminate()
            for i, pr in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = pr.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        pr.kill()
                        codes[i] = pr.wait()
            break
        time.sleep(poll_s)
    return codes
