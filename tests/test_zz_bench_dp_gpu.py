"""GPU: `bench.py --gpus 2` end to end on the one GPU of the test box -- bench.py spawns its own two ranks
(dist.spawn_ranks), they rendezvous on 127.0.0.1, broadcast the parameters, run the data-parallel s2 step (overlapped
reductions between nine HIP graphs) and the s1 micro-steps, take the max over ranks and rank 0 prints the JSON line.  Both
ranks share device 0, which RCCL refuses, so the transport is gloo (EVT_BENCH_BACKEND); everything else is the code path the
driver's 8-GPU run takes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_device(gpu):
    env = dict(os.environ, EVT_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2",
           "--clip-seconds", "2", "--s1-batch", "2"]          # no --no-extras: with N > 1 the extra legs switch themselves off
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"].startswith("dp2, reduce-scatter + all-gather for buckets >= 8 MiB from 3 ranks up (after a start-up equality check") and d["config"]["global_batch"] == 4
    # the collective plan of every reduced range is spelt out (two ranks: all-reduce throughout)
    assert "per range: D piece 1/6" in d["config"]["parallelism"] and "G rest" in d["config"]["parallelism"]
    # the line explains its own gradient exchange: collectives and MiB per step, and the wait the overlap did not hide
    c = d["comm"]
    assert c["all_reduce_per_step"] >= 2 and c["mib_per_step"] > 300 and c["exposed_wait_ms_per_step"] is not None
    assert d["losses_finite"] and d["value"] > 0
    assert "graph" in d["config"]["launch"] and "reductions between them" in d["config"]["launch"]
    assert "roofline" not in d and "roofline_note" in d and "roofline" not in d["s1"]
    assert d["s1"]["n_gpus"] == 2 and d["s1"]["value"] > 0 and d["s1"]["config"]["parallelism"].startswith("dp2, reduce-scatter + all-gather")
    assert d["s1"]["comm"]["mib_per_step"] >= 0        # three micro-steps after one warm-up: no optimiser step in the window
