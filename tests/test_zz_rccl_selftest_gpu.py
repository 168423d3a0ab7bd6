"""GPU: the data-parallel paths on a one-rank RCCL communicator (tests/rccl_selftest_worker.py).  The multi-GPU runs are the
driver's; what a single GPU can establish is that RCCL initialises in this environment, that every collective the reducer
issues (all-reduce, reduce-scatter into the shard buffer + all-gather, broadcast) runs on the side stream in order with the
compute stream and with HIP-graph replays, and that the cut (nine-graph) s2 program and the hook-driven s1 pieces with
REAL collectives between them train exactly like the plain program.  Reference: src/train/sovits.py:219-224,321-322
(init_process_group("nccl") + DDP), src/train/gpt.py:147-162."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "rccl_selftest_worker.py")


def _run(tmp_path, mode, port):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    pre = str(tmp_path / "o")
    subprocess.run([sys.executable, WORKER, pre, mode], check=True, env=env, timeout=900)
    return pre


def _cos(a, b):
    return (torch.dot(a.double(), b.double()) / (a.double().norm() * b.double().norm() + 1e-30)).item()


def _rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def test_reducer_primitives_on_one_rank_rccl(gpu, tmp_path):
    pre = _run(tmp_path, "prims", 29561)
    r = json.load(open(pre + "_prims.json"))
    for mode in ("0", "1"):
        assert r[f"all_reduce_rsag{mode}"] and r[f"broadcast_rsag{mode}"], r
        assert r[f"scalars_rsag{mode}"] == [1.0, 2.0, 3.0]
    assert r["stats_rsag0"]["all_reduce"] == 4 and r["stats_rsag0"]["rs_ag"] == 0, r      # 3 full buckets + the ragged one
    assert r["stats_rsag1"]["rs_ag"] == 4, r
    assert r["graph_interleave"] == [5.0, 15.0], r


def test_s2_cut_program_with_rccl_collectives_equals_plain(gpu, tmp_path):
    pre = _run(tmp_path, "s2_rccl", 29562)
    _run(tmp_path, "s2_plain", 29563)
    a, b = torch.load(pre + "_s2_rccl.pt"), torch.load(pre + "_s2_plain.pt")
    assert a["graphs"] == 8 + int(os.environ.get("EVT_DP_G_PIECES", "1")) and b["graphs"] == 3
    assert a["replayed"] >= 2 and b["replayed"] >= 2
    assert a["stats"]["rs_ag"] > 0, a["stats"]           # the reduce-scatter + all-gather path did run on RCCL
    for k in ("g", "d"):
        assert _rel(a["grads"][k], b["grads"][k]) < 1e-4, (k, _rel(a["grads"][k], b["grads"][k]))
        du, dv = a[k] - a["p0"][k], b[k] - b["p0"][k]
        assert _cos(du, dv) > 0.999, (k, _cos(du, dv))
    la, lb = torch.tensor(a["losses"]), torch.tensor(b["losses"])
    assert _rel(la[0], lb[0]) < 1e-4


def test_s1_pieces_with_rccl_collectives_equal_plain(gpu, tmp_path):
    pre = _run(tmp_path, "s1_rccl", 29564)
    _run(tmp_path, "s1_plain", 29565)
    a, b = torch.load(pre + "_s1_rccl.pt"), torch.load(pre + "_s1_plain.pt")
    assert a["stepped"] == b["stepped"] == [False] * 4 + [True]
    assert a["stats"]["rs_ag"] + a["stats"]["all_reduce"] >= 3      # the default cuts "16,8": three pieces
    assert _rel(a["p"], b["p"]) < 1e-5, _rel(a["p"], b["p"])
