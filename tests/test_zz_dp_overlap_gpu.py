"""GPU: the data-parallel s2 step with overlapped gradient exchange (train/s2_engine.py::_program) must train like the step
with the two plain whole-arena reductions, and like one process that sees both ranks' items as one batch.  Two ranks on
the box's single GPU over gloo (see dp_worker_s2.py)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "dp_worker_s2.py")


def _run(tmp_path, mode, steps, graphs):
    from easevoice_trainer_amd.dist import spawn_ranks

    pre = str(tmp_path / f"r{graphs}")
    if mode == "single":
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        subprocess.run([sys.executable, WORKER, pre, mode, str(steps), str(graphs)], check=True, env=env, timeout=600)
        return [torch.load(f"{pre}_{mode}_0.pt")]
    codes = spawn_ranks([sys.executable, WORKER, pre, mode, str(steps), str(graphs)], [0, 0])
    assert codes == [0, 0], codes
    return [torch.load(f"{pre}_{mode}_{r}.pt") for r in range(2)]


def _rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _cos(a, b):
    return (torch.dot(a.double(), b.double()) / (a.double().norm() * b.double().norm() + 1e-30)).item()


@pytest.mark.parametrize("graphs", [0, 1], ids=["eager", "graphs"])
def test_overlapped_exchange_equals_plain_and_full_batch(gpu, tmp_path, graphs):
    steps = 3 if graphs else 2          # graph mode: one eager step, the capture step, one replay
    ov = _run(tmp_path, "overlap", steps, graphs)
    sy = _run(tmp_path, "sync", steps, graphs)
    for k in ("g", "d"):
        assert torch.equal(ov[0][k], ov[1][k]), "replicas diverged"
        assert torch.equal(ov[0]["grads"][k], ov[1]["grads"][k]), "ranks hold different reduced gradients"
        # reduced gradients: the same sums in a different launch order (fp32 atomics in the weight gradients)
        assert _rel(ov[0]["grads"][k], sy[0]["grads"][k]) < 1e-4, (k, _rel(ov[0]["grads"][k], sy[0]["grads"][k]))
        assert ov[0]["grads"][k].abs().max() > 0
        # the trained parameters: AdamW's update is sign-like for tiny gradients, so compare the update DIRECTION
        du, ds = ov[0][k] - ov[0]["p0"][k], sy[0][k] - sy[0]["p0"][k]
        assert _cos(du, ds) > 0.999, (k, _cos(du, ds))
    if not graphs:
        one = _run(tmp_path, "single", steps, 0)
        for k in ("g", "d"):
            # sum over the two ranks = 2 x the gradient of the mean loss over both items
            assert _rel(ov[0]["grads"][k] * 0.5, one[0]["grads"][k]) < 1e-3, (k, _rel(ov[0]["grads"][k] * 0.5, one[0]["grads"][k]))
        avg = (torch.tensor(ov[0]["losses"]) + torch.tensor(ov[1]["losses"])) / 2
        assert _rel(avg[0], torch.tensor(one[0]["losses"])[0]) < 1e-3
