"""GPU: the BASELINE configs[4] launcher (tools/launch_joint.py) with the REAL engines -- s1 data-parallel on two ranks, s2
data-parallel on two ranks, one world, two sub-communicators -- all four ranks on the box's single GPU over gloo.  What it
checks is that the run starts, broadcasts the parameters inside each group, steps with its gradient reductions, stops both
groups on the deadline and prints ONE JSON line with both throughputs; the numbers are not performance (one device shared by
four processes).  The CPU tier drives the same launcher with stand-in jobs (tests/test_host_cpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_joint_launcher_runs_both_engines_on_one_gpu(gpu):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "launch_joint.py"), "--s1-gpus", "0-0", "--s2-gpus", "0-0",
                        "--minutes", "0.1", "--backend", "gloo", "--warmup", "2", "--s1-batch", "4", "--s2-batch", "2",
                        "--check-every", "2"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["s1"]["n_gpus"] == 2 and out["s2"]["n_gpus"] == 2
    assert out["s1"]["tokens_per_sec"] > 0 and out["s2"]["audio_seconds_per_sec"] > 0
    assert out["s1"]["seconds"] >= 6.0 and out["s2"]["seconds"] >= 6.0          # both groups ran to the deadline
