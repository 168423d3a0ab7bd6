"""The process boundary of SURVEY §8(b) on the GPU: both trainers started as CHILD PROCESSES the way the reference's service
starts them (src/service/session.py:234-253: Popen([python, cmd, "-c", params.json])), fed from a feature directory, their
stdout parsed line by line as the reference's connector parses it (src/utils/helper/connector.py:42-144): progress as
`loss-of-easevoice` records, ONE final `response-of-easevoice` record with status success and the model path, files on
disk in the reference's layout."""
import json
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import data_fixture as F  # noqa: E402
from test_process_boundary_cpu import parse_lines  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def train_dir(tmp_path_factory):
    with open(os.path.join(HERE, "golden", "data_readers.json")) as f:
        symbols = json.load(f)["symbols"]
    root = str(tmp_path_factory.mktemp("exp_child"))
    F.build_feature_dir(root, symbols)
    os.remove(os.path.join(root, "5-wav32k", "a_007.wav"))      # see test_zz_readers_train_gpu.py::train_dir
    with open(os.path.join(root, "symbols.json"), "w") as f:
        json.dump(symbols, f)
    return root


def _child(script, config, tmp_path):
    cfg = tmp_path / "params.json"
    cfg.write_text(json.dumps(config))
    env = {k: v for k, v in os.environ.items() if k != "EVT_SYNTHETIC_STEPS"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "cmd", script), "-c", str(cfg)], capture_output=True, text=True,
                       timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return parse_lines(r.stdout.splitlines()), r


def test_train_sovits_as_a_child_process(gpu, train_dir, tmp_path):
    g = torch.Generator().manual_seed(3)
    pre = "quantizer.vq.layers.0._codebook."
    torch.save({"weight": {pre + "inited": torch.ones(1), pre + "embed": torch.randn(1024, 768, generator=g),
                           pre + "embed_avg": torch.randn(1024, 768, generator=g), pre + "cluster_size": torch.ones(1024)}},
               str(tmp_path / "s2G.pth"))
    recs, r = _child("train_sovits.py", dict(batch_size=4, total_epochs=1, save_every_epoch=1, output_model_name="child",
                                             project_dir=str(tmp_path), train_input_dir=train_dir, gpu_ids="0",
                                             pretrained_s2G=str(tmp_path / "s2G.pth")), tmp_path)
    resp = [p for k, p in recs if k == "resp"]
    assert len(resp) == 1 and resp[0]["status"] == "success", (r.stdout[-2000:], r.stderr[-2000:])
    assert recs[-1][0] == "resp"
    losses = [p for k, p in recs if k == "loss"]
    assert len(losses) == 2 and all(p["loss"] == p["loss"] and abs(p["loss"]) < 1e4 for p in losses)
    assert [p["step"] for p in losses] == sorted(p["step"] for p in losses)
    model_path = resp[0]["data"]["model_path"]
    assert os.path.isdir(model_path) and any(f.startswith("child_e1_s") and f.endswith(".pth") for f in os.listdir(model_path))


def test_train_gpt_as_a_child_process(gpu, train_dir, tmp_path):
    recs, r = _child("train_gpt.py", dict(batch_size=8, total_epochs=1, save_every_epoch=1, output_model_name="childg",
                                          project_dir=str(tmp_path), train_input_dir=train_dir, gpu_ids="0"), tmp_path)
    resp = [p for k, p in recs if k == "resp"]
    assert len(resp) == 1 and resp[0]["status"] == "success", (r.stdout[-2000:], r.stderr[-2000:])
    losses = [p for k, p in recs if k == "loss"]
    assert len(losses) >= 1 and all(p["loss"] == p["loss"] and p["loss"] > 0 for p in losses)
    assert os.path.isfile(os.path.join(resp[0]["data"]["model_path"], "childg-e1.ckpt"))
