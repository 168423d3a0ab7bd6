"""GPU: the trainer classes end to end on synthetic batches — directory / checkpoint / export layout, stdout lines,
resume — the surface src/service drives."""
import io
import json
import os
from contextlib import redirect_stdout

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_sovits_train_layout_and_resume(gpu, tmp_path, monkeypatch):
    from easevoice_trainer_amd.train.sovits import SovitsTrain, SovitsTrainParams

    monkeypatch.setenv("EVT_SYNTHETIC_STEPS", "3")
    p = SovitsTrainParams(batch_size=2, total_epochs=1, save_every_epoch=1, output_model_name="unit", project_dir=str(tmp_path))
    buf = io.StringIO()
    with redirect_stdout(buf):
        out = SovitsTrain(p).train()
    d = out.model_path
    assert d == os.path.join(str(tmp_path), "models", "sovits_train", "unit")
    assert os.path.isfile(os.path.join(d, "logs", "G_latest.pth")) and os.path.isfile(os.path.join(d, "logs", "D_latest.pth"))
    exp = torch.load(os.path.join(d, "unit_e1_s3.pth"), weights_only=False)
    assert set(exp) == {"weight", "config", "info"} and exp["info"] == "1epoch_3iteration"
    assert not any("enc_q" in k for k in exp["weight"]) and len(exp["weight"]) == 776 - 103
    assert all(v.dtype == torch.float16 for v in exp["weight"].values() if v.is_floating_point())
    assert exp["weight"]["enc_p.text_embedding.weight"].shape[0] == 732
    g = torch.load(os.path.join(d, "logs", "G_latest.pth"), weights_only=False)
    assert set(g) == {"model", "iteration", "optimizer", "learning_rate"} and g["iteration"] == 1
    assert len(g["optimizer"]["param_groups"]) == 4          # base / text_embedding / encoder_text / mrte
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("loss-of-easevoice")]
    assert len(lines) == 1 and json.loads(lines[0].split(" ", 1)[1])["step"] == 0
    # resume: iteration 1 is loaded, epochs already done -> no further step, global step restored
    p2 = SovitsTrainParams(batch_size=2, total_epochs=2, save_every_epoch=1, output_model_name="unit", project_dir=str(tmp_path))
    t2 = SovitsTrain(p2)
    with redirect_stdout(io.StringIO()):
        t2.train()
    assert t2.global_step == 3 + 3 and os.path.isfile(os.path.join(d, "unit_e2_s6.pth"))


def test_gpt_train_layout(gpu, tmp_path, monkeypatch):
    from easevoice_trainer_amd.train.gpt import GPTTrain, GPTTrainParams

    monkeypatch.setenv("EVT_SYNTHETIC_STEPS", "6")
    p = GPTTrainParams(batch_size=2, total_epochs=1, save_every_epoch=1, output_model_name="g", project_dir=str(tmp_path))
    buf = io.StringIO()
    with redirect_stdout(buf):
        out = GPTTrain(p).train()
    d = out.model_path
    ck = os.listdir(os.path.join(d, "logs", "ckpt"))
    assert ck == ["epoch=0-step=1.ckpt"]                      # one optimiser step at batch_idx 4
    exp = torch.load(os.path.join(d, "g-e1.ckpt"), weights_only=False)
    assert set(exp) == {"weight", "config", "info"} and exp["info"] == "GPT-e1"
    assert len(exp["weight"]) == 295 and all(k.startswith("model.") for k in exp["weight"])
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("loss-of-easevoice")]
    assert len(lines) == 6
    assert 6.0 < json.loads(lines[0].split(" ", 1)[1])["loss"] / (2 * 768) < 8.5    # ~ln(1025) per token at init
