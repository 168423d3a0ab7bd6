"""CPU: the trainers' control flow end to end -- epochs over a feature directory, stdout protocol cadence, TensorBoard
scalars, checkpoints / exports / resume -- without a GPU.  s1 runs the real engine with emulated launches
(tests/cpu_emu.py); s2 runs the real SovitsTrain loop around a stand-in engine (the s2 kernels have no CPU emulation of
the optimiser side), so only the glue is under test there.  The GPU tier runs both trainers for real."""
import io
import json
import os
import sys
from contextlib import redirect_stdout
from types import SimpleNamespace

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import data_fixture as F  # noqa: E402
from cpu_emu import cpu_emulation_s1  # noqa: E402


@pytest.fixture()
def feature_dir(tmp_path):
    gold = json.load(open(os.path.join(HERE, "golden", "data_readers.json")))
    root = str(tmp_path / "exp")
    os.makedirs(root)
    F.build_feature_dir(root, gold["symbols"])
    os.remove(os.path.join(root, "5-wav32k", "a_007.wav"))      # hubert longer than the spectrogram: no model accepts it
    with open(os.path.join(root, "symbols.json"), "w") as f:
        json.dump(gold["symbols"], f)
    return root


def _events(path):
    """(step, {tag: value}) records of a TensorBoard event file (format checked in test_host_cpu.py)"""
    import struct
    blob, pos, out = open(path, "rb").read(), 0, []
    while pos < len(blob):
        (n,) = struct.unpack_from("<Q", blob, pos)
        rec, pos = blob[pos + 12:pos + 12 + n], pos + 16 + n
        step, tags, i = 0, {}, 0
        while i < len(rec):
            key = rec[i]
            i += 1
            if key == 0x09:
                i += 8
            elif key == 0x10:
                step, sh = 0, 0
                while True:
                    b = rec[i]
                    i += 1
                    step |= (b & 0x7F) << sh
                    sh += 7
                    if not b & 0x80:
                        break
            else:
                ln = rec[i]
                i += 1
                if ln & 0x80:
                    ln = (ln & 0x7F) | (rec[i] << 7)
                    i += 1
                body = rec[i:i + ln]
                i += ln
                if key == 0x2A:
                    j = 0
                    while j < len(body):
                        vl = body[j + 1]
                        v = body[j + 2:j + 2 + vl]
                        tl = v[1]
                        tags[v[2:2 + tl].decode()] = struct.unpack("<f", v[2 + tl + 1:2 + tl + 5])[0]
                        j += 2 + vl
        out.append((step, tags))
    return out


def test_gpt_trainer_loop_from_feature_dir(feature_dir, tmp_path, monkeypatch):
    from easevoice_trainer_amd.train import gpt as G

    monkeypatch.setenv("EVT_TB_DIR", str(tmp_path / "tb"))
    monkeypatch.delenv("EVT_SYNTHETIC_STEPS", raising=False)
    monkeypatch.setattr(G.GPTTrain, "_device", staticmethod(lambda local: torch.device("cpu")))
    small = dict(hidden_dim=64, embedding_dim=64, head=4, n_layer=2, linear_units=256)

    def make(total_epochs):
        t = G.GPTTrain(G.GPTTrainParams(batch_size=8, total_epochs=total_epochs, save_every_epoch=1, output_model_name="g",
                                        project_dir=str(tmp_path), train_input_dir=feature_dir), dtype=torch.float32)
        t.config["model"].update(small)
        return t

    with cpu_emulation_s1():
        buf = io.StringIO()
        with redirect_stdout(buf):
            out = make(1).train()
        lines = [json.loads(l.split(" ", 1)[1]) for l in buf.getvalue().splitlines() if l.startswith("loss-of-easevoice")]
        assert len(lines) == 12 and [l["step"] for l in lines] == list(range(12))           # 96 items / batch 8
        assert all(l["loss"] > 0 and l["epoch"] == 0 and 0.0 <= l["acc"] <= 1.0 for l in lines)
        assert [l["lr"] for l in lines] == [1e-05] * 4 + [0.01] * 4 + [0.002] * 4      # get_last_lr() around the pinned-lr steps
        d = out.model_path
        assert d == os.path.join(str(tmp_path), "models", "gpt_train", "g")
        assert os.listdir(os.path.join(d, "logs", "ckpt")) == ["epoch=0-step=2.ckpt"]       # optimiser steps at idx 4, 8
        exp = torch.load(os.path.join(d, "g-e1.ckpt"), weights_only=False)
        assert set(exp) == {"weight", "config", "info"} and exp["info"] == "GPT-e1"
        assert all(k.startswith("model.") and v.dtype == torch.float16 for k, v in exp["weight"].items())
        ev = _events(os.path.join(str(tmp_path / "tb"), "g", "version_0", os.listdir(tmp_path / "tb" / "g" / "version_0")[0]))
        assert [s for s, _ in ev] == [0] + list(range(12)) and set(ev[1][1]) == {"total_loss_step", "lr", "top_3_acc_step"}
        assert abs(ev[3][1]["total_loss_step"] - lines[2]["loss"]) < 1e-3 * lines[2]["loss"]
        # resume: epoch 0 is done -> a 2-epoch run continues with epoch 1, global step keeps counting, new TB version
        buf = io.StringIO()
        with redirect_stdout(buf):
            t2 = make(2)
            t2.train()
        lines2 = [json.loads(l.split(" ", 1)[1]) for l in buf.getvalue().splitlines() if l.startswith("loss-of-easevoice")]
        assert len(lines2) == 12 and all(l["epoch"] == 1 for l in lines2) and t2.global_step == 4
        assert os.listdir(os.path.join(d, "logs", "ckpt")) == ["epoch=1-step=4.ckpt"]       # if_save_latest: older one removed
        assert os.path.isfile(os.path.join(d, "g-e2.ckpt")) and os.path.isdir(tmp_path / "tb" / "g" / "version_1")


class _FakeS2Engine:
    """stands in for train/s2_engine.S2Engine: same surface towards SovitsTrain._run, trivial arithmetic"""

    def __init__(self, hps, device, dtype, reducer=None):
        from easevoice_trainer_amd.runtime import FlatAdamW, ParamArena

        self.net_g = torch.nn.Sequential(torch.nn.Linear(4, 4))
        self.net_d = torch.nn.Sequential(torch.nn.Linear(4, 2))
        cb = SimpleNamespace(inited=torch.ones(1), embed=torch.zeros(2, 2))
        self.net_g.quantizer = SimpleNamespace(vq=SimpleNamespace(layers=[SimpleNamespace(_codebook=cb)]))
        self.rt_g = SimpleNamespace(arena=ParamArena(self.net_g, "cpu"))
        self.rt_d = SimpleNamespace(arena=ParamArena(self.net_d, "cpu"))
        self._mk = lambda rt, net: FlatAdamW(rt.arena, [dict(names=[n for n, _ in net.named_parameters()], lr=hps["train"]["learning_rate"])])
        self.steps, self.graphs = 0, None

    def build_optimizers(self):
        self.optim_g, self.optim_d = self._mk(self.rt_g, self.net_g), self._mk(self.rt_d, self.net_d)
        return self.optim_g, self.optim_d

    def enable_graphs(self, **kw):
        self.graphs = kw

    def step(self, ssl, spec, spec_len, y, text, text_len):
        from easevoice_trainer_amd.train.s2_engine import S2Losses

        assert ssl.shape[2] == spec.shape[2] and spec.shape[1] == 1025 and y.shape[1] == 1 and text.dtype == torch.long
        self.steps += 1
        self.optim_g.step_count += 1
        self.optim_d.step_count += 1
        v = lambda x: torch.tensor(float(x))
        return S2Losses(v(2.0), v(3.0), v(0.5), v(40.0 - self.steps), v(1.5), v(0.0), v(45.0 - self.steps), torch.tensor([4.0]),
                        torch.tensor([9.0]))


def test_sovits_trainer_loop_glue(feature_dir, tmp_path, monkeypatch):
    from easevoice_trainer_amd.train import dataset as D
    from easevoice_trainer_amd.train import sovits as S
    from oracle.s2_step import stft_mag

    monkeypatch.setenv("EVT_TB_DIR", str(tmp_path / "tb"))
    monkeypatch.delenv("EVT_SYNTHETIC_STEPS", raising=False)
    monkeypatch.setattr(S.SovitsTrain, "_device", staticmethod(lambda local: torch.device("cpu")))
    monkeypatch.setattr(S, "S2Engine", _FakeS2Engine)
    real_reader = D.S2Reader
    monkeypatch.setattr(D, "S2Reader", lambda *a, **k: real_reader(*a, spec_fn=lambda y, n_fft, sr, hop, win, center=False:
                                                                   stft_mag(y, n_fft, hop), **k))

    def run(total_epochs):
        tr = S.SovitsTrain(S.SovitsTrainParams(batch_size=4, total_epochs=total_epochs, save_every_epoch=1,
                                               output_model_name="v", project_dir=str(tmp_path), train_input_dir=feature_dir))
        buf = io.StringIO()
        with redirect_stdout(buf):
            out = tr.train()
        return tr, out, [json.loads(l.split(" ", 1)[1]) for l in buf.getvalue().splitlines() if l.startswith("loss-of-easevoice")]

    tr, out, lines = run(1)
    d = out.model_path
    assert tr.global_step == 20 and [l["step"] for l in lines] == [0, 10]                   # 77 items -> 20 batches of 4
    assert lines[0]["loss"] == 44.0 and lines[0]["loss/d/total"] == 2.0 and lines[1]["loss/g/total"] == 34.0
    assert lines[0]["learning_rate"] == pytest.approx(1e-4 * 0.999875)                      # ExponentialLR fast-forwarded once
    assert tr.engine.graphs == dict(warmup_steps=2, max_shapes=64)
    assert sorted(os.listdir(os.path.join(d, "logs"))) == ["D_latest.pth", "G_latest.pth"]
    ck = torch.load(os.path.join(d, "logs", "G_latest.pth"), weights_only=False)
    assert set(ck) == {"model", "iteration", "optimizer", "learning_rate"} and ck["iteration"] == 1
    exp = torch.load(os.path.join(d, "v_e1_s20.pth"), weights_only=False)
    assert set(exp) == {"weight", "config", "info"} and exp["info"] == "1epoch_20iteration"
    tbdir = tmp_path / "tb" / "v"
    ev = _events(str(tbdir / os.listdir(tbdir)[0]))
    assert [s for s, _ in ev] == [0, 0, 5, 10, 15]
    assert set(ev[1][1]) == {"loss/g/total", "loss/d/total", "learning_rate", "grad_norm_d", "grad_norm_g", "loss/g/fm",
                             "loss/g/mel", "loss/g/kl_ssl", "loss/g/kl"}
    assert ev[1][1]["grad_norm_d"] == 2.0 and ev[1][1]["grad_norm_g"] == 3.0 and ev[2][1]["loss/g/total"] == 39.0
    # resume from G_latest / D_latest as the reference does it (sovits.py:327-341): the saved `iteration` is the epoch
    # number and becomes the FIRST epoch of the resumed run again, the step counter restarts at (epoch - 1) * len(loader)
    tr2, _, lines2 = run(2)
    assert tr2.global_step == 40 and [l["step"] for l in lines2] == [0, 10, 20, 30]
    assert lines2[0]["learning_rate"] == pytest.approx(1e-4 * 0.999875) and lines2[2]["learning_rate"] == pytest.approx(1e-4 * 0.999875 ** 2)
    assert os.path.isfile(os.path.join(d, "v_e2_s40.pth"))
    assert torch.load(os.path.join(d, "logs", "D_latest.pth"), weights_only=False)["iteration"] == 2
    assert tr2.engine.optim_g.step_count == 20 + 40            # the optimiser state did come from the checkpoint
