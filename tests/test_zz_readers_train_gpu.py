"""GPU, end to end (named test_zz_* so that the trainer loops run after every kernel-parity test): both trainers fed from
a feature directory instead of synthetic batches.  The s2 run repeats ragged batch shapes, so it goes
through HIP-graph capture and replay; every logged loss must stay finite."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import data_fixture as F  # noqa: E402
from test_data_readers_cpu import CFG, D, close, tstat  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "data_readers.json")) as f:
        g = json.load(f)
    g["blobs"] = torch.load(os.path.join(HERE, "golden", "data_readers.pt"))
    return g


@pytest.fixture(scope="module")
def feature_dir(gold, tmp_path_factory):
    root = str(tmp_path_factory.mktemp("exp"))
    F.build_feature_dir(root, gold["symbols"])
    with open(os.path.join(root, "symbols.json"), "w") as f:
        json.dump(gold["symbols"], f)
    return root


@pytest.fixture(scope="module")
def train_dir(gold, tmp_path_factory):
    """the feature directory without a_007: its hubert features are longer than its spectrogram, which the readers
    handle like the reference does (one more replicate pad) but which no model step accepts, there or here"""
    root = str(tmp_path_factory.mktemp("exp_train"))
    F.build_feature_dir(root, gold["symbols"])
    os.remove(os.path.join(root, "5-wav32k", "a_007.wav"))
    with open(os.path.join(root, "symbols.json"), "w") as f:
        json.dump(gold["symbols"], f)
    return root


def _fake_pretrained_g(path):
    g = torch.Generator().manual_seed(3)
    pre = "quantizer.vq.layers.0._codebook."
    torch.save({"weight": {pre + "inited": torch.ones(1), pre + "embed": torch.randn(1024, 768, generator=g),
                           pre + "embed_avg": torch.randn(1024, 768, generator=g), pre + "cluster_size": torch.ones(1024)}},
               path)


def test_sovits_train_from_feature_dir(gpu, train_dir, tmp_path, monkeypatch):
    from easevoice_trainer_amd.train.sovits import SovitsTrain, SovitsTrainParams

    monkeypatch.delenv("EVT_SYNTHETIC_STEPS", raising=False)
    _fake_pretrained_g(str(tmp_path / "s2G.pth"))
    p = SovitsTrainParams(batch_size=4, total_epochs=1, save_every_epoch=1, output_model_name="fd", project_dir=str(tmp_path),
                          train_input_dir=train_dir, pretrained_s2G=str(tmp_path / "s2G.pth"))
    buf = io.StringIO()
    with redirect_stdout(buf):
        t = SovitsTrain(p)
        out = t.train()
    steps = len(D.S2BucketSampler(D.S2FeatureDir(train_dir, CFG).lengths, 4))
    assert steps == 20 and t.global_step == steps
    assert os.path.isfile(os.path.join(out.model_path, f"fd_e1_s{steps}.pth"))
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("loss-of-easevoice")]
    assert len(lines) == 2
    for l in lines:
        v = json.loads(l.split(" ", 1)[1])
        assert v["loss"] == v["loss"] and abs(v["loss"]) < 1e4


def test_gpt_train_from_feature_dir(gpu, feature_dir, tmp_path, monkeypatch):
    from easevoice_trainer_amd.train.gpt import GPTTrain, GPTTrainParams

    monkeypatch.delenv("EVT_SYNTHETIC_STEPS", raising=False)
    p = GPTTrainParams(batch_size=8, total_epochs=1, save_every_epoch=1, output_model_name="gd", project_dir=str(tmp_path),
                       train_input_dir=feature_dir)
    buf = io.StringIO()
    with redirect_stdout(buf):
        out = GPTTrain(p).train()
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("loss-of-easevoice")]
    assert len(lines) == 12                                   # 96 items / batch 8
    losses = [json.loads(l.split(" ", 1)[1])["loss"] for l in lines]
    assert all(v == v and v > 0 for v in losses)
    assert os.path.isfile(os.path.join(out.model_path, "gd-e1.ckpt"))


def test_padded_time_axis_changes_nothing(gpu):
    """The trainer pads a batch's time axes up to a multiple of 16 frames (train/data.py: EVT_PAD_FRAMES, a departure from the
    reference's collate layout that lets repeated shapes replay as HIP graphs).  That is only sound while every consumer masks
    by the lengths: one fp32 step on a batch of two ragged items, once as the reference collates it (T = 102 for a longest
    item of 100 frames) and once with 12 more frames of padding on every time axis (ssl, spectrogram, waveform, the injected
    noise), must give the same waveform, losses and gradients.  The first version of this test FAILED: the reference's style
    encoder convolves over unmasked frames up to the tensor's end (modules.py:748-756), so its style vector -- and the
    waveform, by more than 1e-3 -- depended on the padding; MelStyleEncoder.mask_beyond_collate (set by the trainer whenever
    its reader pads) restores the reference's view.  A future unmasked reduction over T -- or an unfrozen quantiser, whose
    commitment loss is one -- fails here."""
    from easevoice_trainer_amd.module.mel_processing import spectrogram_torch
    from easevoice_trainer_amd.train.s2_engine import S2Engine
    from util_fill import fill_module, s2_batch

    hps = json.load(open(os.path.join(os.path.dirname(HERE), "configs", "s2.json")))
    hps["model"]["p_dropout"] = 0.0
    b = s2_batch(2, 100, 40)
    lengths = torch.tensor([100, 77])
    res = []
    for pad in (0, 12):
        eng = S2Engine(hps, gpu, torch.float32)
        for m in eng.net_g.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        fill_module(eng.net_g, 1)
        fill_module(eng.net_d, 2)
        eng.net_g.ref_enc.mask_beyond_collate = True     # what the trainer sets when its reader pads (a no-op at pad = 0)
        T = 102 + pad                                    # 102 = the reference collate's length for a longest item of 100
        wav = torch.zeros(2, 1, T * 640)
        wav[:, :, :100 * 640] = b["wav"]
        wav[1, :, 77 * 640:] = 0.0                       # the reference's collate zero-fills behind an item's end
        ssl = torch.zeros(2, 768, T)
        ssl[:, :, :100] = b["ssl"]
        ssl[1, :, 77:] = 0.0
        eps = torch.randn(2, 192, T, generator=torch.Generator().manual_seed(pad + 1))     # padding frames: any noise
        eps[:, :, :100] = b["eps"]
        spec = spectrogram_torch(wav.squeeze(1).to(gpu), 2048, 32000, 640, 2048)
        assert spec.size(2) == T
        ids = torch.tensor([10, 30])                     # both segments inside the live part of their item
        out = eng.step(ssl.to(gpu), spec, lengths.to(gpu), wav.to(gpu), b["text"].to(gpu), b["text_lengths"].to(gpu),
                       eps=eps.to(gpu), ids_slice=ids.to(gpu), do_opt=False)
        torch.cuda.synchronize()
        res.append((dict(disc=float(out.disc), gen=float(out.gen), fm=float(out.fm), mel=float(out.mel), kl=float(out.kl)),
                    eng.rt_g.arena.grad.detach().cpu().clone(), eng.rt_d.arena.grad.detach().cpu().clone(),
                    out.extras["y_hat"].detach().float().cpu().clone()))
        del eng
        torch.cuda.empty_cache()
    (l0, g0, d0, y0), (l1, g1, d1, y1) = res
    # same arithmetic, other tile boundaries (T = 102 vs 114): fp32 summation order differs -- measured 2.4e-5 on the
    # adversarial loss behind ninety layers, 3.9e-4 on the mel term (a log of a near-silent random-init waveform); a reduction
    # that saw the 12 padded frames of 112 would be off by per cent
    assert float((y0 - y1).abs().max()) <= 1e-3 * float(y0.abs().max())
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-3 * max(abs(l0[k]), 1e-3), (k, l0[k], l1[k])
    for a, c, what in ((g0, g1, "G"), (d0, d1, "D")):
        err = float((a - c).abs().max() / (a.abs().max() + 1e-12))
        assert err < 2e-3, (what, err)
