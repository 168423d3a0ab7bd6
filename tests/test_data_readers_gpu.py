"""GPU: the feature-directory readers with the HIP STFT in the loop, against the golden outputs of the
reference's readers.  (The trainer loops over such a directory: tests/test_zz_readers_train_gpu.py.)"""
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


def test_item_spectrograms_match_reference(gpu, feature_dir, gold):
    from easevoice_trainer_amd.module.mel_processing import spectrogram_torch

    ds = D.S2FeatureDir(feature_dir, CFG)
    first = {}
    for i, (name, _) in enumerate(ds.items):
        first.setdefault(name, i)
    for name, g in gold["s2_items"].items():
        ssl, wav, text, frames, ok = ds.load(first[name])
        if not ok:
            continue
        spec = spectrogram_torch(wav.cuda(), F.NFFT, F.SR, F.HOP, F.NFFT, center=False)[0].cpu()
        assert list(spec.shape) == g["spec"]
        ref = gold["blobs"]["spec_rows/" + name]
        # fp32 DFT of 2048 points against torch.stft's FFT: absolute error scales with the frame's energy
        assert torch.allclose(spec[::64], ref, rtol=2e-3, atol=2e-3 * float(ref.max())), name
        assert close(tstat(spec)[:2], g["spec_stat"][:2], rel=1e-3), name


def test_reader_batches_match_reference_collate(gpu, feature_dir, gold):
    """the reader's device batch == the reference collate's output for the same items (spectrogram within fp32 DFT
    tolerance, everything else exact), including the all-zero placeholder row of an undecodable file"""
    rd = D.S2Reader(feature_dir, CFG, batch_size=4, device="cuda")
    first = {}
    for i, (name, _) in enumerate(rd.ds.items):
        first.setdefault(name, i)
    keys = ["ssl", "ssl_len", "spec", "spec_len", "wav", "wav_len", "text", "text_len"]
    for g in gold["s2_collate"]:
        rd.sampler = [[first[n] for n in g["names"]]]       # one hand-picked batch
        (batch,) = list(rd)
        for k, t in zip(keys, batch):
            assert t.is_cuda and list(t.shape) == g[k]["shape"] and str(t.dtype) == g[k]["dtype"], k
            t = t.cpu()
            if "values" in g[k]:
                assert t.tolist() == g[k]["values"], k
            elif k == "spec":
                assert close(tstat(t)[:2], g[k]["stat"][:2], rel=1e-3), k
            else:
                assert tstat(t) == g[k]["stat"], k
        if F.BROKEN in g["names"]:
            row = [i for i in range(4) if int(batch[5][i]) == 100 * F.HOP and not batch[4][i].any()]
            assert len(row) == 1 and not batch[2][row[0]].any()
