"""the feature-directory readers against golden outputs of the reference's own readers
(tests/golden/data_readers.{json,pt}, written by tests/golden/make_golden_data.py from /root/reference).

Integer work (filters, ids, lengths, sampler batches, collate layout) is compared exactly.  The spectrogram in the CPU
tests comes from the oracle's torch.stft restatement (test-only); tests/test_data_readers_gpu.py runs the same batches
through the HIP STFT."""
import importlib
import json
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import data_fixture as F  # noqa: E402

D = importlib.import_module("easevoice_trainer_amd.train.dataset")
CFG = dict(sampling_rate=F.SR, filter_length=F.NFFT, hop_length=F.HOP, win_length=F.NFFT)


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


def tstat(t):
    t = t.double()
    idx = torch.arange(t.numel(), dtype=torch.float64).reshape(t.shape).remainder(97.0)
    return [float(t.sum()), float(t.abs().sum()), float((t * idx).sum())]


def oracle_spec(y, n_fft, sr, hop, win, center=False):
    from oracle.s2_step import stft_mag
    return stft_mag(y, n_fft, hop)


def close(a, b, rel=1e-6):
    return all(abs(x - y) <= rel * max(1.0, abs(x), abs(y)) for x, y in zip(a, b))


def test_symbol_table_sources(feature_dir, gold, monkeypatch, tmp_path):
    assert D.load_symbol_table(feature_dir) == {s: i for i, s in enumerate(gold["symbols"])}
    alt = tmp_path / "alt.json"
    alt.write_text(json.dumps(["x", "y"]))
    monkeypatch.setenv("EVT_SYMBOLS_JSON", str(alt))
    assert D.load_symbol_table(feature_dir) == {"x": 0, "y": 1}
    monkeypatch.delenv("EVT_SYMBOLS_JSON")
    # no table anywhere and no reference checkout importable -> a loud error (other test modules may have put the
    # checkout on sys.path: take it off for this check)
    monkeypatch.setattr(sys, "path", [p for p in sys.path if not os.path.isdir(os.path.join(p, "src", "easevoice"))])
    if not any(k == "src" or k.startswith("src.") for k in sys.modules):
        with pytest.raises(FileNotFoundError):
            D.load_symbol_table(str(tmp_path))


def test_wav_reader(tmp_path):
    import struct
    import numpy as np
    pcm = np.array([[0, 100], [-32768, 32767], [5, -5]], dtype="<i2")
    body = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + 8 + 2 + len(body)) + b"WAVE" + b"LIST" + struct.pack("<I", 1) + b"x\0" + \
        b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, 32000, 128000, 4, 16) + b"data" + struct.pack("<I", len(body))
    p = tmp_path / "st.wav"
    p.write_bytes(hdr + body)
    x = D.read_wav_pcm16(str(p), 32000)
    assert x.dtype == np.float32 and x.tolist() == [50 / 32768, -0.5 / 32768, 0.0]
    with pytest.raises(ValueError):
        D.read_wav_pcm16(str(p), 16000)
    (tmp_path / "bad.wav").write_bytes(b"not a wave file at all")
    with pytest.raises(ValueError):
        D.read_wav_pcm16(str(tmp_path / "bad.wav"), 32000)


def test_s2_index_matches_reference(feature_dir, gold):
    ds = D.S2FeatureDir(feature_dir, CFG)
    g = gold["s2_dataset"]
    assert len(ds) == g["total"]
    seen = {}
    for (name, ids), length in zip(ds.items, ds.lengths):
        e = seen.setdefault(name, dict(ids=ids, length=length, count=0))
        e["count"] += 1
    assert seen == g["per_name"]
    assert ds.skipped_phone == 10 and ds.skipped_dur == 10
    # the same order in every process: names are sorted before the seeded shuffle
    assert [n for n, _ in D.S2FeatureDir(feature_dir, CFG).items] == [n for n, _ in ds.items]


def test_s2_items_match_reference(feature_dir, gold):
    ds = D.S2FeatureDir(feature_dir, CFG)
    first = {}
    for i, (name, _) in enumerate(ds.items):
        first.setdefault(name, i)
    assert sorted(first) == sorted(gold["s2_items"])
    for name, g in gold["s2_items"].items():
        ssl, wav, text, frames, ok = ds.load(first[name])
        assert list(ssl.shape) == g["ssl"] and str(ssl.dtype) == g["ssl_dtype"], name
        assert list(wav.shape) == g["wav"] and g["spec"] == [1025, frames], name
        assert text.tolist() == g["text"]
        assert tstat(ssl) == g["ssl_stat"] and tstat(wav) == g["wav_stat"], name
        assert torch.equal(ssl[0, :, -3:], gold["blobs"]["ssl_tail/" + name])
        assert ok == (name != F.BROKEN)
        if ok:
            spec = oracle_spec(wav, F.NFFT, F.SR, F.HOP, F.NFFT)[0]
            assert torch.allclose(spec[::64], gold["blobs"]["spec_rows/" + name], rtol=1e-4, atol=1e-4)
        else:
            assert g["spec_stat"] == [0.0, 0.0, 0.0] and g["wav_stat"] == [0.0, 0.0, 0.0]


def _collate_like_reader(ds, names, first):
    items = [ds.load(first[n]) for n in names]
    batch, order = D.collate_s2(items, 1025)
    spec = batch[2]
    for row, src in enumerate(order):
        if items[src][4]:
            fr = items[src][3]
            spec[row, :, :fr] = oracle_spec(items[src][1], F.NFFT, F.SR, F.HOP, F.NFFT)[0]
    return batch, order


def test_s2_collate_matches_reference(feature_dir, gold):
    ds = D.S2FeatureDir(feature_dir, CFG)
    first = {}
    for i, (name, _) in enumerate(ds.items):
        first.setdefault(name, i)
    keys = ["ssl", "ssl_len", "spec", "spec_len", "wav", "wav_len", "text", "text_len"]
    for g in gold["s2_collate"]:
        batch, _ = _collate_like_reader(ds, g["names"], first)
        for k, t in zip(keys, batch):
            assert list(t.shape) == g[k]["shape"] and str(t.dtype) == g[k]["dtype"], k
            if "values" in g[k]:
                assert t.tolist() == g[k]["values"], k
            elif k == "spec":
                assert close(tstat(t), g[k]["stat"], rel=2e-5), k
            else:
                assert tstat(t) == g[k]["stat"], k


def test_s2_reader_batches_match_reference_collate(feature_dir, gold):
    """the reader's own path (file dtypes on the host, conversion after the copy, spectrogram buffer created on the device)
    gives the reference collate's batch for the same items"""
    rd = D.S2Reader(feature_dir, CFG, batch_size=4, device="cpu", spec_fn=oracle_spec)
    first = {}
    for i, (name, _) in enumerate(rd.ds.items):
        first.setdefault(name, i)
    keys = ["ssl", "ssl_len", "spec", "spec_len", "wav", "wav_len", "text", "text_len"]
    for g in gold["s2_collate"]:
        rd.sampler = [[first[n] for n in g["names"]]]
        (batch,) = list(rd)
        for k, t in zip(keys, batch):
            assert list(t.shape) == g[k]["shape"] and str(t.dtype) == g[k]["dtype"], k
            if "values" in g[k]:
                assert t.tolist() == g[k]["values"], k
            elif k == "spec":
                assert close(tstat(t), g[k]["stat"], rel=2e-5), k
            else:
                assert tstat(t) == g[k]["stat"], k


def test_s2_bucket_sampler_matches_reference(gold):
    lengths = F.sampler_lengths()
    for g in gold["s2_sampler"]:
        smp = D.S2BucketSampler(lengths, g["batch_size"], None, num_replicas=g["world"], rank=g["rank"])
        smp.set_epoch(g["epoch"])
        assert len(smp) == g["n"] and smp.boundaries == g["boundaries"]
        assert list(iter(smp)) == g["batches"], (g["batch_size"], g["world"], g["rank"], g["epoch"])
    g = gold["s2_sampler_sparse"]
    smp = D.S2BucketSampler(g["lengths"], 2)
    smp.set_epoch(3)
    assert smp.boundaries == g["boundaries"] and list(iter(smp)) == g["batches"]


def test_s2_sampler_ranks_partition_each_bucket():
    lengths = F.sampler_lengths(700, seed=9)
    world, bs = 4, 8
    per_rank = []
    for r in range(world):
        s = D.S2BucketSampler(lengths, bs, None, num_replicas=world, rank=r)
        s.set_epoch(11)
        per_rank.append(list(iter(s)))
    assert len({len(b) for b in per_rank}) == 1           # every rank runs the same number of steps
    inside = {i for i, v in enumerate(lengths) if 32 < v <= 1900}
    seen = {i for b in per_rank for batch in b for i in batch}
    assert seen == inside                                  # every in-range item is visited, nothing else
    for batches in per_rank:
        for batch in batches:                              # a batch never mixes buckets
            lo = max(b for b in D.S2_BUCKET_BOUNDARIES if b < lengths[batch[0]])
            assert all(lo < lengths[i] <= lo + (268 if lo == 32 else 100) for i in batch)


def test_s2_reader_end_to_end_cpu(feature_dir):
    rd = D.S2Reader(feature_dir, CFG, batch_size=4, device="cpu", spec_fn=oracle_spec, prefetch=2)
    rd.set_epoch(1)
    n = 0
    for ssl, ssl_l, spec, spec_l, wav, wav_l, text, text_l in rd:
        n += 1
        assert ssl.shape[0] == 4 and spec.shape[1] == 1025 and spec.shape[2] % 2 == 0
        assert torch.all(spec_l[:-1] >= spec_l[1:])
        assert spec.shape[2] == 2 * (int(spec_l.max()) // 2 + 1)
        for i in range(4):
            assert not spec[i, :, int(spec_l[i]):].any() and not wav[i, :, int(wav_l[i]):].any()
    assert n == len(rd) > 0
    # leaving an epoch early must not leave the reader thread blocked
    it = iter(rd)
    next(it)
    it.close()


def test_s2_reader_shape_quantisation(feature_dir):
    """pad_frames: same items, lengths and content as the reference layout, time axes rounded up to a multiple"""
    a = D.S2Reader(feature_dir, CFG, batch_size=4, device="cpu", spec_fn=oracle_spec, pad_frames=0)
    b = D.S2Reader(feature_dir, CFG, batch_size=4, device="cpu", spec_fn=oracle_spec, pad_frames=64)
    a.set_epoch(2)
    b.set_epoch(2)
    shapes = set()
    for ba, bb in zip(a, b):
        for i in (1, 3, 5, 7):
            assert torch.equal(ba[i], bb[i])                    # lengths
        assert torch.equal(ba[6], bb[6])                        # text
        T = bb[2].shape[2]
        assert T % 64 == 0 and bb[0].shape[2] == T and bb[4].shape[2] % (64 * F.HOP) == 0 and T >= ba[2].shape[2]
        for x, y in ((ba[0], bb[0]), (ba[2], bb[2]), (ba[4], bb[4])):
            n = x.shape[-1]
            assert torch.equal(x, y[..., :n]) and not y[..., n:].any()
        shapes.add(T)
    assert shapes <= {64, 128, 192}


def test_s1_table_matches_reference(feature_dir, gold):
    tab = D.S1SemanticTable(os.path.join(feature_dir, "2-name2text.txt"), os.path.join(feature_dir, "6-name2semantic.tsv"))
    g = gold["s1_dataset"]
    assert tab.item_names == g["item_names"]
    assert [[list(s), list(p)] for s, p in tab.semantic_phoneme] == g["pairs"]
    assert (tab.num_not_in, tab.num_deleted_bigger, tab.num_deleted_ps) == (3, 1, 1)
    c = gold["s1_collate"]
    col = tab.collate([tab.load(i) for i in c["indices"]])
    assert col["ids"] == c["ids"]
    for k in ("phoneme_ids", "phoneme_ids_len", "semantic_ids", "semantic_ids_len"):
        assert col[k].tolist() == c[k] and col[k].dtype == torch.long, k
    assert list(col["bert_feature"].shape) == c["bert_shape"] and tstat(col["bert_feature"]) == c["bert_stat"]


class _Secs:
    def __init__(self, secs):
        self.secs = secs

    def __len__(self):
        return len(self.secs)

    def get_sample_length(self, i):
        return self.secs[i]


def test_s1_bucket_sampler_matches_reference(gold):
    secs = _Secs(F.s1_lengths())
    for g in gold["s1_sampler"]:
        smp = D.S1BucketSampler(secs, g["batch_size"], num_replicas=g["world"], rank=g["rank"])
        smp.set_epoch(g["epoch"])
        assert list(iter(smp)) == g["indices"], (g["batch_size"], g["world"], g["rank"], g["epoch"])
        flat = [i for b in smp.batches() for i in b]
        assert flat == g["indices"] and all(len(b) == g["batch_size"] for b in smp.batches()[:-1])
    with pytest.raises(ValueError):
        D.S1BucketSampler(secs, 4, num_replicas=2, rank=2)


def test_s1_reader_end_to_end_cpu(feature_dir):
    rd = D.S1Reader(feature_dir, dict(max_sec=100, pad_val=1024), batch_size=8, device="cpu")
    assert rd.batch_size == 8
    rd.set_epoch(2)
    seen = 0
    for b in rd:
        B = b["phoneme_ids"].shape[0]
        seen += B
        assert b["bert_feature"].shape == (B, 1024, b["phoneme_ids"].shape[1])
        assert int(b["semantic_ids_len"].max()) == b["semantic_ids"].shape[1]
        for i in range(B):
            assert torch.all(b["semantic_ids"][i, int(b["semantic_ids_len"][i]):] == 1024)
    assert seen == len(rd.table)


def test_open_source_picks_the_feature_directory(feature_dir, monkeypatch):
    from easevoice_trainer_amd.train import data as data_mod
    monkeypatch.delenv("EVT_SYNTHETIC_STEPS", raising=False)
    src = data_mod.open_source("s1", feature_dir, "cpu", lambda n: None, batch_size=8, cfg=dict(max_sec=100, pad_val=1024))
    assert isinstance(src, D.S1Reader)
    with pytest.raises(FileNotFoundError):
        data_mod.open_source("s2", os.path.join(feature_dir, "missing"), "cpu", lambda n: None, batch_size=4, cfg=CFG)


def test_semantic_tsv_matches_reference_token_step(feature_dir, gold, tmp_path):
    """6-name2semantic.tsv written from the 4-cnhubert files == the reference's `token` step with the same weights"""
    from cpu_emu import cpu_emulation
    from util_fill import fill_module
    from easevoice_trainer_amd.inference.semantic import write_semantic_tsv
    from easevoice_trainer_amd.module import models

    hps = json.load(open(os.path.join(os.path.dirname(HERE), "configs", "s2.json")))
    g = gold["semantic_tsv"]
    with cpu_emulation():
        net = models.SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
        fill_module(net, 1)
        net.eval()
        out = str(tmp_path / "6-name2semantic.tsv")
        n = write_semantic_tsv(net.extract_latent, g["names"], os.path.join(feature_dir, "4-cnhubert"), out)
    assert open(out, encoding="utf8").read() == g["text"] and n == g["text"].count("\n") - 1
    # and the file is what the s1 reader parses
    tab = D.S1SemanticTable(os.path.join(feature_dir, "2-name2text.txt"), out, symbol_to_id={s: i for i, s in enumerate(gold["symbols"])})
    assert len(tab) >= 1


def test_wav_reader_against_stdlib_wave_random_files(tmp_path):
    """random RIFF layouts (extra chunks of odd and even size before fmt / data, truncated data chunk) decode to the same
    samples the stdlib `wave` module reads"""
    import random
    import struct
    import wave
    import numpy as np

    rng = random.Random(13)
    nrng = np.random.RandomState(13)
    for case in range(40):
        ch = rng.choice([1, 1, 1, 2])
        n = rng.randint(0, 3000)
        pcm = nrng.randint(-32768, 32768, size=(n, ch)).astype("<i2")
        body = pcm.tobytes()

        def chunk(cid, payload):
            return cid + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")

        parts = []
        for _ in range(rng.randint(0, 2)):
            parts.append(chunk(rng.choice([b"LIST", b"bext", b"junk"]), bytes(rng.randint(0, 255) for _ in range(rng.randint(0, 9)))))
        parts.append(chunk(b"fmt ", struct.pack("<HHIIHH", 1, ch, 32000, 32000 * 2 * ch, 2 * ch, 16)))
        if rng.random() < 0.5:
            parts.append(chunk(b"fact", struct.pack("<I", n)))
        parts.append(chunk(b"data", body))
        blob = b"".join(parts)
        p = tmp_path / f"r{case}.wav"
        p.write_bytes(b"RIFF" + struct.pack("<I", 4 + len(blob)) + b"WAVE" + blob)
        with wave.open(str(p), "rb") as w:
            ref = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, ch)
        want = (ref.astype(np.float32) / np.float32(32768.0)).mean(axis=1, dtype=np.float32) if ch > 1 else \
            ref[:, 0].astype(np.float32) / np.float32(32768.0)
        got = D.read_wav_pcm16(str(p), 32000)
        assert got.dtype == np.float32 and np.array_equal(got, want), case
        if ch == 1:
            raw = D.read_wav_pcm16(str(p), 32000, raw=True)
            assert raw.dtype == np.dtype("<i2") and np.array_equal(raw, ref[:, 0])


def test_collate_mixed_sample_dtypes():
    """a batch mixing raw int16 items with a float item (non-mono file) is scaled on the host, not copied as integers"""
    a = (torch.zeros(1, 8, 40, dtype=torch.float16), torch.tensor([[16384, -32768, 0, 1] * 6400], dtype=torch.int16),
         torch.tensor([1.0, 2.0]), 40, True)
    b = (torch.zeros(1, 8, 35, dtype=torch.float16), torch.full((1, 22400), 0.25), torch.tensor([3.0]), 35, True)
    (ssl, _, shape, _, wav, wav_l, _, _), order = D.collate_s2([a, b], 5, with_spec=False)
    assert wav.dtype == torch.float32 and ssl.dtype == torch.float16 and order == [0, 1]
    assert wav[0, 0, :4].tolist() == [0.5, -1.0, 0.0, 1.0 / 32768.0] and float(wav[1, 0, 0]) == 0.25
    (_, _, _, _, wav2, _, _, _), _ = D.collate_s2([a, a], 5, with_spec=False)
    assert wav2.dtype == torch.int16


def test_s2_reader_worker_processes_same_batches(feature_dir):
    """S2Reader(loader_workers=2): reader PROCESSES (the reference's DataLoader workers, sovits.py:258-267) deliver the very
    batches of the one-thread reader, in the same order, over two epochs; an epoch left early leaks nothing into the next"""
    a = D.S2Reader(feature_dir, CFG, batch_size=4, device="cpu", spec_fn=oracle_spec)
    b = D.S2Reader(feature_dir, CFG, batch_size=4, device="cpu", spec_fn=oracle_spec, loader_workers=2, prefetch=3)
    try:
        for epoch in (1, 2):
            a.set_epoch(epoch)
            b.set_epoch(epoch)
            n = 0
            for ba, bb in zip(a, b):
                n += 1
                for x, y in zip(ba, bb):
                    assert x.dtype == y.dtype and torch.equal(x, y)
            assert n == len(a) == len(b) > 0
        b.set_epoch(3)
        it = iter(b)
        first = next(it)
        it.close()                                   # leave the epoch after one batch
        a.set_epoch(4)
        b.set_epoch(4)
        for ba, bb in zip(a, b):
            for x, y in zip(ba, bb):
                assert torch.equal(x, y)
        assert first[0].shape[0] == 4
    finally:
        b.close()
