"""Golden outputs of the REFERENCE's dataset readers, samplers and collate functions, produced by
importing them read-only from /root/reference (oracle/refshim.py) and running them over the seeded feature directory of
tests/data_fixture.py.

Run in the build container only:   python tests/golden/make_golden_data.py
Writes tests/golden/data_readers.json and data_readers.pt.

Two stand-ins, both outside the code under test: ffmpeg is absent, so `load_audio` is replaced by a stdlib `wave` read
scaled by 1/32768 (what ffmpeg's s16 -> flt conversion produces for the mono 32 kHz files of 5-wav32k; a file it cannot
decode gives the empty array the reference's own error path returns); the phoneme table is the reference's
SYMBOLS, whose first 64 entries are stored in the fixture as the test table."""
import json
import os
import sys
import tempfile
import wave

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import refshim  # noqa: E402

refshim.install()
import data_fixture as F  # noqa: E402


def wave_load_audio(file, sr):
    try:
        with wave.open(file, "rb") as w:
            assert w.getframerate() == sr and w.getsampwidth() == 2 and w.getnchannels() == 1
            pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        return pcm.astype(np.float32) / np.float32(32768.0)
    except Exception:
        return np.array([])


class Hp(dict):
    __getattr__ = dict.__getitem__


class Stub(torch.utils.data.Dataset):
    def __init__(self, lengths):
        self.lengths = lengths

    def __len__(self):
        return len(self.lengths)

    def get_sample_length(self, i):
        return self.lengths[i]


def tstat(t):
    t = t.double()
    return [float(t.sum()), float(t.abs().sum()), float((t * torch.arange(t.numel(), dtype=torch.float64).reshape(t.shape)
                                                         .remainder(97.0)).sum())]


def main():
    from src.easevoice.module import data_utils as DU
    from src.easevoice.soundstorm.auto_reg.data import bucket_sampler as BS
    from src.easevoice.soundstorm.auto_reg.data import dataset as DS
    from src.easevoice.text.symbols import SYMBOLS

    DU.load_audio = wave_load_audio
    symbols = list(SYMBOLS[:64])
    out, blobs = {"symbols": symbols}, {}
    with tempfile.TemporaryDirectory() as root:
        F.build_feature_dir(root, symbols)
        hp = Hp(exp_dir=root, max_wav_value=32768.0, sampling_rate=F.SR, filter_length=F.NFFT, hop_length=F.HOP,
                win_length=F.NFFT)
        ds = DU.TextAudioSpeakerLoader(hp)
        per_name = {}
        for (name, ids), length in zip(ds.audiopaths_sid_text, ds.lengths):
            e = per_name.setdefault(name, dict(ids=ids, length=length, count=0))
            assert e["ids"] == ids and e["length"] == length
            e["count"] += 1
        out["s2_dataset"] = dict(total=len(ds), per_name=per_name)
        # one item per distinct name, in sorted-name order
        first = {}
        for i, (name, _) in enumerate(ds.audiopaths_sid_text):
            first.setdefault(name, i)
        names = sorted(first)
        items = {}
        for name in names:
            ssl, spec, wav, text = ds[first[name]]
            items[name] = (ssl, spec, wav, text)
            blobs["spec_rows/" + name] = spec[::64].clone()          # 17 of the 1025 bins, all frames
            blobs["ssl_tail/" + name] = ssl[0, :, -3:].clone()
        out["s2_items"] = {n: dict(ssl=list(it[0].shape), ssl_dtype=str(it[0].dtype), spec=list(it[1].shape),
                                   wav=list(it[2].shape), text=it[3].tolist(), ssl_stat=tstat(it[0]),
                                   spec_stat=tstat(it[1]), wav_stat=tstat(it[2])) for n, it in items.items()}
        # collate: two batches in a fixed (unsorted) order, one containing the placeholder item
        out["s2_collate"] = []
        for case in (names[:4], names[3:][::-1]):
            res = DU.TextAudioSpeakerCollate()([items[n] for n in case])
            keys = ["ssl", "ssl_len", "spec", "spec_len", "wav", "wav_len", "text", "text_len"]
            rec = dict(names=case)
            for k, t in zip(keys, res):
                rec[k] = dict(shape=list(t.shape), dtype=str(t.dtype), stat=tstat(t))
                if t.dim() == 1 or k == "text":
                    rec[k]["values"] = t.tolist()
            out["s2_collate"].append(rec)

        # the `token` step (normalize.py:181-211) over the same 4-cnhubert files, with the reference's model
        from src.easevoice.module import models as RM
        from util_fill import fill_module
        cfg = json.load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "s2.json")))
        vq = RM.SynthesizerTrn(1025, 32, n_speakers=300, **cfg["model"])
        fill_module(vq, 1)
        vq.eval()
        tsv = ["item_name\tsemantic_audio"]
        token_names = [it[0] for it in F.ITEMS] + ["missing.wav"]
        with torch.no_grad():
            for name in token_names:
                hp_ = os.path.join(root, "4-cnhubert", name + ".pt")
                if not os.path.exists(hp_):
                    continue
                codes = vq.extract_latent(torch.load(hp_, map_location="cpu").float())
                tsv.append("%s\t%s" % (name, " ".join(str(i) for i in codes[0, 0, :].tolist())))
        out["semantic_tsv"] = dict(names=token_names, text="\n".join(tsv) + "\n")

        # s1 table
        sem = DS.Text2SemanticDataset(phoneme_path=os.path.join(root, "2-name2text.txt"),
                                      semantic_path=os.path.join(root, "6-name2semantic.tsv"), max_sec=100, pad_val=1024)
        out["s1_dataset"] = dict(item_names=list(sem.item_names),
                                 pairs=[[list(map(int, s)), list(map(int, p))] for s, p in sem.semantic_phoneme])
        ex = [sem[i] for i in (0, 3, 5, 1)]
        col = sem.collate(ex)
        out["s1_collate"] = dict(indices=[0, 3, 5, 1], ids=col["ids"], phoneme_ids=col["phoneme_ids"].tolist(),
                                 phoneme_ids_len=col["phoneme_ids_len"].tolist(),
                                 semantic_ids=col["semantic_ids"].tolist(),
                                 semantic_ids_len=col["semantic_ids_len"].tolist(),
                                 bert_shape=list(col["bert_feature"].shape), bert_stat=tstat(col["bert_feature"]))

    # bucket samplers over stub datasets (only `lengths` / `get_sample_length` are read)
    lengths = F.sampler_lengths()
    out["s2_sampler"] = []
    for bs, world in ((4, 1), (6, 2), (16, 3)):
        for rank in range(world):
            smp = DU.DistributedBucketSampler(Stub(lengths), bs, [32] + list(range(300, 2000, 100)),
                                              num_replicas=world, rank=rank, shuffle=True)
            for epoch in (1, 2, 7):
                smp.set_epoch(epoch)
                out["s2_sampler"].append(dict(batch_size=bs, world=world, rank=rank, epoch=epoch, n=len(smp),
                                              boundaries=list(smp.boundaries), batches=list(iter(smp))))
    # sparse lengths: empty buckets are removed together with their upper boundary
    sparse = [40, 45, 350, 360, 365, 1250, 1850, 1851, 1852, 5000, 10]
    smp = DU.DistributedBucketSampler(Stub(sparse), 2, [32] + list(range(300, 2000, 100)), num_replicas=1, rank=0)
    smp.set_epoch(3)
    out["s2_sampler_sparse"] = dict(lengths=sparse, boundaries=list(smp.boundaries), batches=list(iter(smp)))

    secs = F.s1_lengths()
    out["s1_sampler"] = []
    for bs, world in ((8, 1), (5, 2), (12, 4)):
        for rank in range(world):
            smp = BS.DistributedBucketSampler(Stub(secs), num_replicas=world, rank=rank, batch_size=bs)
            for epoch in (0, 1, 5):
                smp.set_epoch(epoch)
                out["s1_sampler"].append(dict(batch_size=bs, world=world, rank=rank, epoch=epoch,
                                              indices=list(iter(smp))))
    with open(os.path.join(HERE, "data_readers.json"), "w") as f:
        json.dump(out, f)
    torch.save(blobs, os.path.join(HERE, "data_readers.pt"))
    print("wrote", len(json.dumps(out)), "json bytes,", sum(v.numel() for v in blobs.values()) * 4, "blob bytes")


if __name__ == "__main__":
    main()
