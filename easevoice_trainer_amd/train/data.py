"""Batch sources for the trainers.

The trainers take any iterable of batches with the reference's collate layout.  Three sources ship: the readers of the
reference's feature directories (dataset.py), fixed-shape synthetic batches and a
tensor bundle (`torch.save` of a list of batch tuples) for pre-collated features."""
import os

import torch


class SyntheticS2Batches:
    """(ssl, ssl_lengths, spec, spec_lengths, y, y_lengths, text, text_lengths) of data_utils.py:167-226"""

    def __init__(self, batch_size, seconds, steps_per_epoch, device, seed=1234, t_text=60, rank=0, world=1):
        self.B, self.T, self.steps, self.device = batch_size, int(seconds * 50), steps_per_epoch, device
        self.seed, self.t_text, self.rank, self.world = seed, t_text, rank, world
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.steps

    def __iter__(self):
        from ..module.mel_processing import spectrogram_torch

        g = torch.Generator().manual_seed(self.seed + 1000 * self.epoch + self.rank)
        for _ in range(self.steps):
            wav = (torch.rand(self.B, 1, self.T * 640, generator=g) - 0.5).to(self.device)
            ssl = torch.randn(self.B, 768, self.T, generator=g).to(self.device)
            text = torch.randint(0, 732, (self.B, self.t_text), generator=g).to(self.device)
            lens = torch.full((self.B,), self.T, dtype=torch.long, device=self.device)
            tl = torch.full((self.B,), self.t_text, dtype=torch.long, device=self.device)
            spec = spectrogram_torch(wav.squeeze(1), 2048, 32000, 640, 2048)
            yield ssl, lens, spec, lens, wav, lens * 640, text, tl


class SyntheticS1Batches:
    """dict batches of soundstorm/auto_reg/data/dataset.py:234-271"""

    def __init__(self, batch_size, x_len, y_len, steps_per_epoch, device, seed=1234, rank=0):
        self.B, self.x_len, self.y_len, self.steps, self.device, self.seed, self.rank = \
            batch_size, x_len, y_len, steps_per_epoch, device, seed, rank
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.steps

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + 1000 * self.epoch + self.rank)
        d = self.device
        for _ in range(self.steps):
            yield dict(phoneme_ids=torch.randint(0, 732, (self.B, self.x_len), generator=g).to(d),
                       phoneme_ids_len=torch.full((self.B,), self.x_len, dtype=torch.long, device=d),
                       semantic_ids=torch.randint(0, 1024, (self.B, self.y_len), generator=g).to(d),
                       semantic_ids_len=torch.full((self.B,), self.y_len, dtype=torch.long, device=d),
                       bert_feature=torch.randn(self.B, 1024, self.x_len, generator=g).to(d))


class TensorBundle:
    """a `torch.save`d list of batches (tuples for s2, dicts for s1) with the reference's collate layout"""

    def __init__(self, path, device):
        self.batches = torch.load(path, map_location="cpu", weights_only=False)
        self.device = device

    def set_epoch(self, epoch):
        pass

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for b in self.batches:
            if isinstance(b, dict):
                yield {k: v.to(self.device) for k, v in b.items()}
            else:
                yield tuple(t.to(self.device) for t in b)


def open_source(kind, train_input_dir, device, synthetic_factory, batch_size=None, cfg=None, rank=0, world=1):
    """Batch source of a trainer, in this order: a tensor bundle `evt_<kind>_batches.pt`, fixed-shape synthetic batches
    when EVT_SYNTHETIC_STEPS is set, else the reference's feature directory (dataset.py: 2-name2text.txt with
    4-cnhubert + 5-wav32k for s2, with 6-name2semantic.tsv + 3-bert for s1)."""
    bundle = os.path.join(train_input_dir or "", f"evt_{kind}_batches.pt")
    if train_input_dir and os.path.isfile(bundle):
        return TensorBundle(bundle, device)
    if os.environ.get("EVT_SYNTHETIC_STEPS"):
        return synthetic_factory(int(os.environ["EVT_SYNTHETIC_STEPS"]))
    marker = os.path.join(train_input_dir or "", "5-wav32k" if kind == "s2" else "6-name2semantic.tsv")
    if train_input_dir and os.path.exists(marker):
        from .dataset import S1Reader, S2Reader

        if kind == "s2":
            # the trainer rounds the padded time axes of a batch up to 16 frames unless told otherwise (EVT_PAD_FRAMES; 0 =
            # the reference's exact layout): every consumer masks by the lengths, so the step computes the same thing, and a
            # run then repeats a couple of dozen batch shapes instead of > 100, which is what lets the per-shape HIP-graph
            # replay of the s2 step engage.  Measured (profiles/r04_realdata_pad*.json, 2-10 s clips, B = 16, 240 steps):
            # 0 -> 1060, 8 -> 1370, 16 -> 1633, 32 -> 1512 audio-s/s (eager share 90 % / 22 % / 18 % / 16 %).
            # INVARIANTS this rests on (tests/test_zz_readers_train_gpu.py::test_padded_time_axis_changes_nothing pins them):
            # (1) no reduction or convolution over the time axis sees an unmasked padded frame -- losses, statistics,
            # attention keys.  ONE consumer of the reference does: the style encoder's temporal convolutions run over
            # spectral(zero frames) up to the collate's tensor length (modules.py:748-756); the trainer therefore switches
            # MelStyleEncoder.mask_beyond_collate on when it pads, which makes frames beyond the reference's own tensor
            # length contribute zeros there (found by that test: without it the waveform moved by more than 1e-3);
            # (2) the quantiser is frozen (its commitment loss, an unmasked mean over T, is the constant 0:
            # core_vq.py:311-316); (3) the two random draws are per frame, so a seeded run with padding draws different noise
            # for the SAME frames than an unpadded one -- equal in distribution, not bit for bit.
            pad = int(os.environ.get("EVT_PAD_FRAMES", "16")) if str(device).startswith("cuda") else None
            # reader processes, as the reference's DataLoader(num_workers=6, persistent_workers=True, prefetch_factor=4)
            # (sovits.py:258-267): on a GPU the one-thread reader was what set the real-data rate (profiles/
            # r06_reader_processes.txt: 1591 -> 2848 audio-s/s over 400 steps, 1825 -> 3563 once the shapes are captured).
            # EVT_READER_WORKERS overrides (0 = the prefetch thread); CPU runs keep the thread.
            workers = None
            if str(device).startswith("cuda") and "EVT_READER_WORKERS" not in os.environ:
                ncpu = os.cpu_count() or 1
                workers = 6 if ncpu >= 12 else max(0, min(4, ncpu // 2))
            return S2Reader(train_input_dir, cfg, batch_size, device, rank=rank, world=world, pad_frames=pad,
                            loader_workers=workers)
        return S1Reader(train_input_dir, cfg, batch_size, device, rank=rank, world=world)
    raise FileNotFoundError(
        f"no training input under {train_input_dir!r}: expected the reference's feature directory ({marker}), a tensor "
        f"bundle {bundle}, or EVT_SYNTHETIC_STEPS=<n> for fixed-shape synthetic batches")
