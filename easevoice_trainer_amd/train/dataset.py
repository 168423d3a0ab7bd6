"""Readers of the reference's on-disk training features: the two callers that feed the hot path.

  s2: <exp_dir>/2-name2text.txt + 4-cnhubert/<name>.pt + 5-wav32k/<name>   src/easevoice/module/data_utils.py:14-323
  s1: <exp_dir>/2-name2text.txt + 6-name2semantic.tsv + 3-bert/<name>.pt  soundstorm/auto_reg/data/dataset.py:38-271,
                                                                           bucket_sampler.py:29-170, data_module.py:42-54

What is kept bit-for-bit: the item filters, the <100-item repetition rule, the seeded shuffles, the bucket samplers'
batch composition for every (epoch, rank, world), the collate layouts and padding rules, the placeholder item on a read
failure.  What is laid out differently: files are read and collated by a host thread into pinned buffers a few batches
ahead of the step, and the per-item linear spectrogram runs on the GPU through the HIP STFT (`spectrogram_torch`) after
the copy instead of in DataLoader workers -- the 5-wav32k files are 16-bit PCM at the training rate, so the reference's
ffmpeg subprocess (src/utils/audio/__init__.py:13-32) reduces to a RIFF parse and a scale by 1/32768.

The phoneme table is data of the reference's text front-end (src/easevoice/text/symbols.py) and is not restated here:
it is read from `symbols.json` (`<exp_dir>/symbols.json` or `$EVT_SYMBOLS_JSON`, written by tools/dump_symbols.py); the
product never imports the reference."""
import json
import math
import os
import queue
import random
import struct
import threading
import traceback

import numpy as np
import torch

S2_BUCKET_BOUNDARIES = [32] + list(range(300, 2000, 100))  # src/train/sovits.py:233-253


# ---------------------------------------------------------------------------------------------------------------------
# phoneme table
# ---------------------------------------------------------------------------------------------------------------------
def load_symbol_table(exp_dir=None):
    """symbol -> id map of `cleaned_text_to_sequence` (src/easevoice/text/__init__.py:4-13).

    Looked up in: $EVT_SYMBOLS_JSON, then <exp_dir>/symbols.json (a JSON list, index == id).  The table is data of the
    reference's text front-end; `tools/dump_symbols.py <reference checkout> <exp_dir>/symbols.json` writes it once.  The
    product never imports the reference."""
    for path in (os.environ.get("EVT_SYMBOLS_JSON"), os.path.join(exp_dir, "symbols.json") if exp_dir else None):
        if path and os.path.isfile(path):
            with open(path, "r", encoding="utf8") as f:
                symbols = json.load(f)
            return {s: i for i, s in enumerate(symbols)}
    raise FileNotFoundError(
        "phoneme table not found: set EVT_SYMBOLS_JSON or put symbols.json (tools/dump_symbols.py) next to the feature "
        "directories")


def read_name2text(path):
    """{name: [phones, word2ph, text]} from 2-name2text.txt; lines without exactly 4 tab fields are dropped
    (data_utils.py:31-39, dataset.py:67-75)."""
    with open(path, "r", encoding="utf8") as f:
        lines = f.read().strip("\n").split("\n")
    table = {}
    for line in lines:
        tmp = line.split("\t")
        if len(tmp) != 4:
            continue
        table[tmp[0]] = [tmp[1], tmp[2], tmp[3]]
    return table


# ---------------------------------------------------------------------------------------------------------------------
# 5-wav32k reader
# ---------------------------------------------------------------------------------------------------------------------
def read_wav_pcm16(path, sampling_rate, raw=False):
    """float32 mono samples in [-1, 1) of a RIFF/WAVE PCM16 file whose rate is already `sampling_rate` (raw=True: the int16
    samples themselves for mono files -- the caller scales them by 1/32768 later, e.g. on the GPU; exact either way).

    Equals what load_audio (src/utils/audio/__init__.py:13-32: ffmpeg -> f32le, ac=1, ar=sr) returns for the files the
    reference's normalisation step writes into 5-wav32k: no resampling happens there and s16 -> f32 is x/32768.  Other
    encodings or rates raise (the caller turns that into the reference's placeholder item)."""
    with open(path, "rb") as f:
        blob = f.read()
    if len(blob) < 12 or blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(blob):
        cid, size = blob[pos:pos + 4], struct.unpack_from("<I", blob, pos + 4)[0]
        body = blob[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", body, 0)
        elif cid == b"data":
            data = body
            break
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, channels, rate, _, _, bits = fmt
    if tag not in (1, 0xFFFE) or bits != 16:
        raise ValueError(f"{path}: only 16-bit PCM is supported (format {tag}, {bits} bits)")
    if rate != sampling_rate:
        raise ValueError(f"{path}: sample rate {rate} != {sampling_rate} (5-wav32k is written at the training rate)")
    pcm = np.frombuffer(data[:len(data) // (2 * channels) * 2 * channels], dtype="<i2")
    if raw and channels == 1:
        return pcm
    x = pcm.astype(np.float32) * np.float32(1.0 / 32768.0)
    if channels > 1:
        x = x.reshape(-1, channels).mean(axis=1, dtype=np.float32)
    return x


def spec_frames(wav_len, n_fft, hop):
    """frame count of spectrogram_torch(center=False) with its (n_fft-hop)/2 reflect pad, mel_processing.py:50-66"""
    return (wav_len + 2 * ((n_fft - hop) // 2) - n_fft) // hop + 1


# ---------------------------------------------------------------------------------------------------------------------
# s2: TextAudioSpeakerLoader / TextAudioSpeakerCollate / DistributedBucketSampler
# ---------------------------------------------------------------------------------------------------------------------
class S2FeatureDir:
    """Item index of an s2 feature directory, data_utils.py:21-99.

    The reference lists the usable names as `list(set & set & set)`, whose order changes with the interpreter's string
    hash seed; here the names are sorted first, so the seeded shuffle that follows gives one order on every rank and
    every run.  `items` = [name, phoneme_ids], `lengths` = file_size // (2*hop) (header bytes included, as there)."""

    def __init__(self, exp_dir, data_cfg, val=False, symbol_to_id=None):
        self.path2 = "%s/2-name2text.txt" % exp_dir
        self.path4 = "%s/4-cnhubert" % exp_dir
        self.path5 = "%s/5-wav32k" % exp_dir
        for p in (self.path2, self.path4, self.path5):
            if not os.path.exists(p):
                raise FileNotFoundError(p)
        self.sampling_rate, self.filter_length = data_cfg["sampling_rate"], data_cfg["filter_length"]
        self.hop_length, self.win_length = data_cfg["hop_length"], data_cfg["win_length"]
        self.val = val
        sym = symbol_to_id if symbol_to_id is not None else load_symbol_table(exp_dir)
        names4 = {n[:-3] for n in os.listdir(self.path4)}
        names5 = set(os.listdir(self.path5))
        phoneme_data = read_name2text(self.path2)
        names = sorted(set(phoneme_data) & names4 & names5)
        if len(names) == 0:
            raise ValueError(f"data in {exp_dir} is all skipped, please check the data")
        if len(names) < 100:
            names = names * max(2, int(100 / len(names)))
        random.Random(1234).shuffle(names)
        self.items, self.lengths = [], []
        self.skipped_phone = self.skipped_dur = 0
        for name in names:
            try:
                ids = [sym[p] for p in phoneme_data[name][0].split(" ")]
            except KeyError:
                self.skipped_phone += 1
                continue
            size = os.path.getsize("%s/%s" % (self.path5, name))
            duration = size / self.sampling_rate / 2
            if duration == 0:
                self.skipped_dur += 1
                continue
            if 54 > duration > 0.6 or val:
                self.items.append([name, ids])
                self.lengths.append(size // (2 * self.hop_length))
            else:
                self.skipped_dur += 1
        if len(self.items) <= 1:
            raise ValueError(f"data in {exp_dir} is all skipped, please check the data")

    def __len__(self):
        return len(self.items)

    def load(self, index, raw=False):
        """(ssl [1,768,T'], wav [1,L], text float [n], frames, ok) of one item, data_utils.py:101-121 minus the
        spectrogram (computed on the GPU by S2Reader); a failed read gives the reference's all-zero placeholder item
        (ok=False: its spectrogram stays exactly zero, as there).  raw=True keeps the file dtypes (int16 samples, the
        hubert tensor as stored): the reader converts after the copy to the device, with the same values."""
        name, ids = self.items[index]
        text, ok = torch.tensor(ids, dtype=torch.float32), True
        try:
            wav = torch.from_numpy(np.array(read_wav_pcm16("%s/%s" % (self.path5, name), self.sampling_rate, raw=raw))
                                   ).unsqueeze(0)
            pad = (self.filter_length - self.hop_length) // 2
            if wav.size(1) <= pad:
                raise ValueError(f"{name}: {wav.size(1)} samples cannot be reflect-padded by {pad}")
            frames = spec_frames(wav.size(1), self.filter_length, self.hop_length)
            ssl = torch.load("%s/%s.pt" % (self.path4, name), map_location="cpu")
            if ssl.shape[-1] != frames:
                ssl = torch.cat([ssl, ssl[..., -1:]], dim=-1) if raw else \
                    torch.nn.functional.pad(ssl.float(), (0, 1), mode="replicate").to(ssl.dtype)
            ssl.requires_grad = False
        except Exception:
            traceback.print_exc()
            frames = 100
            wav = torch.zeros(1, 100 * self.hop_length)
            ssl = torch.zeros(1, 768, 100)
            text, ok = text[-1:], False
            print("load audio or ssl error!!!!!!", name)
        return ssl, wav, text, frames, ok


class S2BucketSampler:
    """Length-bucketed batches of one rank, data_utils.py:229-323 (same batches for the same lengths, boundaries,
    batch size, epoch, rank and replica count; items outside (boundaries[0], boundaries[-1]] are dropped, each bucket
    is padded by repetition to a multiple of world*batch_size)."""

    def __init__(self, lengths, batch_size, boundaries=None, num_replicas=1, rank=0, shuffle=True):
        self.lengths, self.batch_size = list(lengths), batch_size
        self.boundaries = list(S2_BUCKET_BOUNDARIES if boundaries is None else boundaries)
        self.num_replicas, self.rank, self.shuffle, self.epoch = num_replicas, rank, shuffle, 0
        buckets = [[] for _ in range(len(self.boundaries) - 1)]
        for i, length in enumerate(self.lengths):
            b = self._bucket_of(length)
            if b != -1:
                buckets[b].append(i)
        keep = [i for i, b in enumerate(buckets) if len(b) > 0]
        self.boundaries = [self.boundaries[0]] + [self.boundaries[i + 1] for i in keep] if keep else self.boundaries[:1]
        self.buckets = [buckets[i] for i in keep]
        total = self.num_replicas * self.batch_size
        self.num_samples_per_bucket = [len(b) + (total - len(b) % total) % total for b in self.buckets]
        self.total_size = sum(self.num_samples_per_bucket)
        self.num_samples = self.total_size // self.num_replicas

    def _bucket_of(self, x):
        # the reference bisects (data_utils.py:304-318); with increasing boundaries that is the unique i with
        # boundaries[i] < x <= boundaries[i+1]
        for i in range(len(self.boundaries) - 1):
            if self.boundaries[i] < x <= self.boundaries[i + 1]:
                return i
        return -1

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples // self.batch_size

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        if self.shuffle:
            orders = [torch.randperm(len(b), generator=g).tolist() for b in self.buckets]
        else:
            orders = [list(range(len(b))) for b in self.buckets]
        batches = []
        for bucket, ids, padded in zip(self.buckets, orders, self.num_samples_per_bucket):
            rem = padded - len(bucket)
            ids = ids + ids * (rem // len(bucket)) + ids[:rem % len(bucket)]
            ids = ids[self.rank::self.num_replicas]
            for j in range(len(ids) // self.batch_size):
                batches.append([bucket[k] for k in ids[j * self.batch_size:(j + 1) * self.batch_size]])
        if self.shuffle:
            batches = [batches[i] for i in torch.randperm(len(batches), generator=g).tolist()]
        assert len(batches) * self.batch_size == self.num_samples
        return iter(batches)


def _even_pad(n):
    return int(2 * ((n // 2) + 1))


def _round_up(n, k):
    return -(-n // k) * k


def collate_s2(items, spec_bins, pin=False, pad_frames=0, hop=640, with_spec=True, alloc=None):
    """TextAudioSpeakerCollate (data_utils.py:167-226) for items (ssl, wav, text, frames, ...): rows sorted by spectrogram
    length, longest first; ssl and spec time axes padded to 2*(max//2+1); everything else to the batch maximum.
    Returns the reference's 8-tuple with `spec_padded` zero-filled plus `order` (source index of each row): the caller
    writes row i's spectrogram into spec_padded[i, :, :spec_lengths[i]].
    pad_frames > 0 (not in the reference) rounds the padded time axes up to a multiple of that many frames (samples: x hop):
    every consumer masks by the lengths, so the step's result does not change, but the batches of a run then repeat a
    few dozen shapes instead of several hundred -- what the trainer's per-shape HIP-graph replay needs.
    with_spec=False leaves `spec_padded` out (None; its shape is returned as the third element instead): the reader
    creates it on the device, there is nothing in it to copy from the host.
    alloc(shape, dtype) -> zero-filled tensor: where the four padded tensors live (a reader process puts them into its
    shared-memory slot); default: fresh (optionally pinned) tensors."""
    n = len(items)
    _, order = torch.sort(torch.tensor([it[3] for it in items], dtype=torch.long), dim=0, descending=True)
    order = order.tolist()
    max_ssl = _even_pad(max(it[0].size(2) for it in items))
    max_spec = _even_pad(max(it[3] for it in items))
    max_wav = max(it[1].size(1) for it in items)
    if pad_frames > 0:
        max_ssl = max_spec = _round_up(max(max_ssl, max_spec), pad_frames)
        max_wav = _round_up(max_wav, pad_frames * hop)
    max_text = max(it[2].size(0) for it in items)

    def buf(shape, dtype):
        if alloc is not None:
            return alloc(shape, dtype)
        return torch.zeros(shape, dtype=dtype, pin_memory=pin)

    def buffer_dtype(k):
        """float32 like the reference's collate; on the reader's path (with_spec=False) the items' own dtype when they
        agree (int16 samples / fp16 features: half the bytes to pin and copy, converted on the device)"""
        dts = {it[k].dtype for it in items if len(it) < 5 or it[4]}        # placeholder items do not vote
        return next(iter(dts)) if (not with_spec and len(dts) == 1) else torch.float32

    ssl_dt, wav_dt = buffer_dtype(0), buffer_dtype(1)
    ssl_p = buf((n, items[0][0].size(1), max_ssl), ssl_dt)
    spec_p = buf((n, spec_bins, max_spec), torch.float32) if with_spec else (n, spec_bins, max_spec)
    wav_p = buf((n, 1, max_wav), wav_dt)
    text_p = buf((n, max_text), torch.long)
    ssl_l, spec_l, wav_l, text_l = (torch.zeros(n, dtype=torch.long) for _ in range(4))
    for i, src in enumerate(order):
        ssl, wav, text, frames = items[src][:4]
        ssl_p[i, :, :ssl.size(2)] = ssl[0]
        ssl_l[i] = ssl.size(2)
        spec_l[i] = frames
        if wav.dtype == torch.int16 and wav_p.dtype != torch.int16:      # mixed batch (a non-mono file came as float)
            wav = wav.float() * (1.0 / 32768.0)
        wav_p[i, :, :wav.size(1)] = wav
        wav_l[i] = wav.size(1)
        text_p[i, :text.size(0)] = text
        text_l[i] = text.size(0)
    return (ssl_p, ssl_l, spec_p, spec_l, wav_p, wav_l, text_p, text_l), order


class _Prefetch:
    """host-side read+collate of the next batches on one thread while the GPU runs the current step"""

    def __init__(self, make, keys, depth):
        self.q = queue.Queue(maxsize=max(1, depth))
        self.stop = threading.Event()

        def run():
            try:
                for k in keys:
                    if self.stop.is_set():
                        return
                    self.q.put(("ok", make(k)))
                self.q.put(("end", None))
            except BaseException as e:  # surfaced on the consumer side
                self.q.put(("err", e))

        self.t = threading.Thread(target=run, daemon=True)
        self.t.start()

    def __iter__(self):
        try:
            while True:
                kind, val = self.q.get()
                if kind == "end":
                    return
                if kind == "err":
                    raise val
                yield val
        finally:
            self.stop.set()
            while self.t.is_alive():
                try:
                    self.q.get_nowait()
                except queue.Empty:
                    self.t.join(0.01)


class _SlotAlloc:
    """carves zero-filled tensors out of one flat uint8 buffer (a reader process's shared-memory slot), 64-byte aligned;
    remembers (offset, shape, dtype) of each so that the parent can rebuild the views"""

    def __init__(self, slab):
        self.slab, self.off, self.metas = slab, 0, []

    def __call__(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        off = (self.off + 63) // 64 * 64
        if off + nbytes > self.slab.numel():
            raise RuntimeError(f"reader slot of {self.slab.numel()} bytes is too small for a batch tensor {tuple(shape)} {dtype}")
        t = self.slab[off: off + nbytes].view(dtype).view(*shape)
        t.zero_()
        self.off = off + nbytes
        self.metas.append((off, tuple(int(d) for d in shape), dtype))
        return t


def _slot_views(slab, metas):
    out = []
    for off, shape, dtype in metas:
        n = 1
        for d in shape:
            n *= d
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        out.append(slab[off: off + nbytes].view(dtype).view(*shape))
    return out


def _s2_worker(exp_dir, data_cfg, symbol_to_id, spec_bins, pad_frames, hop, slabs, task_q, out_q):
    """body of one reader process (S2Reader(loader_workers=N)): read + collate the batches whose (sequence number, slot,
    item indices) arrive on task_q STRAIGHT INTO the shared-memory slot the parent named -- file dtypes: int16 samples,
    fp16 features -- and answer with the layout only (offsets, shapes, lengths): no tensor crosses the queue (torch's
    shared-memory pickler took 30-170 ms per 13 MB batch for its per-tensor shm files; the slots are made once).  Never
    touches a GPU: this is the role of the reference's DataLoader workers (src/train/sovits.py:258-267: num_workers=6,
    persistent, prefetch_factor=4)."""
    torch.set_num_threads(1)
    try:
        ds = S2FeatureDir(exp_dir, data_cfg, symbol_to_id=symbol_to_id)
        while True:
            task = task_q.get()
            if task is None:
                return
            seq, slot, indices = task
            items = [ds.load(i, raw=True) for i in indices]
            al = _SlotAlloc(slabs[slot])
            batch, order = collate_s2(items, spec_bins, pad_frames=pad_frames, hop=hop, with_spec=False, alloc=al)
            ssl_p, ssl_l, spec_shape, spec_l, wav_p, wav_l, text_p, text_l = batch
            out_q.put((seq, slot, al.metas, ssl_l.tolist(), tuple(spec_shape), spec_l.tolist(), wav_l.tolist(),
                       text_l.tolist(), [bool(items[src][4]) for src in order], None))
    except BaseException as e:  # noqa: BLE001 -- surfaced on the consumer side
        out_q.put((-1, -1, None, None, None, None, None, None, None, f"{e!r}\n{traceback.format_exc()}"))


class _ProcPrefetch:
    """read + collate on worker PROCESSES (one GIL each), batches delivered in sampler order; at most `depth` batches are
    in flight, each in its own shared-memory slot (page-locked in the parent when the consumer is a GPU: the copy to the
    device is then asynchronous straight from the slot).  The workers are started once per reader and stay
    (persistent_workers=True there)."""

    def __init__(self, reader, workers, depth):
        import multiprocessing as mp

        ds = reader.ds
        self.depth = max(depth, workers)
        # the largest batch the sampler can form: batch_size items of the longest clip (+ the pad_frames rounding)
        frames = int(max(ds.lengths)) + 2 + max(reader.pad_frames, 0)
        bsz = reader.sampler.batch_size
        text_max = max((len(ids) for _name, ids in ds.items), default=1)
        slot_bytes = bsz * (768 * frames * 4 + (frames + 4) * ds.hop_length * 4 + text_max * 8) + 4096
        self.slabs = [torch.zeros(slot_bytes, dtype=torch.uint8).share_memory_() for _ in range(self.depth)]
        self.pinned = False
        if reader.device.type == "cuda":
            try:       # page-lock the slots once: non_blocking copies from their views are then truly asynchronous
                rt = torch.cuda.cudart()
                for sl in self.slabs:
                    rt.cudaHostRegister(sl.data_ptr(), sl.numel(), 0)
                self.pinned = True
            except Exception:  # noqa: BLE001 -- unpinned slots still work (the copy is then staged by the runtime)
                self.pinned = False
        ctx = mp.get_context("spawn")          # the parent has a HIP context: never fork it
        self.task_q, self.out_q = ctx.Queue(), ctx.Queue()
        args = (reader._exp_dir, reader._data_cfg, reader._symbol_to_id, reader.spec_bins, reader.pad_frames, ds.hop_length,
                self.slabs, self.task_q, self.out_q)
        self.procs = [ctx.Process(target=_s2_worker, args=args, daemon=True) for _ in range(workers)]
        for p in self.procs:
            p.start()
        self.free = list(range(self.depth))

    def run(self, keys, release):
        """generator over (batch, ok, slot) of `keys` (lists of item indices), in order.  `release` = list the consumer
        appends a slot to once nothing reads it any more (after its device copies have completed)."""
        keys = iter(keys)
        sent = got = 0
        held, done = {}, False

        def feed():
            nonlocal sent, done
            while release:
                self.free.append(release.pop())
            while not done and self.free:
                try:
                    k = next(keys)
                except StopIteration:
                    done = True
                    return
                self.task_q.put((sent, self.free.pop(), list(k)))
                sent += 1

        try:
            feed()
            while got < sent:
                while got not in held:
                    msg = self.out_q.get()
                    if msg[-1] is not None:
                        raise RuntimeError(f"s2 reader worker failed: {msg[-1]}")
                    held[msg[0]] = msg
                _seq, slot, metas, ssl_l, spec_shape, spec_l, wav_l, text_l, ok, _err = held.pop(got)
                got += 1
                ssl_p, wav_p, text_p = _slot_views(self.slabs[slot], metas)
                lt = lambda v: torch.tensor(v, dtype=torch.long)
                yield (ssl_p, lt(ssl_l), spec_shape, lt(spec_l), wav_p, lt(wav_l), text_p, lt(text_l)), ok, slot
                if isinstance(release, _AutoRelease):
                    release.append(slot)      # the consumer came back for the next batch: it is done with this one
                feed()
        finally:
            # an epoch left early: collect what the workers still produce for it (nothing may leak into the next epoch)
            while got < sent:
                try:
                    msg = self.out_q.get(timeout=30)
                except Exception:  # noqa: BLE001
                    break
                if msg[1] >= 0:
                    self.free.append(msg[1])
                got += 1
            for seq, msg in held.items():
                self.free.append(msg[1])

    def close(self):
        for _ in self.procs:
            self.task_q.put(None)
        for p in self.procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()          # by handle: our own children only
        self.procs = []
        if self.pinned:
            try:
                rt = torch.cuda.cudart()
                for sl in self.slabs:
                    rt.cudaHostUnregister(sl.data_ptr())
            except Exception:  # noqa: BLE001
                pass
            self.pinned = False


class _AutoRelease(list):
    """release list of a consumer that is done with a slot as soon as it asks for the next batch (host-only consumers:
    tools/bench_reader.py): run() hands the slot out, the consumer never appends -- so every slot handed out is recycled
    on the next feed"""

    def __init__(self):
        super().__init__()
        self.last = None


class S2Reader:
    """Iterable of device batches with the layout of the reference's s2 DataLoader (src/train/sovits.py:229-267)."""

    def __init__(self, exp_dir, data_cfg, batch_size, device, rank=0, world=1, boundaries=None, prefetch=4,
                 symbol_to_id=None, spec_fn=None, pad_frames=None, loader_threads=1, loader_workers=None):
        """loader_workers (default: EVT_READER_WORKERS, else 0): that many reader PROCESSES read and collate the batches
        (the reference's DataLoader(num_workers=6, persistent_workers=True, prefetch_factor=4), sovits.py:258-267); 0 keeps
        the one prefetch thread.  The batches are the same either way (same sampler order, same collate); the GPU side --
        copy, dtype conversion, spectrogram -- is unchanged."""
        self._exp_dir, self._data_cfg, self._symbol_to_id = exp_dir, dict(data_cfg), symbol_to_id
        self.ds = S2FeatureDir(exp_dir, data_cfg, symbol_to_id=symbol_to_id)
        self.sampler = S2BucketSampler(self.ds.lengths, batch_size, boundaries, num_replicas=world, rank=rank)
        self.device, self.prefetch = torch.device(device), prefetch
        self.pad_frames = int(os.environ.get("EVT_PAD_FRAMES", "0")) if pad_frames is None else int(pad_frames)
        # one host thread reads ~830 items/s from the page cache (8 cores of this container; a B=16 step every 32 ms needs
        # 500): threads on top of it only fight over the interpreter lock (2: 620, 4: 475 items/s measured), so the option
        # stays off; the cheap part -- sample and feature conversion to float32 -- runs on the GPU after the copy instead
        self._pool = None
        if loader_threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=loader_threads, thread_name_prefix="evt-read")
        self.spec_bins = self.ds.filter_length // 2 + 1
        if spec_fn is None:
            from ..module.mel_processing import spectrogram_torch as spec_fn
        self.spec_fn = spec_fn
        self.loader_workers = int(os.environ.get("EVT_READER_WORKERS", "0")) if loader_workers is None else int(loader_workers)
        self._procs = None

    def close(self):
        """stop the reader processes (they are daemons: a parent that exits takes them along anyway)"""
        if self._procs is not None:
            self._procs.close()
            self._procs = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _host_batches(self, release=None):
        """iterator of (collated host batch, per-item ok flags[, slot]) in sampler order"""
        if self.loader_workers > 0:
            if self._procs is None:
                try:
                    self._procs = _ProcPrefetch(self, self.loader_workers, max(self.prefetch, 2 * self.loader_workers))
                except Exception as e:  # noqa: BLE001 -- e.g. a /dev/shm too small for the slots: read on the one thread
                    import warnings

                    warnings.warn(f"S2Reader: reader processes unavailable ({e!r}); using the prefetch thread")
                    self.loader_workers = 0
                    return ((b, ok, None) for b, ok in _Prefetch(self._host_batch, iter(self.sampler), self.prefetch))
            return self._procs.run(iter(self.sampler), release if release is not None else _AutoRelease())
        return ((b, ok, None) for b, ok in _Prefetch(self._host_batch, iter(self.sampler), self.prefetch))

    def set_epoch(self, epoch):
        self.sampler.set_epoch(epoch)

    def __len__(self):
        return len(self.sampler)

    def _host_batch(self, indices):
        ld = lambda i: self.ds.load(i, raw=True)
        items = list(self._pool.map(ld, indices)) if self._pool is not None else [ld(i) for i in indices]
        batch, order = collate_s2(items, self.spec_bins, pin=self.device.type == "cuda", pad_frames=self.pad_frames,
                                  hop=self.ds.hop_length, with_spec=False)
        return batch, [items[src][4] for src in order]

    def __iter__(self):
        ds, dev = self.ds, self.device
        release, pending = [], []             # slots the workers may refill; (slot, event of its device copies) in flight
        for (ssl, ssl_l, spec, spec_l, wav, wav_l, text, text_l), ok, slot in self._host_batches(release):
            if slot is not None and dev.type != "cuda":
                ssl, wav, text = ssl.clone(), wav.clone(), text.clone()       # the slot is refilled behind the consumer
            ssl, wav, text = (t.to(dev, non_blocking=True) for t in (ssl, wav, text))
            if slot is not None:
                if dev.type == "cuda":
                    ev = torch.cuda.Event()
                    ev.record()
                    pending.append((slot, ev))
                    while pending and (pending[0][1].query() or len(pending) >= self._procs.depth - 1):
                        s0, e0 = pending.pop(0)
                        e0.synchronize()                                      # (already complete unless the ring is full)
                        release.append(s0)
                else:
                    release.append(slot)
            ssl = ssl.float()                                              # file dtype (fp16) -> float32, as the collate there
            if wav.dtype == torch.int16:
                wav = wav.float() * (1.0 / 32768.0)                        # what ffmpeg's s16 -> flt does, exactly
            spec = torch.zeros(spec, dtype=torch.float32, device=dev)      # `spec` came as the shape (with_spec=False)
            for i in range(wav.size(0)):
                if not ok[i]:
                    continue
                n, frames = int(wav_l[i]), int(spec_l[i])
                s = self.spec_fn(wav[i, :, :n], ds.filter_length, ds.sampling_rate, ds.hop_length, ds.win_length,
                                 center=False)
                spec[i, :, :frames] = s[0]
            yield (ssl, ssl_l.to(dev), spec, spec_l.to(dev), wav, wav_l.to(dev), text, text_l.to(dev))


# ---------------------------------------------------------------------------------------------------------------------
# s1: Text2SemanticDataset / DistributedBucketSampler / collate
# ---------------------------------------------------------------------------------------------------------------------
class S1SemanticTable:
    """(semantic_ids, phoneme_ids) pairs of 6-name2semantic.tsv x 2-name2text.txt, dataset.py:38-183.

    Filters: more than max_sec*hz semantic tokens; more than max_sec*hz/2.5 phonemes; phonemes per second outside
    [min_ps_ratio, max_ps_ratio]; unknown names/phonemes.  Fewer than 100 survivors are repeated max(2, 100//n)
    times.  The tsv's first line is a header (the reference reads it with pandas.read_csv)."""

    def __init__(self, phoneme_path, semantic_path, max_sec=100, pad_val=1024, min_ps_ratio=3, max_ps_ratio=25,
                 max_sample=None, symbol_to_id=None):
        for p in (phoneme_path, semantic_path):
            if not os.path.exists(p):
                raise FileNotFoundError(p)
        self.path3 = "%s/3-bert" % os.path.dirname(phoneme_path)
        self.PAD = pad_val
        self.hz = int(os.environ.get("hz", "25hz")[:-2])
        sym = symbol_to_id if symbol_to_id is not None else load_symbol_table(os.path.dirname(phoneme_path))
        phoneme_data = read_name2text(phoneme_path)
        with open(semantic_path, "r", encoding="utf-8") as f:
            rows = [ln.split("\t") for ln in f.read().split("\n")[1:] if ln.strip() != ""]
        if max_sample is not None:
            rows = rows[:max_sample]
        self.semantic_phoneme, self.item_names = [], []
        self.num_not_in = self.num_deleted_bigger = self.num_deleted_ps = 0
        for row in rows:
            name = row[0]
            if name not in phoneme_data:
                self.num_not_in += 1
                continue
            semantic_ids = [int(v) for v in row[1].split(" ")]
            if len(semantic_ids) > max_sec * self.hz:
                self.num_deleted_bigger += 1
                continue
            try:
                phoneme_ids = [sym[p] for p in phoneme_data[name][0].split(" ")]
            except KeyError:
                self.num_not_in += 1
                continue
            if len(phoneme_ids) > max_sec * self.hz / 2.5:
                self.num_deleted_ps += 1
                continue
            ps_ratio = len(phoneme_ids) / (len(semantic_ids) / self.hz)
            if ps_ratio > max_ps_ratio or ps_ratio < min_ps_ratio:
                self.num_deleted_ps += 1
                continue
            self.semantic_phoneme.append((semantic_ids, phoneme_ids))
            self.item_names.append(name)
        n = len(self.semantic_phoneme)
        if n == 0:
            raise ValueError(f"no valid data in {semantic_path}, please check the data and try again")
        if n < 100:
            rep = max(2, int(100 / n))
            self.semantic_phoneme, self.item_names = self.semantic_phoneme * rep, self.item_names * rep

    def __len__(self):
        return len(self.semantic_phoneme)

    def get_sample_length(self, idx):
        return 1.0 * len(self.semantic_phoneme[idx][0]) / self.hz

    def load(self, idx):
        semantic_ids, phoneme_ids = self.semantic_phoneme[idx]
        path_bert = "%s/%s.pt" % (self.path3, self.item_names[idx])
        bert = None
        if os.path.exists(path_bert):
            bert = torch.load(path_bert, map_location="cpu")
            assert bert.shape[-1] == len(phoneme_ids), (self.item_names[idx], tuple(bert.shape), len(phoneme_ids))
        return dict(idx=idx, phoneme_ids=phoneme_ids, phoneme_ids_len=len(phoneme_ids), semantic_ids=semantic_ids,
                    semantic_ids_len=len(semantic_ids), bert_feature=bert)

    def collate(self, examples, pin=False):
        """dataset.py:222-271: phonemes padded with 0, semantic tokens with PAD, bert features [B,1024,max_phones]
        zero-filled where an item has none"""
        n = len(examples)
        max_ph = max(e["phoneme_ids_len"] for e in examples)
        max_se = max(e["semantic_ids_len"] for e in examples)
        ph = torch.zeros((n, max_ph), dtype=torch.long, pin_memory=pin)
        se = torch.full((n, max_se), self.PAD, dtype=torch.long, pin_memory=pin)
        bert = torch.zeros((n, 1024, max_ph), dtype=torch.float32, pin_memory=pin)
        for i, e in enumerate(examples):
            ph[i, :e["phoneme_ids_len"]] = torch.tensor(e["phoneme_ids"], dtype=torch.long)
            se[i, :e["semantic_ids_len"]] = torch.tensor(e["semantic_ids"], dtype=torch.long)
            if e["bert_feature"] is not None:
                bert[i, :, :e["bert_feature"].shape[-1]] = e["bert_feature"]
        return dict(ids=[e["idx"] for e in examples], phoneme_ids=ph,
                    phoneme_ids_len=torch.tensor([e["phoneme_ids_len"] for e in examples]),
                    semantic_ids=se, semantic_ids_len=torch.tensor([e["semantic_ids_len"] for e in examples]),
                    bert_feature=bert)


class S1BucketSampler:
    """Index stream of one rank, bucket_sampler.py:29-170: items sorted by duration into 2-second buckets, shuffled
    inside each bucket and then as world*batch_size groups with Python's Mersenne Twister seeded by seed+epoch, padded
    by repetition to a multiple of the replica count and strided by rank.  `batches()` cuts it the way the reference's
    DataLoader(batch_size=..., drop_last=False) does."""

    def __init__(self, table, batch_size, num_replicas=1, rank=0, shuffle=True, seed=0, drop_last=False):
        if rank >= num_replicas or rank < 0:
            raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {num_replicas - 1}]")
        self.n, self.batch_size = len(table), batch_size
        self.num_replicas, self.rank, self.shuffle, self.seed, self.drop_last, self.epoch = \
            num_replicas, rank, shuffle, seed, drop_last, 0
        if drop_last and self.n % num_replicas != 0:
            self.num_samples = math.ceil((self.n - num_replicas) / num_replicas)
        else:
            self.num_samples = math.ceil(self.n / num_replicas)
        self.total_size = self.num_samples * num_replicas
        by_len = sorted(((i, table.get_sample_length(i)) for i in range(self.n)), key=lambda x: x[1])
        self.id_buckets, cur, max_sec = [], [], 2.0
        for i, sec in by_len:
            if sec < max_sec:
                cur.append(i)
            else:
                self.id_buckets.append(cur)
                cur = [i]
                max_sec += 2.0
        if len(cur) > 0:
            self.id_buckets.append(cur)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        if self.shuffle:
            rng = random.Random(self.epoch + self.seed)
            flat = []
            for buc in self.id_buckets:
                buc = buc.copy()
                rng.shuffle(buc)
                flat += buc
            group = self.batch_size * self.num_replicas
            groups = [flat[b * group:(b + 1) * group] for b in range(int(math.ceil(len(flat) / group)))]
            rng.shuffle(groups)
            indices = [i for g in groups for i in g]
        else:
            indices = list(range(self.n))
        if not self.drop_last:
            pad = self.total_size - len(indices)
            if pad <= len(indices):
                indices += indices[:pad]
            else:
                indices += (indices * math.ceil(pad / len(indices)))[:pad]
        else:
            indices = indices[:self.total_size]
        assert len(indices) == self.total_size
        indices = indices[self.rank:self.total_size:self.num_replicas]
        assert len(indices) == self.num_samples
        return iter(indices)

    def batches(self):
        idx = list(iter(self))
        return [idx[i:i + self.batch_size] for i in range(0, len(idx), self.batch_size)]


class S1Reader:
    """Iterable of device batches with the layout of Text2SemanticDataModule.train_dataloader (data_module.py:42-54)."""

    def __init__(self, exp_dir, data_cfg, batch_size, device, rank=0, world=1, prefetch=8, symbol_to_id=None,
                 phoneme_name="2-name2text.txt", semantic_name="6-name2semantic.tsv", if_dpo=False):
        self.table = S1SemanticTable(os.path.join(exp_dir, phoneme_name), os.path.join(exp_dir, semantic_name),
                                     max_sec=data_cfg.get("max_sec", 100), pad_val=data_cfg.get("pad_val", 1024),
                                     symbol_to_id=symbol_to_id)
        if if_dpo or data_cfg.get("if_dpo", False):
            batch_size = batch_size // 2
        self.batch_size = max(min(batch_size, len(self.table) // 4), 1)
        self.sampler = S1BucketSampler(self.table, self.batch_size, num_replicas=world, rank=rank)
        self.device, self.prefetch = torch.device(device), prefetch

    def set_epoch(self, epoch):
        self.sampler.set_epoch(epoch)

    def __len__(self):
        return math.ceil(len(self.sampler) / self.batch_size)

    def _host_batch(self, indices):
        return self.table.collate([self.table.load(i) for i in indices], pin=self.device.type == "cuda")

    def __iter__(self):
        for b in _Prefetch(self._host_batch, self.sampler.batches(), self.prefetch):
            yield {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in b.items()}
