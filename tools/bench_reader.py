"""Real-data path of the s2 trainer on a synthetic feature directory of realistic size: how fast the
host side of the reader is, and -- on a GPU -- how a training run over bucketed, ragged batches behaves (step time,
how many steps were replayed from a captured HIP graph, with and without EVT_PAD_FRAMES).

    python tools/bench_reader.py                      # host side only (no GPU needed)
    python tools/bench_reader.py --train-steps 200    # + train from the directory on cuda:0
    EVT_PAD_FRAMES=64 python tools/bench_reader.py --train-steps 200
"""
import argparse
import json
import os

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime loads: easevoice_trainer_amd/__init__.py
import sys
import tempfile
import time
import wave

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_feature_dir(root, n_items, min_s, max_s, seed=0, n_symbols=64):
    """2-name2text.txt + 4-cnhubert + 5-wav32k with random content: clip lengths uniform in [min_s, max_s] seconds"""
    rng = np.random.RandomState(seed)
    for d in ("4-cnhubert", "5-wav32k"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    symbols = [f"s{i}" for i in range(n_symbols)]
    lines = []
    for i in range(n_items):
        n = int(rng.uniform(min_s, max_s) * 32000)
        name = f"u{i:05d}.wav"
        with wave.open(os.path.join(root, "5-wav32k", name), "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(32000)
            w.writeframes((rng.randn(n) * 3000).astype("<i2").tobytes())
        frames = (n - 640) // 640 + 1
        torch.save(torch.randn(1, 768, frames).half(), os.path.join(root, "4-cnhubert", name + ".pt"))
        phones = " ".join(symbols[j] for j in rng.randint(0, n_symbols, max(4, n // 4000)))
        lines.append("\t".join([name, phones, "[1]", "t"]))
    with open(os.path.join(root, "2-name2text.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(root, "symbols.json"), "w") as f:
        json.dump(symbols, f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=512)
    ap.add_argument("--min-seconds", type=float, default=2.0)
    ap.add_argument("--max-seconds", type=float, default=10.0)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--train-steps", type=int, default=0)
    ap.add_argument("--workers", type=int, default=0, help="reader processes (S2Reader(loader_workers=N)); 0 = the one thread")
    args = ap.parse_args()
    from easevoice_trainer_amd.train import dataset as D

    cfg = dict(sampling_rate=32000, filter_length=2048, hop_length=640, win_length=2048)
    out = {"items": args.items, "seconds": [args.min_seconds, args.max_seconds], "batch": args.batch,
           "pad_frames": int(os.environ.get("EVT_PAD_FRAMES", "0"))}
    with tempfile.TemporaryDirectory() as root:
        t0 = time.perf_counter()
        make_feature_dir(root, args.items, args.min_seconds, args.max_seconds)
        out["make_dir_s"] = round(time.perf_counter() - t0, 2)
        rd = D.S2Reader(root, cfg, args.batch, "cpu", spec_fn=lambda y, *a, **k: torch.zeros(
            1, 1025, D.spec_frames(y.size(1), 2048, 640)))
        rd.set_epoch(1)
        batches = list(iter(rd.sampler))
        t0 = time.perf_counter()
        shapes = set()
        for b in batches:
            (ssl, _sl, spec_shape, *_rest), _ok = rd._host_batch(b)
            shapes.add(spec_shape[2])
        dt = time.perf_counter() - t0
        n = sum(len(b) for b in batches)
        out["host"] = dict(items_per_s=round(n / dt, 1), ms_per_batch=round(1e3 * dt / len(batches), 2),
                           batches_per_epoch=len(batches), distinct_padded_lengths=len(shapes))
        if args.workers > 0:
            # the same epoch through reader processes: what the consumer sees per batch (second epoch: workers are up)
            rw = D.S2Reader(root, cfg, args.batch, "cpu", loader_workers=args.workers, spec_fn=lambda y, *a, **k: torch.zeros(
                1, 1025, D.spec_frames(y.size(1), 2048, 640)))
            try:
                for ep in (1, 2, 3):      # the third epoch: workers up, slot pages touched
                    rw.set_epoch(ep)
                    t0 = time.perf_counter()
                    nb = sum(1 for _ in rw._host_batches())
                    dtw = time.perf_counter() - t0
                out["host_workers"] = dict(workers=args.workers, items_per_s=round(n / dtw, 1),
                                           ms_per_batch=round(1e3 * dtw / nb, 2))
            finally:
                rw.close()
        if args.train_steps > 0:
            from easevoice_trainer_amd.train.s2_engine import S2Engine
            from easevoice_trainer_amd.train.dataset import S2Reader

            dev = torch.device("cuda:0")
            hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
            torch.manual_seed(hps["train"]["seed"])
            eng = S2Engine(hps, dev, torch.bfloat16)
            cb = eng.net_g.quantizer.vq.layers[0]._codebook
            cb.embed.normal_()
            cb.inited.fill_(1.0)
            eng.build_optimizers()
            eng.enable_graphs(warmup_steps=2, max_shapes=int(os.environ.get("EVT_GRAPH_SHAPES", "64")))      # the trainer's default
            src = S2Reader(root, cfg, args.batch, dev, loader_workers=args.workers)
            steps, epoch, wait, audio_s = 0, 0, 0.0, 0.0
            tail_from = args.train_steps * 2 // 3           # the last third: most shapes are captured by then
            tail_t0 = tail_audio0 = tail_rep0 = None
            torch.cuda.synchronize()
            t_all = time.perf_counter()
            while steps < args.train_steps:
                epoch += 1
                src.set_epoch(epoch)
                it = iter(src)
                while steps < args.train_steps:
                    t0 = time.perf_counter()
                    try:
                        ssl, _l, spec, spec_len, y, _yl, text, text_len = next(it)
                    except StopIteration:
                        break
                    wait += time.perf_counter() - t0          # time the step loop spent waiting for the reader
                    if steps == tail_from:
                        torch.cuda.synchronize()
                        tail_t0, tail_audio0 = time.perf_counter(), audio_s
                        tail_rep0 = getattr(eng, "graph_steps", {"replayed": 0})["replayed"]
                    eng.step(ssl, spec, spec_len, y, text, text_len)
                    steps += 1
                    audio_s += float(spec_len.sum()) * 640 / 32000        # real (unpadded) audio of the batch
            torch.cuda.synchronize()
            dt = time.perf_counter() - t_all
            cache = getattr(eng, "_graph_cache", {})
            captured = sum(1 for e in cache.values() if e["graphs"] is not None)
            gs = getattr(eng, "graph_steps", {"replayed": 0, "eager": steps})
            out["train"] = dict(steps=steps, ms_per_step=round(1e3 * dt / steps, 2), reader_wait_ms_per_step=round(1e3 * wait / steps, 2),
                                audio_seconds_per_s=round(audio_s / dt, 1), shapes_seen=len(cache), shapes_captured=captured,
                                steps_replayed=gs["replayed"], eager_share=round(1.0 - gs["replayed"] / max(steps, 1), 3))
            if tail_t0 is not None and steps > tail_from:
                nt = steps - tail_from
                tdt = time.perf_counter() - tail_t0
                out["train"]["last_third"] = dict(steps=nt, ms_per_step=round(1e3 * tdt / nt, 2),
                                                  audio_seconds_per_s=round((audio_s - tail_audio0) / tdt, 1),
                                                  eager_share=round(1.0 - (gs["replayed"] - tail_rep0) / nt, 3))
            free, total = torch.cuda.mem_get_info(dev)
            out["train"]["device_memory_gb"] = dict(used=round((total - free) / 2 ** 30, 1), total=round(total / 2 ** 30, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
