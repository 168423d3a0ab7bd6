"""Times the oracle's s2 step (oracle/s2_step.py) on the host cores; run by bench.py as a SUBPROCESS with a hard
timeout so the GPU bench line never waits on a slow host.  Prints one JSON line after every timed step (the parent
keeps the last one)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--clip-seconds", type=int, default=4)
    ap.add_argument("--budget", type=float, default=40.0)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = a.threads or max(1, min(avail, 32))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    import torch

    torch.set_num_threads(threads)
    from easevoice_trainer_amd.module.models import MultiPeriodDiscriminator, SynthesizerTrn
    from oracle import s2_step as O

    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    d, m, t = hps["data"], hps["model"], hps["train"]
    B, T, tt = a.batch, a.clip_seconds * 50, 60
    torch.manual_seed(1234)
    g = SynthesizerTrn(d["filter_length"] // 2 + 1, t["segment_size"] // d["hop_length"], n_speakers=d["n_speakers"], **m)
    dd = MultiPeriodDiscriminator(False)
    sd_g = {k: v.detach().clone() for k, v in g.state_dict().items()}
    sd_d = {k: v.detach().clone() for k, v in dd.state_dict().items()}
    sd_g["quantizer.vq.layers.0._codebook.embed"].normal_()
    gen = torch.Generator().manual_seed(1234)
    wav = torch.rand(B, 1, T * 640, generator=gen) - 0.5
    ssl = torch.randn(B, 768, T, generator=gen)
    text = torch.randint(0, 732, (B, tt), generator=gen)
    eps = torch.randn(B, 192, T, generator=gen)
    ids = torch.randint(0, T - 32 + 1, (B,), generator=gen)
    lens, tl = torch.full((B,), T), torch.full((B,), tt)
    state = {}

    def one_step(step):
        out = O.s2_losses(sd_g, sd_d, hps, ssl, wav, text, lens, tl, eps, ids, with_grads=True)
        for sd, grads in ((sd_d, out["d_grads"]), (sd_g, out["g_grads"])):
            for k, gr in grads.items():
                if gr is None:
                    continue
                st = state.setdefault((id(sd), k), (torch.zeros_like(gr), torch.zeros_like(gr)))
                O.adamw_step(sd[k], gr, st[0], st[1], step, t["learning_rate"], tuple(t["betas"]), t["eps"])

    t_start = time.perf_counter()
    one_step(1)   # warm-up
    times, step = [], 2
    while len(times) < 3 and (time.perf_counter() - t_start) < a.budget:
        t0 = time.perf_counter()
        one_step(step)
        times.append(time.perf_counter() - t0)
        step += 1
        med = sorted(times)[len(times) // 2]
        print(json.dumps(dict(
            value=B * a.clip_seconds / med, unit="audio-s/s", cores=threads, kind="port", seconds_per_step=med,
            sample=f"oracle s2 step (fwd + D/G backward + AdamW on every tensor), batch {B} x {a.clip_seconds} s clips, "
                   f"fp32, {threads} threads, 1 warm-up + {len(times)} timed steps, median {med:.2f} s/step")), flush=True)


if __name__ == "__main__":
    main()
