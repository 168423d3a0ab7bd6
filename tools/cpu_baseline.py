"""Times the oracle's s2 step (oracle/s2_step.py) or s1 micro-step (oracle/s1_step.py) on the host cores; run by
bench.py as a SUBPROCESS with a hard timeout so the GPU bench line never waits on a slow host.  Prints one JSON line
after every timed step (the parent keeps the last one)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def main_s1(a, threads):
    """one s1 micro-step (forward_old + backward through autograd) at L = 256 + 768.  The CPU math path keeps the
    [B*16, L, L] probabilities of all 24 layers: ~6.5 GB per item (B = 8 is 52.7 GB), so the batch is
    sized to the host's free memory: 8 (BASELINE.md's figure) with >= 120 GB free, else 2."""
    import torch
    import yaml

    torch.set_num_threads(threads)
    from easevoice_trainer_amd.auto_reg.t2s_model import Text2SemanticDecoder
    from oracle import s1_step as O

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    B = a.batch if a.batch > 0 else (8 if _mem_available_gb() >= 120 else 2)
    x_len, y_len = 256, 768
    torch.manual_seed(1234)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point())
          for k, v in Text2SemanticDecoder(cfg).state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    x = torch.randint(0, 732, (B, x_len), generator=g)
    y = torch.randint(0, 1024, (B, y_len), generator=g)
    bert = torch.randn(B, 1024, x_len, generator=g)
    xl, yl = torch.full((B,), x_len), torch.full((B,), y_len)
    t_start, times = time.perf_counter(), []
    while len(times) < 3 and (not times or (time.perf_counter() - t_start) + min(times) < a.budget):
        t0 = time.perf_counter()
        loss, _acc, _ = O.forward_old(sd, cfg, x, xl, y, yl, bert)
        grads = torch.autograd.grad(loss, [v for v in sd.values() if v.requires_grad], allow_unused=True)
        del grads, loss
        times.append(time.perf_counter() - t0)
        best = min(times)
        print(json.dumps(dict(
            value=B * (x_len + y_len) / best, unit="tokens/s", cores=threads, kind="port", seconds_per_step=best,
            sample=f"oracle s1 micro-step (forward_old + backward, no optimiser), batch {B} x (256 + 768) tokens, fp32, "
                   f"{threads} threads, best of {len(times)} step(s), {best:.2f} s/step "
                   f"(batch sized to {_mem_available_gb():.0f} GB of free host memory)")), flush=True)


def time_reference_s2(B, clip_seconds, threads, steps=3):
    """Build box only (needs /root/reference): the REFERENCE's own modules through the reference's loop body
    (src/train/sovits.py:459-525 in fp32: G forward, D step, G step, both torch.optim.AdamW updates) on the batch the port is
    timed on -- so that the `cpu_baseline` of the bench line (kind "port": the GPU box has no reference checkout) can be read
    against the reference's own step time.  Returns seconds per step (median)."""
    import torch

    from oracle import refshim

    refshim.install()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_r2 import g_groups
    from src.easevoice.module import commons, models
    from src.easevoice.module.losses import discriminator_loss, feature_loss, generator_loss, kl_loss
    from src.easevoice.module.mel_processing import mel_spectrogram_torch, spec_to_mel_torch, spectrogram_torch

    torch.set_num_threads(threads)
    cfg = json.load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "s2.json")))
    t = cfg["train"]
    torch.manual_seed(1234)
    net_g = models.SynthesizerTrn(1025, 32, n_speakers=300, **cfg["model"])
    net_d = models.MultiPeriodDiscriminator(False)
    cb = net_g.quantizer.vq.layers[0]._codebook
    cb.embed.normal_()
    cb.inited.fill_(1.0)
    lr, low = t["learning_rate"], t["learning_rate"] * t["text_low_lr_rate"]
    optim_g = torch.optim.AdamW(g_groups(net_g, lr, low), lr, betas=t["betas"], eps=t["eps"])
    optim_d = torch.optim.AdamW(net_d.parameters(), lr, betas=t["betas"], eps=t["eps"])
    T, tt = clip_seconds * 50, 60
    gen = torch.Generator().manual_seed(1234)
    wav = torch.rand(B, 1, T * 640, generator=gen) - 0.5
    ssl = torch.randn(B, 768, T, generator=gen)
    text = torch.randint(0, 732, (B, tt), generator=gen)
    lens, tl = torch.full((B,), T), torch.full((B,), tt)
    spec = spectrogram_torch(wav.squeeze(1), 2048, 32000, 640, 2048, center=False)
    times = []
    for step in range(steps + 1):
        t0 = time.perf_counter()
        y_hat, kl_ssl, ids_slice, x_mask, z_mask, (z, z_p, m_p, logs_p, m_q, logs_q), _ = net_g(ssl, spec, lens, text, tl)
        mel = spec_to_mel_torch(spec, 2048, 128, 32000, 0.0, None)
        y_mel = commons.slice_segments(mel, ids_slice, 32)
        y_hat_mel = mel_spectrogram_torch(y_hat.squeeze(1), 2048, 128, 32000, 640, 2048, 0.0, None)
        y = commons.slice_segments(wav, ids_slice * 640, 20480)
        y_d_hat_r, y_d_hat_g, _, _ = net_d(y, y_hat.detach())
        loss_disc, _, _ = discriminator_loss(y_d_hat_r, y_d_hat_g)
        optim_d.zero_grad()
        loss_disc.backward()
        optim_d.step()
        y_d_hat_r, y_d_hat_g, fmap_r, fmap_g = net_d(y, y_hat)
        loss_gen_all = (generator_loss(y_d_hat_g)[0] + feature_loss(fmap_r, fmap_g)
                        + torch.nn.functional.l1_loss(y_mel, y_hat_mel) * t["c_mel"] + kl_ssl * 1
                        + kl_loss(z_p, logs_q, m_p, logs_p, z_mask) * t["c_kl"])
        optim_g.zero_grad()
        loss_gen_all.backward()
        optim_g.step()
        if step:
            times.append(time.perf_counter() - t0)
    return sorted(times)[len(times) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--with-reference", action="store_true",
                    help="build box only: also time the reference's own modules on the same batch and print the port / reference ratio")
    ap.add_argument("--stage", default="s2", choices=["s2", "s1"])
    ap.add_argument("--batch", type=int, default=0)
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
    if a.stage == "s1":
        return main_s1(a, threads)
    a.batch = a.batch or 16
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
    if a.with_reference and os.path.isdir("/root/reference"):
        ref = time_reference_s2(B, a.clip_seconds, threads)
        print(json.dumps(dict(kind="reference", seconds_per_step=ref, value=B * a.clip_seconds / ref, unit="audio-s/s",
                              cores=threads, port_over_reference=med / ref,
                              sample=f"the reference's own modules (src/train/sovits.py:459-525 body, fp32, torch AdamW), batch "
                                     f"{B} x {a.clip_seconds} s clips, {threads} threads, 1 warm-up + 3 timed steps, median "
                                     f"{ref:.2f} s/step; the oracle port takes {med / ref:.2f} x that")), flush=True)


if __name__ == "__main__":
    main()
