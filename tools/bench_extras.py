"""bench.py's `roofline` and `cpu_baseline` legs for the s2 workload.

roofline: after the timed region, two more steps run with per-launch HIP events (torch.cuda.Event on the stream the
kernels are launched on) around every conv entry point; launches are grouped by the kernel instantiation the C++
dispatcher actually launched (evt_last_kernel_tag(), same names as rocprofv3 --kernel-trace).  The dominant
instantiation by total time is reported: achieved = sum(algorithmic flops or bytes of its launches) / sum(duration).
Algorithmic bytes of one conv launch = every operand tensor once (inputs + outputs in the compute dtype + the weight
image) — the unfused-per-conv figure of SURVEY §8(d); flops = 2 * MACs.  Peaks: 8 TB/s HBM, 2.5 PFLOP/s dense bf16
MFMA (/opt/skills/guides/MI355X_MICROARCH.md).  `traffic` (PMC HBM bytes) is collected in a separate rocprofv3 --pmc
pass (profiles/), not inside bench.py.

cpu_baseline: the oracle's s2 step (oracle/s2_step.py: forward, both backward passes, AdamW on every tensor) timed on
this box's host cores on a bounded sample of the same workload (2 clips of 4 s instead of 16).
"""
import json
import os
import time

import torch

HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TF = 2500.0
MFMA_F32_PEAK_TF = 157.3


def roofline_s2(args, eng, step_fn, n_steps=2):
    from easevoice_trainer_amd.hip import conv as HC

    HC.TRACE = []
    try:
        for _ in range(n_steps):
            step_fn()
        torch.cuda.synchronize()
        rec = HC.TRACE
    finally:
        HC.TRACE = None
    agg = {}
    for tag, kind, flops, nbytes, e0, e1 in rec:
        a = agg.setdefault(tag, dict(ms=0.0, calls=0, flops=0.0, bytes=0.0))
        a["ms"] += e0.elapsed_time(e1)
        a["calls"] += 1
        a["flops"] += flops
        a["bytes"] += nbytes
    if not agg:
        return None
    tag, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
    sec = a["ms"] / 1e3
    tf = a["flops"] / sec / 1e12
    gbs = a["bytes"] / sec / 1e9
    peak_tf = MFMA_BF16_PEAK_TF if args.dtype == "bf16" else MFMA_F32_PEAK_TF
    ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
    intensity = a["flops"] / max(a["bytes"], 1.0)
    if intensity >= ridge:
        r = dict(bound="mfma", achieved=tf, peak=peak_tf, unit="TFLOP/s", frac=tf / peak_tf)
    else:
        r = dict(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS)
    r.update(traffic=None, kernel=tag, launches_per_step=a["calls"] / n_steps,
             avg_launch_us=a["ms"] * 1e3 / a["calls"], ms_per_step=a["ms"] / n_steps,
             algorithmic_gflop_per_launch=a["flops"] / a["calls"] / 1e9,
             algorithmic_mb_per_launch=a["bytes"] / a["calls"] / 1e6, intensity_flop_per_byte=intensity,
             also={k: dict(ms_per_step=round(v["ms"] / n_steps, 3), tflops=round(v["flops"] / (v["ms"] / 1e3) / 1e12, 1),
                           gbs=round(v["bytes"] / (v["ms"] / 1e3) / 1e9, 1))
                   for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:8]})
    return r


def cpu_baseline_s2(args, hps, budget_s=30.0):
    from oracle import s2_step as O

    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    B, T, tt = 2, args.clip_seconds * 50, 60
    from easevoice_trainer_amd.module.models import MultiPeriodDiscriminator, SynthesizerTrn

    d, m, t = hps["data"], hps["model"], hps["train"]
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
        return out

    one_step(1)   # warm-up
    times, step = [], 2
    t_start = time.perf_counter()
    while len(times) < 3 and (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        one_step(step)
        times.append(time.perf_counter() - t0)
        step += 1
    med = sorted(times)[len(times) // 2]
    return dict(value=B * args.clip_seconds / med, unit="audio-s/s", cores=threads, kind="port",
                sample=f"oracle s2 step (fwd + D/G backward + AdamW), batch {B} x {args.clip_seconds} s clips, fp32, "
                       f"1 warm-up + {len(times)} timed steps, median {med:.2f} s/step",
                seconds_per_step=med)


def s2_extras(args, eng, world, rank, step_fn=None):
    out = {}
    if step_fn is not None:
        r = roofline_s2(args, eng, step_fn)
        if r is not None:
            out["roofline"] = r
    if world == 1 and rank == 0:
        out["cpu_baseline"] = cpu_baseline_s2(args, eng.hps)
    return out
