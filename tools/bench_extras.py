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

import torch

HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TF = 2500.0
MFMA_F32_PEAK_TF = 157.3


def roofline_s2(args, eng, step_fn, n_steps=2):
    from easevoice_trainer_amd.hip import conv as HC

    HC.set_trace([])
    graphs, eng.graphs_enabled = eng.graphs_enabled, False   # per-launch events need the eager path
    try:
        for _ in range(n_steps):
            step_fn()
        torch.cuda.synchronize()
        rec = HC.TRACE
    finally:
        HC.set_trace(None)
        eng.graphs_enabled = graphs
    agg = {}
    dec_mods = {id(m) for m in eng.net_g.dec.modules()}
    voc = dict(ms=0.0, bytes=0.0, flops=0.0, calls=0)
    for tag, kind, flops, nbytes, e0, e1, _shape, mod in rec:
        if id(mod) in dec_mods:
            voc["ms"] += e0.elapsed_time(e1)
            voc["bytes"] += nbytes
            voc["flops"] += flops
            voc["calls"] += 1
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
    # the HiFi-GAN vocoder (`dec`) convolutions forward + backward against the HBM roofline, with the
    # per-launch algorithmic bytes (every operand tensor once) summed over its 91 convs x (fwd, bwd-data, bwd-weight)
    if voc["calls"]:
        vsec = voc["ms"] / 1e3
        r_voc = dict(bound="hbm", achieved=voc["bytes"] / vsec / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=voc["bytes"] / vsec / 1e9 / HBM_PEAK_GBS, ms_per_step=voc["ms"] / n_steps,
                     launches_per_step=voc["calls"] / n_steps, algorithmic_gb_per_step=voc["bytes"] / n_steps / 1e9,
                     tflops=voc["flops"] / vsec / 1e12,
                     note="late stages (C <= 64) are HBM/launch-bound, the k=7/11 convs at C >= 64 are MFMA-bound")
    else:
        r_voc = None
    # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process; they come from the
    # separate rocprofv3 --pmc passes recorded in profiles/r01_pmc_traffic.json (same command, same shapes)
    traffic = None
    try:
        pmc = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                                          "r01_pmc_traffic.json")))
        if tag in pmc:
            traffic = pmc[tag]["traffic_bytes_per_launch"]
    except Exception:
        traffic = None
    r.update(traffic=traffic, traffic_unit="bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate pass)",
             kernel=tag, hifigan_dec=r_voc, launches_per_step=a["calls"] / n_steps,
             avg_launch_us=a["ms"] * 1e3 / a["calls"], ms_per_step=a["ms"] / n_steps,
             algorithmic_gflop_per_launch=a["flops"] / a["calls"] / 1e9,
             algorithmic_mb_per_launch=a["bytes"] / a["calls"] / 1e6, intensity_flop_per_byte=intensity,
             also={k: dict(ms_per_step=round(v["ms"] / n_steps, 3), tflops=round(v["flops"] / (v["ms"] / 1e3) / 1e12, 1),
                           gbs=round(v["bytes"] / (v["ms"] / 1e3) / 1e9, 1))
                   for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:8]})
    return r


def cpu_baseline_s2(args, hps, hard_timeout_s=150.0):
    """runs tools/cpu_baseline.py in a subprocess; whatever it printed before the hard timeout is reported"""
    import subprocess
    import sys

    cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpu_baseline.py"),
           "--batch", "2", "--clip-seconds", str(args.clip_seconds), "--budget", "40"]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=hard_timeout_s, env=env)
        out = r.stdout
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if not lines:
        return dict(value=None, unit="audio-s/s", cores=None, kind="port",
                    sample=f"oracle s2 step did not finish one timed step within {hard_timeout_s:.0f} s on this host")
    return json.loads(lines[-1])


def s2_extras(args, eng, world, rank, step_fn=None):
    out = {}
    if step_fn is not None:
        r = roofline_s2(args, eng, step_fn)
        if r is not None:
            out["roofline"] = r
    if world == 1 and rank == 0:
        out["cpu_baseline"] = cpu_baseline_s2(args, eng.hps)
    return out
