"""bench.py's measurement legs beside the headline throughput: per-kernel timing, roofline objects, CPU baselines.

Per-kernel durations come from torch.profiler (kineto -> roctracer/rocprofiler activity records: the same tracer and
clock `rocprofv3 --kernel-trace` reads), collected inside bench.py over eager steps of the timed workload.  The
rocprofv3 summaries of the same command are committed under profiles/ (run with --no-extras: two tracers do not share
a process); the average duration bench.py reports for the dominant kernel is the figure those files show.
Algorithmic flops / bytes per launch: hip/conv.py's trace records (every operand tensor of a launch once).
"""
import json
import os
import re

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable by a copy kernel)
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16
MFMA_F32_PEAK_TF = 157.3


def short_name(sym: str) -> str:
    """'void evt_conv::(anonymous namespace)::conv_ring<2, 2, 4>(evt_conv::ConvP)' -> 'conv_ring<2, 2, 4>'"""
    s = re.sub(r"^void ", "", sym)
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):          # strip the argument list: the first '(' at template depth 0 that is not "(anonymous"
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and not s.startswith("(anonymous namespace)", i):
            cut = i
            break
    s = s[:cut]
    return s.split("::")[-1] if "::" in s else s


def kernel_profile(run):
    """run() under torch.profiler.  Returns (kernels, seq): kernels = {symbol: [calls, total_us]} over every GPU kernel
    of the run; seq = [(start, symbol, us)] in execution order."""
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        run()
        torch.cuda.synchronize()
    kernels, seq = {}, []
    dev = torch.autograd.DeviceType.CUDA
    for e in prof.events():
        if e.device_type == dev:
            if e.name.startswith(("Memcpy", "Memset")):
                continue
            us = float(e.time_range.elapsed_us())
            k = kernels.setdefault(e.name, [0, 0.0])
            k[0] += 1
            k[1] += us
            seq.append((e.time_range.start, e.name, us))
    seq.sort(key=lambda t: t[0])
    dump = os.environ.get("EVT_BENCH_DUMP")
    if dump:
        with open(dump, "a") as f:
            for name, (c, us) in sorted(kernels.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{us / 1e3:10.3f} ms {c:6d} calls {us / c:9.2f} us  {name[:160]}\n")
            f.write("\n")
    return kernels, seq


_MAIN = re.compile(r"^(conv_|wgrad_|grouped_|cout1_|cin1_|resunit_|wn_layer_|rows16_|add3_scale_kernel|lrelu_kernel|dact_mul_kernel|"
                   r"fold_partials_multi)")


def align_records(rec, seq):
    """{record index: kernel microseconds}: trace records (launch order) against the profiler's kernel records (start
    order; one stream, so the same order).  Every traced entry point launches exactly one kernel of the conv family
    (plus, for some weight gradients, a column-sum helper), and the library's tags start with that kernel's function
    name -- so the records are a subsequence of the conv-family kernels of the run; conv-family kernels launched by an
    untraced entry point are stepped over (greedy two-pointer match on the function name).  The element-wise launches of
    hip/conv.py and the fold launch behind a fused ResBlock backward have records of their own (kind "elt")."""
    base = lambda n: n.split("<")[0].strip()
    main = [(base(short_name(n)), us) for _t, n, us in seq if _MAIN.match(base(short_name(n)))]
    out, j = {}, 0
    for i, r in enumerate(rec):
        tb = base(r[0])
        while j < len(main) and not (main[j][0] == tb or main[j][0].startswith(tb) or tb.startswith(main[j][0])):
            j += 1
        if j >= len(main):
            break
        out[i] = main[j][1]
        j += 1
    return out


def _empty_event_pair_us(n=200):
    """HIP-event pair overhead (fallback timing only): elapsed time between two back-to-back event records"""
    vals = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        vals.append((a, b))
    torch.cuda.synchronize()
    v = sorted(x.elapsed_time(y) * 1e3 for x, y in vals)
    return v[len(v) // 2]


def dec_algorithmic_bytes(B, elem_size):
    """HiFi-GAN `dec`, forward + backward, unfused-per-conv accounting: 3 x (52.98 M activation elements
    per item x B + 14.66 M weight elements) x sizeof(dtype)  ->  5.17 GB at B = 16 in bf16"""
    return 3.0 * (52.98e6 * B + 14.66e6) * elem_size


def roofline_s2(args, eng, step_fn, n_steps=2):
    from easevoice_trainer_amd.hip import conv as HC

    rec = []
    graphs, eng.graphs_enabled = eng.graphs_enabled, False   # per-launch records need the eager path
    timing = "torch.profiler kernel records (roctracer, the clock rocprofv3 uses)"
    try:
        HC.set_trace(rec)
        try:
            kernels, seq = kernel_profile(lambda: [step_fn() for _ in range(n_steps)])
            ranges = align_records(rec, seq)
        except Exception as e:     # no tracer in this environment: HIP events minus the measured event-pair overhead
            kernels, ranges, timing = {}, {}, f"hip events minus empty-pair overhead (profiler unavailable: {e!r})"
            del rec[:]
            for _ in range(n_steps):
                step_fn()
            torch.cuda.synchronize()
    finally:
        HC.set_trace(None)
        eng.graphs_enabled = graphs
    if not rec:
        return None
    have = len(ranges) >= 0.98 * len(rec) and sum(ranges.values()) > 0
    ovh = 0.0 if have else _empty_event_pair_us()
    if not have and kernels:
        timing = "hip events minus empty-pair overhead (trace records could not be aligned with the kernel records)"

    def dur_us(i, r):
        if have:
            return ranges.get(i, 0.0)
        return max(r[4].elapsed_time(r[5]) * 1e3 - ovh, 0.5)

    agg = {}
    dec_mods = {id(m) for m in eng.net_g.dec.modules()}
    voc = dict(us=0.0, bytes=0.0, flops=0.0, calls=0, elt_us=0.0, elt_calls=0)
    for i, r in enumerate(rec):
        tag, kind, flops, nbytes, _e0, _e1, _shape, mod = r
        us = dur_us(i, r)
        if id(mod) in dec_mods:
            voc["us"] += us
            voc["bytes"] += nbytes
            voc["flops"] += flops
            voc["calls"] += 1
            if kind == "elt":
                voc["elt_us"] += us
                voc["elt_calls"] += 1
        if kind == "elt":
            continue                       # element-wise launches count for `dec`, not for the dominant-kernel pick
        a = agg.setdefault(tag, dict(us=0.0, calls=0, flops=0.0, bytes=0.0))
        a["us"] += us
        a["calls"] += 1
        a["flops"] += flops
        a["bytes"] += nbytes
    tag, a = max(agg.items(), key=lambda kv: kv[1]["us"])
    sec = a["us"] / 1e6
    tf, gbs = a["flops"] / sec / 1e12, a["bytes"] / sec / 1e9
    peak_tf = MFMA_BF16_PEAK_TF if args.dtype in ("bf16", "f16") else MFMA_F32_PEAK_TF
    ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
    intensity = a["flops"] / max(a["bytes"], 1.0)
    if intensity >= ridge:
        r = dict(bound="mfma", achieved=tf, peak=peak_tf, unit="TFLOP/s", frac=tf / peak_tf)
    else:
        r = dict(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS)
    r_voc = None
    if voc["calls"]:
        vsec = voc["us"] / 1e6
        esz = 2 if args.dtype in ("bf16", "f16") else 4
        alg = dec_algorithmic_bytes(args.batch, esz) * n_steps
        r_voc = dict(bound="hbm", achieved=alg / vsec / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=alg / vsec / 1e9 / HBM_PEAK_GBS, ms_per_step=voc["us"] / 1e3 / n_steps,
                     launches_per_step=voc["calls"] / n_steps, algorithmic_gb_per_step=alg / n_steps / 1e9,
                     elementwise_ms_per_step=voc["elt_us"] / 1e3 / n_steps,
                     elementwise_launches_per_step=voc["elt_calls"] / n_steps,
                     bytes_with_saved_reads_gb_per_step=voc["bytes"] / n_steps / 1e9,
                     gbs_with_saved_reads=voc["bytes"] / vsec / 1e9, tflops=voc["flops"] / vsec / 1e12,
                     note="algorithmic bytes = SURVEY 8(d): 3 x (52.98 M x B + 14.66 M) elements, every conv's input "
                          "and output once per direction; time and launches = EVERY launch of `dec`: convolutions, "
                          "fused / grouped ResBlock steps, and its element-wise launches (leaky-relu copies, stage "
                          "means, activation derivatives -- listed separately as elementwise_*); the fold launch "
                          "behind a fused backward is inside that backward's time")
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if tag in pmc:
            traffic = pmc[tag]["traffic_bytes_per_launch"]
        dec_only = pmc.get("hifigan_dec_only_kernels")
        if r_voc is not None and dec_only:
            # counter bytes (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes) of the kernels only `dec` launches:
            # the fused ResBlock steps, their weight gradients and fold launches -- NOT the shared conv kernels of its
            # up-sampling / C = 256 stage (conv_ring / conv_deep rows above are per launch over all their callers)
            r_voc["traffic"] = dict(dec_only_kernels_gb_per_step=dec_only["traffic_gb_per_step"],
                                    dec_only_kernel_launches_per_step=dec_only["launches_per_step"],
                                    source="profiles/pmc_traffic.json: hifigan_dec_only_kernels (per kernel instantiation there)")
    except Exception:
        traffic = None
    top = sorted(kernels.items(), key=lambda kv: -kv[1][1])[:10]
    r.update(traffic=traffic,
             traffic_unit="bytes per launch from profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                          "passes, FETCH doubled per the gfx950 note), null when not collected for this kernel",
             kernel=tag, timing=timing, hifigan_dec=r_voc, launches_per_step=a["calls"] / n_steps,
             avg_launch_us=a["us"] / a["calls"], ms_per_step=a["us"] / 1e3 / n_steps,
             algorithmic_gflop_per_launch=a["flops"] / a["calls"] / 1e9,
             algorithmic_mb_per_launch=a["bytes"] / a["calls"] / 1e6, intensity_flop_per_byte=intensity,
             also={k: dict(ms_per_step=round(v["us"] / 1e3 / n_steps, 3), avg_us=round(v["us"] / v["calls"], 2),
                           tflops=round(v["flops"] / (v["us"] / 1e6) / 1e12, 1), gbs=round(v["bytes"] / (v["us"] / 1e6) / 1e9, 1))
                   for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["us"])[:8]},
             gpu_kernels_top=[dict(kernel=short_name(k), calls_per_step=v[0] / n_steps, avg_us=round(v[1] / v[0], 2),
                                   ms_per_step=round(v[1] / 1e3 / n_steps, 3)) for k, v in top],
             gpu_kernel_ms_per_step=round(sum(v[1] for v in kernels.values()) / 1e3 / n_steps, 3) if kernels else None,
             at_native_share=(round(sum(v[1] for k, v in kernels.items() if "at::native" in k) /
                                    max(sum(v[1] for v in kernels.values()), 1e-9), 4) if kernels else None))
    return r


def _cpu_subprocess(cmd, hard_timeout_s, unit, what):
    import subprocess

    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=hard_timeout_s, env=env)
        out = r.stdout
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if not lines:
        return dict(value=None, unit=unit, cores=None, kind="port",
                    sample=f"{what} did not finish one timed step within {hard_timeout_s:.0f} s on this host")
    return json.loads(lines[-1])


def cpu_baseline_s2(args, hard_timeout_s=170.0):
    """oracle s2 step at the BENCH shape (B = 16 x 4 s by default) in a subprocess with a hard timeout"""
    import sys

    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--stage", "s2", "--batch", str(args.batch),
           "--clip-seconds", str(args.clip_seconds), "--budget", "60"]
    out = _cpu_subprocess(cmd, hard_timeout_s, "audio-s/s", "oracle s2 step")
    if isinstance(out, dict):
        # kind "port": the GPU box has no reference checkout.  How far the port's step time is from the reference's own was
        # measured where both exist (tools/cpu_baseline.py --with-reference on the build box, 8 threads, same batch)
        out["port_vs_reference"] = ("the oracle port's step takes 0.94-1.02 x the reference's own modules' on the build box "
                                    "(profiles/r05_cpu_port_vs_reference.txt)")
    return out


def cpu_baseline_s1(hard_timeout_s=170.0):
    import sys

    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--stage", "s1", "--budget", "60"]
    return _cpu_subprocess(cmd, hard_timeout_s, "tokens/s", "oracle s1 micro-step")


def s2_extras(args, eng, world, rank, step_fn=None):
    out = {}
    if step_fn is not None:
        r = roofline_s2(args, eng, step_fn)
        if r is not None:
            out["roofline"] = r
    if world == 1 and rank == 0:
        out["cpu_baseline"] = cpu_baseline_s2(args)
    return out
