"""Which torch (non-evt) device ops does one s2 step launch?  torch.profiler over two steps, grouped by op and input
shapes, sorted by device time.  Development tool for finding element-wise glue worth fusing.

    python tools/torch_ops.py [--top 80]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=80)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--clip-seconds", type=int, default=4)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--stacks", action="store_true")
    args = ap.parse_args()
    import bench
    from torch.profiler import ProfilerActivity, profile

    world, rank, local = bench.init_dist(1)
    res, eng, step = bench.run_s2(args, world, rank, local)
    # forward ranges per sub-module (two levels below net_g / net_d) so ops can be attributed
    from torch.autograd.profiler import record_function

    def hook(mod, name):
        st = []
        mod.register_forward_pre_hook(lambda m, a: st.append(record_function("mod:" + name).__enter__()))
        mod.register_forward_hook(lambda m, a, o: st.pop().__exit__(None, None, None))

    for top, net in (("g", eng.net_g), ("d", eng.net_d)):
        for n1, m1 in net.named_children():
            hook(m1, f"{top}.{n1}")
            for n2, m2 in m1.named_children():
                hook(m2, f"{top}.{n1}.{n2}")
    n = 2
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
                 with_stack=args.stacks) as prof:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=6 if args.stacks else 0)
    rows = []
    for e in ka:
        dev = getattr(e, "self_device_time_total", None)
        if dev is None:
            dev = getattr(e, "self_cuda_time_total", 0)
        if dev <= 0:
            continue
        rows.append((dev / n, e.count // n, e.key, str(e.input_shapes)[:110], e.self_cpu_time_total / n,
                     list(getattr(e, "stack", []) or [])))
    # ---- attribute forward ops (device-time > 0) to the innermost "mod:" range ----
    by_mod, by_name = {}, {}
    for ev in prof.events():
        dev = getattr(ev, "self_device_time_total", None)
        if dev is None:
            dev = getattr(ev, "self_cuda_time_total", 0)
        if dev <= 0 or ev.name.startswith("mod:"):
            continue
        a = by_name.setdefault(ev.name[:60], [0, 0.0])
        a[0] += 1
        a[1] += dev
        par, tag = ev.cpu_parent, None
        while par is not None:
            if par.name.startswith("mod:"):
                tag = par.name
                break
            par = par.cpu_parent
        b = by_mod.setdefault(tag or "(backward / outside modules)", [0, 0.0])
        b[0] += 1
        b[1] += dev
    print("---- by op name (per step) ----")
    for k, v in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{v[1] / n:9.1f} us {v[0] // n:6d}  {k}")
    print("---- forward ops by module range (per step) ----")
    for k, v in sorted(by_mod.items(), key=lambda kv: -kv[1][1]):
        print(f"{v[1] / n:9.1f} us {v[0] // n:6d}  {k}")
    rows.sort(key=lambda r: -r[0])
    print(f"step {res['ms_per_step']:.1f} ms; ops with device time: {sum(r[0] for r in rows) / 1e3:.2f} ms/step, "
          f"{sum(r[1] for r in rows)} calls/step")
    print(f"{'dev us':>9} {'calls':>5} {'cpu us':>8}  op | shapes")
    for dev, cnt, key, shp, cpu, stack in rows[: args.top]:
        print(f"{dev:9.1f} {cnt:5d} {cpu:8.1f}  {key[:40]:40s} | {shp}")
        for fr in stack[:6]:
            if "easevoice" in fr or "s2_engine" in fr:
                print(f"{'':26s}{fr[-110:]}")


if __name__ == "__main__":
    main()
