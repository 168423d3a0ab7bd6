"""How full is the chip over one graph-replayed s2 step?  Reads a rocprofv3 --kernel-trace CSV (bench.py --workload s2 with
graphs), takes the last `--steps` steps (delimited by the generator's AdamW launch) and walks the timeline: at every instant
the kernels in flight and the work-groups they were launched with.  Prints the share of the step during which the launches in
flight together have fewer than 256 / 512 work-groups (one / two per CU), and which kernels run ALONE in those stretches --
the critical-path launches that something else could run beside.

    python tools/timeline_fill.py kernel_trace.csv[.gz] [--steps 6]
"""
import argparse
import collections
import csv
import gzip
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.replace("evt_conv::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    op = gzip.open if args.csv.endswith(".gz") else open
    rows = list(csv.DictReader(op(args.csv, "rt")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ad = [i for i, r in enumerate(rows) if "adamw_flat_dev_kernel" in r["Kernel_Name"]]
    start = ad[-(2 * args.steps + 1)] + 1
    win = rows[start:]
    ev = []
    for i, r in enumerate(win):
        wg = max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        nwg = max(1, grid // wg)
        ev.append((int(r["Start_Timestamp"]), 1, i, nwg))
        ev.append((int(r["End_Timestamp"]), 0, i, nwg))
    ev.sort()
    t0, t1 = ev[0][0], ev[-1][0]
    live = {}
    last = t0
    idle = lt256 = lt512 = 0
    conc = collections.Counter()
    alone = collections.defaultdict(lambda: [0, 0, 0])      # kernel -> [ns alone with < 512 wgs, launches, wgs]
    seen_alone = set()
    queues = collections.Counter(r.get("Queue_Id", "?") for r in win)
    for t, kind, i, nwg in ev:
        dt = t - last
        if dt > 0:
            tot = sum(live.values())
            conc[min(len(live), 4)] += dt
            if not live:
                idle += dt
            else:
                if tot < 256:
                    lt256 += dt
                if tot < 512:
                    lt512 += dt
                    if len(live) == 1:
                        k = next(iter(live))
                        a = alone[short(win[k]["Kernel_Name"])]
                        a[0] += dt
                        if k not in seen_alone:
                            seen_alone.add(k)
                            a[1] += 1
                            a[2] += live[k]
        last = t
        if kind == 1:
            live[i] = nwg
        else:
            live.pop(i, None)
    span = t1 - t0
    n = args.steps
    print(f"{len(win) / n:.0f} kernels/step, wall {span / 1e6 / n:.2f} ms/step, queues {dict(queues)}")
    print(f"idle {idle / 1e6 / n:.2f} ms/step; in flight < 256 work-groups {lt256 / 1e6 / n:.2f}, < 512 {lt512 / 1e6 / n:.2f} ms/step")
    print("time by number of kernels in flight (ms/step): " + ", ".join(f"{k}{'+' if k == 4 else ''}: {v / 1e6 / n:.2f}" for k, v in sorted(conc.items())))
    print("kernels running ALONE with < 512 work-groups (ms/step, launches/step, mean work-groups):")
    for k, a in sorted(alone.items(), key=lambda kv: -kv[1][0])[: args.top]:
        print(f"  {a[0] / 1e6 / n:7.3f} {a[1] / n:7.1f} {a[2] / max(a[1], 1):8.0f}  {k}")


if __name__ == "__main__":
    main()
