#!/bin/bash
# kernel stats of the GRAPH-REPLAYED s2 step (what the bench line times), torch / runtime kernels listed separately
export TMPDIR=/tmp
O=gpurun_out/${1:-r04af}
mkdir -p $O/p
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/p -- python bench.py --workload s2 --steps 6 --warmup 4 --no-extras > $O/p.log 2>&1
f=$(find $O/p -name '*kernel_trace.csv' | head -1)
python - "$f" > $O/replay_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 6 steps are replays: find them as the last 6/10 of the wn_fold... simpler: take kernels after the last capture -> use the final 50% of the timeline by count of 'adamw' launches
ad = [i for i, r in enumerate(rows) if 'adamw_flat_dev_kernel' in r['Kernel_Name']]
# two adamw per step (D, G); last 6 steps = last 12 launches; window from the G-adamw of step -7 to the end
start = ad[-13] + 1 if len(ad) >= 13 else 0
win = rows[start:]
steps = 6
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    a = agg[r['Kernel_Name']]
    a[0] += 1; a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(v[1] for v in agg.values()); n = sum(v[0] for v in agg.values())
span = int(win[-1]['End_Timestamp']) - int(win[0]['Start_Timestamp'])
print(f"window: {n/steps:.0f} kernels/step, kernel time {tot/1e6/steps:.2f} ms/step, wall {span/1e6/steps:.2f} ms/step")
def lib(nm): return not (nm.startswith('void at::') or 'rocclr' in nm or 'rocprim' in nm or nm.startswith('at::') or 'hipcub' in nm)
t = [(v[1], v[0], k) for k, v in agg.items() if not lib(k)]
print(f"torch/runtime kernels: {sum(x[1] for x in t)/steps:.0f} per step, {sum(x[0] for x in t)/1e6/steps:.2f} ms/step")
for d, c, k in sorted(t, reverse=True)[:40]:
    print(f"{d/1e6/steps:7.3f} ms {c/steps:7.1f} calls {d/c/1e3:7.1f} us  {k[:140]}")
print("library kernels:")
t = [(v[1], v[0], k) for k, v in agg.items() if lib(k)]
for d, c, k in sorted(t, reverse=True)[:45]:
    print(f"{d/1e6/steps:7.3f} ms {c/steps:7.1f} calls {d/c/1e3:7.1f} us  {k[:140]}")
PY
rm -rf $O/p
head -50 $O/replay_kernels.txt | cut -c1-200
