#!/bin/bash
# kernel list of the GRAPH-REPLAYED s2 step for a given bench configuration: bash tools/visits/graphstats.sh <out file> [bench args / env already exported]
export TMPDIR=/tmp
OUT=$1; shift
P=$(mktemp -d /tmp/gs.XXXX)
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $P -- python bench.py --workload s2 --steps 6 --warmup 4 --no-extras "$@" > $P/log 2>&1
f=$(find $P -name '*kernel_trace.csv' | head -1)
python - "$f" > $OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if 'adamw_flat_dev_kernel' in r['Kernel_Name']]
start = ad[-13] + 1 if len(ad) >= 13 else 0
win = rows[start:]
steps = 6
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    a = agg[r['Kernel_Name']]
    a[0] += 1; a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(v[1] for v in agg.values()); n = sum(v[0] for v in agg.values())
span = int(win[-1]['End_Timestamp']) - int(win[0]['Start_Timestamp'])
# time during which NO kernel is running (gaps on the merged timeline)
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in win)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"window: {n/steps:.0f} kernels/step, kernel time {tot/1e6/steps:.2f} ms/step, wall {span/1e6/steps:.2f} ms/step, "
      f"device idle (no kernel running) {(span-busy)/1e6/steps:.2f} ms/step")
def lib(nm): return not (nm.startswith('void at::') or 'rocclr' in nm or 'rocprim' in nm or nm.startswith('at::') or 'hipcub' in nm)
t = [(v[1], v[0], k) for k, v in agg.items() if not lib(k)]
print(f"torch/runtime kernels: {sum(x[1] for x in t)/steps:.0f} per step, {sum(x[0] for x in t)/1e6/steps:.2f} ms/step")
for d, c, k in sorted(t, reverse=True)[:40]:
    print(f"{d/1e6/steps:7.3f} ms {c/steps:7.1f} calls {d/c/1e3:7.1f} us  {k[:140]}")
print("library kernels:")
t = [(v[1], v[0], k) for k, v in agg.items() if lib(k)]
for d, c, k in sorted(t, reverse=True)[:60]:
    print(f"{d/1e6/steps:7.3f} ms {c/steps:7.1f} calls {d/c/1e3:7.1f} us  {k[:140]}")
PY
rm -rf $P
head -3 $OUT | cut -c1-200
