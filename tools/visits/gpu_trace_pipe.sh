#!/bin/bash
# kernel trace (timestamps) of a few eager s2 steps: do the bookkeeping kernels overlap with the backward?
export TMPDIR=/tmp
O=gpurun_out/${1:-r04t}
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python bench.py --workload s2 --steps 3 --warmup 4 --no-extras --graphs ${2:-0} > $O/tr.log 2>&1
f=$(find $O/tr -name '*kernel_trace.csv' | head -1)
python - "$f" > $O/overlap.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id'), r.get('Stream_Id')) for r in rows]
ks.sort()
t0 = ks[0][0]
book = [k for k in ks if any(s in k[2] for s in ('wn_grad_kernel', 'wn_fold_kernel', 'adamw_flat_dev_kernel'))]
import bisect
starts = [k[0] for k in ks]
for b in book[-40:]:
    # time inside b's interval covered by other kernels
    cov = 0
    n = 0
    for k in ks:
        if k is b: continue
        lo, hi = max(k[0], b[0]), min(k[1], b[1])
        if hi > lo:
            cov += hi - lo; n += 1
    print(f"{(b[0]-t0)/1e6:10.3f} ms  dur {(b[1]-b[0])/1e3:8.1f} us  q={b[3]} s={b[4]}  others inside: {n:3d} kernels, {cov/1e3:8.1f} us   {b[2][:40]}")
PY
cat $O/overlap.txt | cut -c1-220
rm -rf $O/tr
