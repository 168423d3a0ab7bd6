#!/bin/bash
# round 5, visit n: HBM traffic of the WN layer kernels (two PMC passes: FETCH_SIZE, WRITE_SIZE) at the B = 16 x 200 shape
export TMPDIR=/tmp
O=gpurun_out/r05n; mkdir -p $O/f $O/w
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python -m pytest tests/test_wn_layer_gpu.py -q -k "16-200 and bf16 and equals" > $O/f.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python -m pytest tests/test_wn_layer_gpu.py -q -k "16-200 and bf16 and equals" > $O/w.log 2>&1
python - <<'PY'
import csv, glob, collections
def means(pat, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(set)
    for path in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter or "wn_layer" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("::")[-1][:48]
            tot[k] += float(r["Counter_Value"]); cnt[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    return {k: (tot[k] / len(cnt[k]), len(cnt[k])) for k in tot}
f = means("gpurun_out/r05n/f/**/*counter_collection.csv", "FETCH_SIZE"); w = means("gpurun_out/r05n/w/**/*counter_collection.csv", "WRITE_SIZE")
with open("gpurun_out/r05n/wn_layer_traffic.txt", "w") as out:
    for k in sorted(f):
        fk, n = f[k]; wk = w.get(k, (0.0, 0))[0]
        line = f"{k:50s} dispatches {n:2d}  FETCH_SIZE {fk:9.1f} KB raw (x2 = {2*fk/1024:6.2f} MB)  WRITE_SIZE {wk/1024:6.2f} MB  traffic {(2*fk+wk)/1024:6.2f} MB per launch"
        print(line); out.write(line + "\n")
PY
rm -rf $O/f $O/w
