#!/bin/bash
# XCD-aware tile order of wgrad_deep: parity, time and FETCH/WRITE bytes per launch with and without it
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04at}
mkdir -p $O
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_bank_gpu.py -x -q 2>&1 | tail -3 | tee $O/tests.txt
for v in 0 1; do
  echo "== EVT_WGRAD_RING_XCD=$v"
  EVT_WGRAD_RING_XCD=$v timeout 200 python tools/bench_conv.py --wonly --only "WN in,FFN 192->768 k3 T200,plain C256" --iters 20 2>&1 | grep -v amdgpu.ids | grep "dP" | cut -c1-170
  for c in FETCH_SIZE WRITE_SIZE; do
    EVT_WGRAD_RING_XCD=$v timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p_$v_$c -- python tools/bench_conv.py --wonly --only "WN in" --iters 3 > $O/p.log 2>&1
    f=$(find $O/p_$v_$c -name '*counter_collection.csv' | head -1)
    python - "$f" $c <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'wgrad_ring' in r['Kernel_Name'] and r['Counter_Name'] == sys.argv[2]:
        agg[r['Dispatch_Id']].append(float(r['Counter_Value']))
vals = [sum(v) for v in agg.values()]
print(f"   {sys.argv[2]} per wgrad_ring launch: {sum(vals)/max(len(vals),1):.0f} KB over {len(vals)} launches (raw counter; FETCH is doubled for bytes per the gfx950 note)")
PY
    rm -rf $O/p_$v_$c
  done
done | tee $O/wgrad_ring_xcd.txt
