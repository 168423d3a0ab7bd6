#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r04h}
mkdir -p $O
for v in 0 1 2 3; do
  echo "== EVT_WIDE_V=$v"; EVT_WIDE_V=$v timeout 200 python tools/bench_resunit.py --wide-fwd 2>&1 | grep "^C" 
done | tee $O/wide_var.txt
