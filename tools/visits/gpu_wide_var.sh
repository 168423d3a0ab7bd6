#!/bin/bash
# weight-ring depth of csrc/resunit_wide.hip: fused forward at the B=16 vocoder shapes for each EVT_WIDE_R
export TMPDIR=/tmp
O=gpurun_out/${1:-r04h}
mkdir -p $O
for v in 4 5 6 8 12; do
  echo "== EVT_WIDE_R=$v"; EVT_WIDE_R=$v timeout 200 python tools/bench_resunit.py --wide-fwd 2>&1 | grep "^C" 
done | tee $O/wide_ring.txt
