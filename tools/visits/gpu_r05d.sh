#!/bin/bash
# round 5, visit 4: tile / ring-depth variants of conv_ring on the 3200-position layers (EVT_RING_TILE), stand-alone and in the step
export TMPDIR=/tmp
O=gpurun_out/${1:-r05d}
mkdir -p $O
for t in 224 244 243 424 423 443 223 226; do
  echo "== EVT_RING_TILE=$t"
  EVT_RING_TILE=$t timeout 200 python tools/bench_conv.py --only "WN in,WN rs,FFN,res C256" --iters 50 2>&1 | grep -v amdgpu.ids | cut -c1-36,52-75,92-113
done | tee $O/ring_tiles.txt
for r in 1 2; do for t in 224 244 424 443; do
  EVT_RING_TILE=$t timeout 300 python bench.py --workload s2 --steps 30 --warmup 5 --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('EVT_RING_TILE=$t run $r:', round(d['ms_per_step'],3), 'ms')"
done; done | tee $O/ring_step.txt
