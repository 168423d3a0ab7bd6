#!/bin/bash
tag=${1:-r06z}
out=gpurun_out/$tag
mkdir -p $out
for v in 64 16 32 128 64 16 32 128; do
  EVT_WGRAD_DEFER=$v python bench.py --workload s2 --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer=$v', d['ms_per_step'], d['value'])" | tee -a $out/ab.txt
done
for v in 256 128 512 256 128 512; do
  EVT_RING_BLOCKS=$v python bench.py --workload s2 --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ring_blocks=$v', d['ms_per_step'], d['value'])" | tee -a $out/ab.txt
done
