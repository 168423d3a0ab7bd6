#!/bin/bash
tag=${1:-r06z}
out=gpurun_out/$tag
mkdir -p $out
for v in 16 12 20 24 16 12 20 24; do
  EVT_WGRAD_PARTS_CAP=$v python bench.py --workload s2 --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('parts_cap=$v', d['ms_per_step'], d['value'])" | tee -a $out/ab.txt
done
