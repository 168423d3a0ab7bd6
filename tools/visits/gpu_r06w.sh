#!/bin/bash
tag=${1:-r06w}
out=gpurun_out/$tag
mkdir -p $out
for v in 0 1 0 1; do
  EVT_MEL_LANE=$v python bench.py --workload s2 --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mel_lane=$v', d['ms_per_step'], d['value'])" | tee -a $out/ab.txt
done
