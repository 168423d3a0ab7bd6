#!/bin/bash
tag=${1:-r06r}
out=gpurun_out/$tag
mkdir -p $out
for v in 0 2 0 2; do
  EVT_DEC_STREAM=1 EVT_ENC_LATE=$v python bench.py --workload s2 --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('enc_late=$v', d['ms_per_step'], d['value'])" | tee -a $out/ab.txt
done
