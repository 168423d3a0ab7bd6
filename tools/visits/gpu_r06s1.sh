#!/bin/bash
tag=${1:-r06s1}
out=gpurun_out/$tag
mkdir -p $out
for v in 0 1 0 1; do
  EVT_S1_WGRAD_PAIR=$v python bench.py --workload s1 --steps 24 --warmup 6 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('s1 pair=$v', d['ms_per_step'], d['value'])" | tee -a $out/ab.txt
done
python -m pytest tests/test_s1_gpu.py tests/test_s1_c3_gpu.py tests/test_s1_fp16_gpu.py -q 2>&1 | tail -3
