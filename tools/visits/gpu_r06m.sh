#!/bin/bash
# upsampler pre-activation: parity + s2 step A/B
tag=${1:-r06m}
out=gpurun_out/$tag
mkdir -p $out
python -m pytest tests/test_conv_gpu.py -q -k "upsampler or parity" -x 2>&1 | tail -5 > $out/conv_tests.txt
cat $out/conv_tests.txt
for v in 0 1 0 1; do
  EVT_CONV_PLAIN_X=$v python bench.py --workload s2 --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain_x=$v', d['ms_per_step'], d['value'])" | tee -a $out/ab.txt
done
