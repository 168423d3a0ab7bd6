#!/bin/bash
# sub-discriminators on branch streams: parity + s2 step A/B
tag=${1:-r06n}
out=gpurun_out/$tag
mkdir -p $out
python -m pytest tests/test_disc_gen_gpu.py -q -x 2>&1 | tail -5 > $out/tests.txt
cat $out/tests.txt
for v in 1 2 1 2; do
  EVT_MPD_STREAMS=$v python bench.py --workload s2 --steps 30 --warmup 8 --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mpd_streams=$v', d['ms_per_step'], d['value'])" | tee -a $out/ab.txt
done
