#!/bin/bash
# round 5, visit l: the driver's bench line on the final tree (record alignment knows the WN layer kernels), run-to-run spread
export TMPDIR=/tmp
O=gpurun_out/r05l; mkdir -p $O; rm -rf $O/*
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05l/bench_line.json') if l.startswith('{')][-1])
r=d['roofline']; print(d['ms_per_step'], r['timing'][:60], r['frac'], r['hifigan_dec']['frac'], r['hifigan_dec']['ms_per_step'], d['s1']['ms_per_step'])
PY
for i in 1 2 3 4; do
  timeout 300 python bench.py --workload s2 --no-extras --steps 20 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms/step', d['config']['launch'])" >> $O/step.txt
done
cat $O/step.txt
timeout 200 python tools/trace_shapes.py --top 400 > $O/conv_time_by_shape.txt 2>&1; grep -i "WN layer" $O/conv_time_by_shape.txt | head
